#!/bin/bash
# round 4, call a: (1) column-strip single-launch traffic experiment, (2) slab mode with the groups' chains on several streams
set -o pipefail
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a
timeout -k 10 240 ./tools/colbench.bin > $O/colbench.log 2>&1 || { echo "colbench failed"; tail -5 $O/colbench.log; exit 1; }
cat $O/colbench.log
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu --no-batch --steps 30 --warmup 3 > $O/bench_$label.json 2> $O/bench_$label.err || { echo "bench $label failed"; tail -5 $O/bench_$label.err; return 1; }
  python - "$label" $O/bench_$label.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:>28s}  {d['ms_per_step']:.3f} ms  {d['value']:.1f} frames/s")
PY
}
run default SARX_X=0 &&
run slab16_s1 SARX_SLAB_MIB=16 SARX_SLAB_STREAMS=1 &&
run slab16_s2 SARX_SLAB_MIB=16 SARX_SLAB_STREAMS=2 &&
run slab16_s3 SARX_SLAB_MIB=16 SARX_SLAB_STREAMS=3 &&
run slab16_s4 SARX_SLAB_MIB=16 SARX_SLAB_STREAMS=4 &&
run slab32_s2 SARX_SLAB_MIB=32 SARX_SLAB_STREAMS=2 &&
run slab32_s3 SARX_SLAB_MIB=32 SARX_SLAB_STREAMS=3 &&
run slab32_s4 SARX_SLAB_MIB=32 SARX_SLAB_STREAMS=4 &&
run slab64_s2 SARX_SLAB_MIB=64 SARX_SLAB_STREAMS=2 &&
run slab64_s3 SARX_SLAB_MIB=64 SARX_SLAB_STREAMS=3 &&
run slab64_s4 SARX_SLAB_MIB=64 SARX_SLAB_STREAMS=4 &&
run slab32_s8 SARX_SLAB_MIB=32 SARX_SLAB_STREAMS=8 &&
run default2 SARX_X=0
