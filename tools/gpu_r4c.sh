#!/bin/bash
# round 4, call c: per-kernel trace of slab mode (fixed output format), 8192^2 slab x streams, new tests, host-array call
set -o pipefail
O=gpurun_out/r4c; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or fallback or global_max or rccl or two_channel" > $O/tests_parity.log 2>&1 || { echo "parity tests FAILED"; tail -30 $O/tests_parity.log; exit 1; }
tail -2 $O/tests_parity.log
timeout -k 10 600 python -m pytest tests/test_gpu_batch64.py -x -q -m gpu -k "two_ranks" > $O/tests_batch.log 2>&1 || { echo "batch tests FAILED"; tail -30 $O/tests_batch.log; exit 1; }
tail -2 $O/tests_batch.log
timeout -k 10 120 ./tools/pcibench.bin > $O/pcibench.log 2>&1; tail -12 $O/pcibench.log
timeout -k 10 300 python3 tools/bench_hostpath.py 16384 --json $O/hostpath_16384.json > $O/hostpath_16384.log 2>&1 || { echo "hostpath FAILED"; tail -20 $O/hostpath_16384.log; exit 1; }
cat $O/hostpath_16384.log
timeout -k 10 300 python3 tools/bench_hostpath.py 8192 --json $O/hostpath_8192.json > $O/hostpath_8192.log 2>&1 || { echo "hostpath FAILED"; tail -20 $O/hostpath_8192.log; exit 1; }
cat $O/hostpath_8192.log
for cfg in "0 1" "32 4" "64 4" "128 4"; do
  set -- $cfg
  SARX_SLAB_MIB=$1 SARX_SLAB_STREAMS=$2 timeout -k 10 200 python bench.py --size 8192 --no-cpu --no-batch --steps 50 --warmup 3 > $O/bench8192_slab$1_s$2.json 2> $O/bench8192_slab$1_s$2.err || { echo "bench8192 $cfg failed"; exit 1; }
  python - "8192 slab $1 MiB x $2 streams" $O/bench8192_slab$1_s$2.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:>32s}  {d['ms_per_step']:.3f} ms  {d['value']:.1f} frames/s")
PY
done
cd /tmp && export TMPDIR=/tmp
for cfg in "0 1" "64 1" "64 4" "32 8"; do
  set -- $cfg
  SARX_SLAB_MIB=$1 SARX_SLAB_STREAMS=$2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_slab$1_s$2 -- python3 $R/tools/run_pass.py 0 16384 10 > $R/$O/prof_slab$1_s$2.log 2>&1 || { echo "prof $cfg FAILED"; tail -5 $R/$O/prof_slab$1_s$2.log; exit 1; }
  grep "^pass" $R/$O/prof_slab$1_s$2.log
  f=$(find /tmp/prof_slab$1_s$2 -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/$O/slab$1_s$2_kernel_stats.csv
  python3 - $R/$O/slab$1_s$2_kernel_stats.csv <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    if 'fill_noise' in r['Name']: continue
    tot+=float(r['TotalDurationNs'])
    if float(r['Percentage'])>1: print(f"   {r['Name'][:70]:70s} {r['Calls']:>5s} x {float(r['AverageNs'])/1e3:9.1f} us = {float(r['TotalDurationNs'])/1e6:8.3f} ms")
print(f"   sum of kernel durations over 12 frames: {tot/1e6:.3f} ms = {tot/12e6:.3f} ms per frame")
PY
done
