#!/bin/bash
set -o pipefail
O=gpurun_out/r4h; mkdir -p $O
for cus in 176 192 208 224 240 0; do
  echo "== SARX_RANGE_CUS=$cus" | tee -a $O/range_cus_sweep2.log
  SARX_RANGE_CUS=$cus timeout -k 10 300 python3 tools/bench_two_streams.py 16384 60 lanes ctx 2>&1 | grep "2 frame" | tee -a $O/range_cus_sweep2.log || { echo FAILED; exit 1; }
done
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu --no-batch > $O/bench_$i.json 2> $O/bench_$i.err || { echo "bench FAILED"; tail -20 $O/bench_$i.err; exit 1; }
python3 - $O/bench_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench default: %.3f ms/step %.1f frames/s | in flight %s cus %s | one in flight %.3f ms | roofline frac %.3f (solo %.3f)" % (
  d['ms_per_step'], d['value'], d['config']['frames_in_flight_per_gpu'], d['config']['range_launch_cus'], d['one_frame_in_flight']['ms_per_step'], d['roofline']['frac'], d['roofline_solo']['frac']))
PY
done
