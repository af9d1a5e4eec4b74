#!/bin/bash
# round 4, call e: frames in flight (lanes) in bench.py and the batch driver
set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_batch64.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1 || { echo "tests FAILED"; tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for l in 1 2 3; do
  timeout -k 10 300 python3 tools/bench_batch64.py --lanes $l > $O/batch64_lanes$l.json 2> $O/batch64_lanes$l.err || { echo "batch64 lanes $l FAILED"; tail -20 $O/batch64_lanes$l.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$O/batch64_lanes$l.json').read().strip().splitlines()[-1]); print('batch64 lanes $l: %.1f frames/s  %.3f ms/frame' % (d['value'], d['ms_per_frame']))"
done
timeout -k 10 300 python3 tools/bench_batch64.py --lanes 2 --stack products > $O/batch64_products_lanes2.json 2> $O/batch64_products_lanes2.err && tail -1 $O/batch64_products_lanes2.json | cut -c1-200
timeout -k 10 600 python bench.py --steps 100 --warmup 5 --passes > $O/bench_default.json 2> $O/bench_default.err || { echo "bench FAILED"; tail -20 $O/bench_default.err; exit 1; }
python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench default: %.3f ms/step %.1f frames/s | in flight %s | one in flight %.3f ms | roofline frac %.3f (solo %.3f) | batch64 %.1f | cpu %.4f" % (
  d['ms_per_step'], d['value'], d['config']['frames_in_flight_per_gpu'], d['one_frame_in_flight']['ms_per_step'], d['roofline']['frac'], d['roofline_solo']['frac'], d['batch64']['value'], d['cpu_baseline']['value']))
PY
timeout -k 10 300 python bench.py --steps 100 --warmup 5 --in-flight 1 --no-cpu --no-batch > $O/bench_inflight1.json 2> $O/bench_inflight1.err || { echo "bench FAILED"; tail -20 $O/bench_inflight1.err; exit 1; }
python3 -c "import json; d=json.loads(open('$O/bench_inflight1.json').read().strip().splitlines()[-1]); print('bench --in-flight 1: %.3f ms/step %.1f frames/s roofline frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))"
