#!/bin/bash
# the driver's own command line (BENCH_r03.json: `python3 bench.py --gpus 1 --steps 20 --warmup 5`), three times, then with --in-flight 1
set -o pipefail
O=gpurun_out/r4ab; mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_$i.json 2>/dev/null || { echo FAILED; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/driver_$i.json').read().strip().splitlines()[-1])
print('driver cmd run $i: %.3f ms/step %.1f frames/s | one in flight %.3f ms | roofline %.3f | batch64 %.1f' % (d['ms_per_step'], d['value'], d['one_frame_in_flight']['ms_per_step'], d['roofline']['frac'], d['batch64']['value']))" | tee -a $O/driver.log
done
for i in 1 2; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --in-flight 1 --no-cpu --no-batch > $O/solo_$i.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/solo_$i.json').read().strip().splitlines()[-1]); print('  --in-flight 1 run $i: %.3f ms/step %.1f frames/s' % (d['ms_per_step'], d['value']))" | tee -a $O/driver.log
done
