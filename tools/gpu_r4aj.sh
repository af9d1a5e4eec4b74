#!/bin/bash
set -o pipefail
O=gpurun_out/r4aj; mkdir -p $O /tmp/ex
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch64.py tests/test_gpu_example.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --batch-reps 1 > $O/bench_$i.json 2>/dev/null || { echo "bench FAILED"; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1])
print('bench run $i: %.3f ms/step %.1f fps | lanes %s probe %s | one in flight %.3f | batch64 %.1f' % (d['ms_per_step'], d['value'], d['config']['lanes'], d['config']['lane_probe_ratio'], d['one_frame_in_flight']['ms_per_step'], d['batch64']['value']))"
done
