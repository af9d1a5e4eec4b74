#!/bin/bash
set -o pipefail
O=gpurun_out/r4ak; mkdir -p $O
TAG=$(date +%H%M%S)
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu --no-batch > $O/b_${TAG}_$i.json 2>/dev/null || { echo "bench FAILED"; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/b_${TAG}_$i.json').read().strip().splitlines()[-1])
print('box $TAG run $i: %.3f ms/step %.1f fps | lanes %s probe %s | one in flight %.3f' % (d['ms_per_step'], d['value'], d['config']['lanes'], d['config']['lane_probe_ratio'], d['one_frame_in_flight']['ms_per_step']))" | tee -a $O/boxes.log
done
