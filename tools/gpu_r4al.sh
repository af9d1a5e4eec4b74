#!/bin/bash
set -o pipefail
O=gpurun_out/r4al; mkdir -p $O
for fl in 2 3 4 2 3 4; do
timeout -k 10 300 python bench.py --steps 60 --warmup 6 --no-cpu --no-batch --in-flight $fl > $O/b_$fl.json 2>/dev/null || { echo "bench FAILED"; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/b_$fl.json').read().strip().splitlines()[-1])
print('in-flight $fl: %.3f ms/step %.1f fps | lanes %s probe %s' % (d['ms_per_step'], d['value'], d['config']['lanes'], d['config']['lane_probe_ratio']))" | tee -a $O/inflight3.log
done
