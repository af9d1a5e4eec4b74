#!/bin/bash
# one fresh box: the default line (two frames in flight) and --in-flight 1, twice each in ABAB order, GPU legs only
set -o pipefail
O=gpurun_out/r4box; mkdir -p $O
TAG=$(date +%H%M%S)
for rep in 1 2; do
  for fl in 2 1; do
    timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-cpu --no-batch --in-flight $fl > $O/b_${TAG}_${fl}_$rep.json 2>/dev/null || { echo FAILED; exit 1; }
    python3 -c "
import json; d=json.loads(open('$O/b_${TAG}_${fl}_$rep.json').read().strip().splitlines()[-1])
print('box $TAG in-flight $fl: %.3f ms/step %.1f frames/s' % (d['ms_per_step'], d['value']))" | tee -a $O/boxes.log
  done
done
