#!/bin/bash
# bench.py flag matrix: nothing may crash, every line must parse
set -o pipefail
O=gpurun_out/r4s; mkdir -p $O
i=0
for args in "--size 256 --in-flight 1" "--size 256 --in-flight 4" "--size 1024 --in-flight 3 --unfused" "--size 4096 --in-flight 2 --unfused --passes" "--size 128 --in-flight 2" "--size 2048 --in-flight 2 --range-cus 64" "--size 16384 --in-flight 4 --steps 12 --warmup 2" "--size 16384 --in-flight 2 --unfused --steps 12 --warmup 2"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py $args --no-cpu --batch-frames 6 --batch-size 1024 --steps 20 --warmup 3 > $O/m$i.json 2> $O/m$i.err || { echo "FAILED: $args"; tail -15 $O/m$i.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/m$i.json').read().strip().splitlines()[-1])
print('%-60s %.4f ms/step  roofline %.3f  batch64 %.0f  in-flight %s' % ('$args', d['ms_per_step'], d['roofline']['frac'], d['batch64']['value'], d['config']['frames_in_flight_per_gpu']))"
done
