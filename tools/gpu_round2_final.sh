#!/bin/bash
# what the driver runs at round end, plus the 2-rank launcher rehearsal
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== gpu suite"; timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_final.log 2>&1; echo "rc $?"; tail -3 gpurun_out/t_final.log
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench default"; timeout -k 10 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "rc $?"; python3 -c "
import json; d=json.load(open('gpurun_out/bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['batch64']['value'], d['cpu_baseline']['value'])"
echo "== 2-rank rehearsal"; timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --batch-frames 8 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err; echo "rc $?"; python3 -c "
import json; d=json.load(open('gpurun_out/bench_2rank.json')); print(d['n_gpus'], d['value'], d['collective'], d['batch64']['value'])"
