#!/bin/bash
# round 3, call l: what the driver runs at round end (build check is CPU-side): GPU tests, smoke, default bench; + config-3 block with its CPU baseline
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3l; O=gpurun_out/r3l
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "tests rc $?"; tail -4 $O/gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
/usr/bin/time -v timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; grep -E "Elapsed|Maximum resident" $O/bench_default.err; head -c 600 $O/bench_default.json; echo
timeout -k 10 400 python bench.py --size 8192 --steps 40 --warmup 5 --no-batch --config3 > $O/bench_8192.json 2>/dev/null; echo "bench 8192 rc $?"
python3 -c "
import json
d=json.load(open('$O/bench_8192.json')); print(d['value'], d['config3_two_channel'])"
