#!/bin/bash
# round 4, call ae: closing validation at the last library (whole GPU suite, smoke, default bench)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r4ae; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { echo "GPU TESTS FAILED"; tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "bench default FAILED"; tail -20 $O/bench_default.err; exit 1; }
tail -c 400 $O/bench_default.json
