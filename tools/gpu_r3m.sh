#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3m; O=gpurun_out/r3m
S=$(date +%s); timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? in $(( $(date +%s) - S )) s"; head -c 500 $O/bench_default.json; echo
timeout -k 10 300 python -m pytest tests/test_gpu_rda.py -x -q > $O/rda_tests.log 2>&1; echo "rda tests rc $?"; tail -2 $O/rda_tests.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_videosar -- python3 $R/tools/bench_videosar.py > $R/$O/videosar_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find $O/prof_videosar -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/videosar_kernel_stats.csv && head -10 $f | cut -c1-130
rm -rf $O/prof_videosar
