#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3n; O=gpurun_out/r3n
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch64.py -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
for i in 1 2 3; do timeout -k 10 300 python tools/bench_batch64.py; done 2>&1 | cut -c1-220 | tee $O/batch64.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_b64 -- python3 $R/tools/bench_batch64.py --frames 16 > $R/$O/b64_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find $O/prof_b64 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/batch64_kernel_stats.csv && head -9 $f | cut -c1-130
rm -rf $O/prof_b64
