#!/bin/bash
# round 3, call e: rgbench layout / prefetch sweep after the scalar-load + wait-state fixes; RDA + benchsize + batch64 tests; RDA profile
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3e
for v in p0h7L0 p0h7L1 p3h4L0 p3h4L1 p5h4L0 p5h4L1 p1h4L1 p2h4L1 p3h4L1nt0 p3h4L0nt0 p3h0L1 p4h4L1; do timeout -k 10 100 tools/rgbench_$v.bin > gpurun_out/r3e/rgbench_$v.log 2>&1; echo "$v rc $?"; done
timeout -k 10 900 python -m pytest tests/test_gpu_rda.py tests/test_gpu_benchsize.py tests/test_gpu_batch64.py -x -q > gpurun_out/r3e/tests.log 2>&1; echo "tests rc $?"; tail -15 gpurun_out/r3e/tests.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3e/prof_videosar -- python3 $R/tools/bench_videosar.py > $R/gpurun_out/r3e/videosar_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find gpurun_out/r3e/prof_videosar -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r3e/videosar_kernel_stats.csv && head -12 $f
