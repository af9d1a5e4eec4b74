#!/bin/bash
# round 3, call d: rgbench prefetch / nt sweep, RDA direct-route tests, RDA kernel profile
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3d
for v in p3h4nt0 p3h4nt1 p3h4nt2 p3h4nt3 p3h5nt3 p3h7nt3 p3h0nt3 p4h4nt3 p5h4nt3 p4h4nt0 p3h4alt; do timeout -k 10 100 tools/rgbench_$v.bin > gpurun_out/r3d/rgbench_$v.log 2>&1; echo "$v rc $?"; done
grep -h "WP_PRE\|wp  FFT\|wp  IFFT" gpurun_out/r3d/rgbench_*.log > gpurun_out/r3d/summary.txt
timeout -k 10 500 python -m pytest tests/test_gpu_rda.py -x -q > gpurun_out/r3d/rda_tests.log 2>&1; echo "rda tests rc $?"; tail -15 gpurun_out/r3d/rda_tests.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3d/prof_videosar -- python3 $R/tools/bench_videosar.py > $R/gpurun_out/r3d/videosar_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find gpurun_out/r3d/prof_videosar -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r3d/videosar_kernel_stats.csv && head -30 $f
cat gpurun_out/r3d/videosar_profiled.log | tail -5
