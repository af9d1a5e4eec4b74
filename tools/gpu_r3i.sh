#!/bin/bash
# round 3, call i: why does the forward permuted pass vary between 0.73 and 0.89 ms from process to process while the inverse does not?
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3i; O=gpurun_out/r3i
for off in 0 4 68 1028 0; do timeout -k 10 100 tools/rgbench_p5.bin 16384 $off >> $O/p5_offsets.log 2>&1; echo "p5 off $off rc $?"; done
for v in p6 p3 p5L1 p6L1 p5nt0 p0 p6 p5L1; do timeout -k 10 100 tools/rgbench_$v.bin >> $O/$v.log 2>&1; echo "$v rc $?"; done
grep -h "output offset\|wp  FFT\|wp  IFFT" $O/p5_offsets.log | awk '{print $0}' | cut -c1-100
for v in p6 p3 p5L1 p6L1 p5nt0 p0; do echo "== $v"; grep -h "wp  FFT\|wp  IFFT" $O/$v.log | cut -c1-70; done
timeout -k 10 400 python -m pytest tests/test_gpu_rda.py -x -q > $O/rda_tests.log 2>&1; echo "rda tests rc $?"; tail -3 $O/rda_tests.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_videosar -- python3 $R/tools/bench_videosar.py > $R/$O/videosar_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find $O/prof_videosar -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/videosar_kernel_stats.csv && head -10 $f | cut -c1-140
rm -rf $O/prof_videosar
