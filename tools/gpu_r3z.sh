#!/bin/bash
# round 3, call z: 23-point forward launch with its row constants through the scalar cache before the stores; planes kernel without the hoisted spills
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3z; O=gpurun_out/r3z
timeout -k 10 600 python -m pytest tests/test_gpu_anysize.py -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
for i in 1 2; do
timeout -k 10 120 python3 tools/run_pass.py 1 7199 30 13200 || exit 1
timeout -k 10 120 python3 tools/run_pass.py 23 7199 30 13200 || exit 1
timeout -k 10 300 python tools/bench_native.py || exit 1
done
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_native -- python3 $R/tools/bench_native.py > $R/$O/native_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find $O/prof_native -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/native_kernel_stats.csv && head -8 $f
rm -rf $O/prof_native
