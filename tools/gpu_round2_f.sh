#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_anysize.py tests/test_gpu_batch64.py -q -m gpu -x > gpurun_out/t_f.log 2>&1; echo "rc $?"; tail -25 gpurun_out/t_f.log
echo "== native"; timeout -k 10 300 python tools/bench_native.py > gpurun_out/native3.log 2>&1; echo "rc $?"; cat gpurun_out/native3.log
SARX_AZ_PFA=0 timeout -k 10 300 python tools/bench_native.py 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_native3 -- python3 $R/tools/bench_native.py > $R/gpurun_out/prof_native3.log 2>&1; echo "rc $?"
cd $R; f=$(find gpurun_out/prof_native3 -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-160
timeout -k 10 300 python tools/bench_batch64.py 2>&1 | tail -1
