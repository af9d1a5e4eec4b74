#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
echo "== dfttest"; timeout -k 10 120 tools/dfttest.bin > gpurun_out/dfttest.log 2>&1; echo "rc $?"; tail -4 gpurun_out/dfttest.log
echo "== anysize + rda tests"; timeout -k 10 900 python -m pytest tests/test_gpu_anysize.py tests/test_gpu_rda.py -q -m gpu > gpurun_out/t_any.log 2>&1; echo "rc $?"; tail -15 gpurun_out/t_any.log
echo "== native"; timeout -k 10 300 python tools/bench_native.py > gpurun_out/native.log 2>&1; echo "rc $?"; cat gpurun_out/native.log
SARX_RANGE_MIXED=0 timeout -k 10 300 python tools/bench_native.py > gpurun_out/native_chirpz.log 2>&1; echo "rc $?"; cat gpurun_out/native_chirpz.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_native -- python3 $R/tools/bench_native.py > $R/gpurun_out/prof_native.log 2>&1; echo "rc $?"
cd $R; find gpurun_out/prof_native -name "*kernel_stats.csv" | head -2
