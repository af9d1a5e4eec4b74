#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 600 python -m pytest tests/test_gpu_anysize.py -q -m gpu -x -k "prime_factor or native" > gpurun_out/t_g.log 2>&1; echo "rc $?"; tail -5 gpurun_out/t_g.log
for I in 13 24; do echo "SARX_PFA_IMPL=$I"; SARX_PFA_IMPL=$I timeout -k 10 300 python tools/bench_native.py 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_native4 -- python3 $R/tools/bench_native.py > $R/gpurun_out/prof_native4.log 2>&1; echo "rc $?"
cd $R; f=$(find gpurun_out/prof_native4 -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-150
