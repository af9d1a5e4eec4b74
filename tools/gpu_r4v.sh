#!/bin/bash
set -o pipefail
O=gpurun_out/r4v; mkdir -p $O
for sp in 1 2 3 4 2 1; do
  echo "== SARX_AZ_SPLIT=$sp" | tee -a $O/az_split.log
  SARX_AZ_SPLIT=$sp timeout -k 10 120 python3 tools/run_pass.py 1 16384 10 2>&1 | tee -a $O/az_split.log
  SARX_AZ_SPLIT=$sp timeout -k 10 120 python3 tools/run_pass.py 4 16384 10 2>&1 | tee -a $O/az_split.log
  SARX_AZ_SPLIT=$sp TWO_STREAMS_K=1,2 SARX_RANGE_CUS=192 timeout -k 10 200 python3 tools/bench_two_streams.py 16384 40 lanes 2>&1 | tee -a $O/az_split.log
done
SARX_AZ_SPLIT=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchsize.py -x -q -m gpu 2>&1 | tail -2
