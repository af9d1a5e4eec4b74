#!/bin/bash
set -o pipefail
O=gpurun_out/r4y; mkdir -p $O
for tpw in 1 2 4 8 16 1; do
  echo "== SARX_AZ_TPW=$tpw" | tee -a $O/az_tpw.log
  SARX_AZ_TPW=$tpw timeout -k 10 120 python3 tools/run_pass.py 1 16384 10 2>&1 | tee -a $O/az_tpw.log
  SARX_AZ_TPW=$tpw timeout -k 10 120 python3 tools/run_pass.py 4 16384 10 2>&1 | tee -a $O/az_tpw.log
  SARX_AZ_TPW=$tpw TWO_STREAMS_K=1,2 SARX_RANGE_CUS=192 timeout -k 10 200 python3 tools/bench_two_streams.py 16384 40 lanes 2>&1 | head -2 | tee -a $O/az_tpw.log
done
SARX_AZ_TPW=4 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
