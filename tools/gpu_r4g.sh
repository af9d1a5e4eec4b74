#!/bin/bash
set -o pipefail
O=gpurun_out/r4g; mkdir -p $O
for cus in 0 64 96 128 160 192; do
  echo "== SARX_RANGE_CUS=$cus" | tee -a $O/range_cus_sweep.log
  SARX_RANGE_CUS=$cus timeout -k 10 300 python3 tools/bench_two_streams.py 16384 60 lanes 2>&1 | grep -v "1 frame" | tee -a $O/range_cus_sweep.log || { echo FAILED; exit 1; }
done
