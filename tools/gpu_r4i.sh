#!/bin/bash
set -o pipefail
O=gpurun_out/r4i; mkdir -p $O
for cus in 192 0 160; do
  echo "== SARX_RANGE_CUS=$cus" | tee -a $O/stagger.log
  SARX_RANGE_CUS=$cus timeout -k 10 300 python3 tools/bench_two_streams.py 16384 60 stag mark 2>&1 | grep "2 frame\|3 frame" | tee -a $O/stagger.log || { echo FAILED; exit 1; }
done
