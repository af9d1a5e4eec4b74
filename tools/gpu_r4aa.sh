#!/bin/bash
set -o pipefail
O=gpurun_out/r4aa; mkdir -p $O
for nf in 0 2 4 6 8 0 4; do
  for rep in 1 2 3; do
    echo -n "delay fills $nf run $rep: " | tee -a $O/head_start.log
    TWO_STREAMS_DELAY_FILLS=$nf TWO_STREAMS_K=2 SARX_RANGE_CUS=192 timeout -k 10 200 python3 tools/bench_two_streams.py 16384 100 lanes 2>&1 | grep "2 frame" | awk '{printf "%s ", $7}' | tee -a $O/head_start.log
    echo | tee -a $O/head_start.log
  done
done
