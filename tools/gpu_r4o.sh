#!/bin/bash
set -o pipefail
O=gpurun_out/r4o; mkdir -p $O
for prio in none 1 -1 2 -2; do
  for cus in 192 0; do
    echo "== SARX_LANE_PRIO=$prio RANGE_CUS=$cus" | tee -a $O/lane_prio.log
    if [ $prio = none ]; then SARX_RANGE_CUS=$cus timeout -k 10 200 python3 tools/bench_two_streams.py 16384 60 lanes 2>&1 | grep "2 frame" | tee -a $O/lane_prio.log
    else SARX_LANE_PRIO=$prio SARX_RANGE_CUS=$cus timeout -k 10 200 python3 tools/bench_two_streams.py 16384 60 lanes 2>&1 | grep "2 frame" | tee -a $O/lane_prio.log; fi
  done
done
