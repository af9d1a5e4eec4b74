#!/bin/bash
set -o pipefail
O=gpurun_out/r4k; mkdir -p $O
for cfg in "0 1 192" "32 1 192" "64 1 192" "128 1 192" "64 2 192" "64 1 0" "32 2 0"; do
  set -- $cfg
  echo "== SLAB_MIB=$1 SLAB_STREAMS=$2 RANGE_CUS=$3" | tee -a $O/slab_lanes.log
  SARX_SLAB_MIB=$1 SARX_SLAB_STREAMS=$2 SARX_RANGE_CUS=$3 timeout -k 10 300 python3 tools/bench_two_streams.py 16384 40 lanes 2>&1 | grep "2 frame\|3 frame" | tee -a $O/slab_lanes.log || { echo FAILED; exit 1; }
done
timeout -k 10 300 ./tools/mallpipe.bin > $O/mallpipe.log 2>&1 || { echo "mallpipe FAILED"; tail -20 $O/mallpipe.log; exit 1; }
cat $O/mallpipe.log
