#!/bin/bash
set -o pipefail
O=gpurun_out/r4u; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cus in 192 0; do
  TWO_STREAMS_K=2 SARX_RANGE_CUS=$cus timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$cus -- python3 $R/tools/bench_two_streams.py 16384 24 lanes > $R/$O/run_$cus.log 2>&1 || { echo FAILED; tail $R/$O/run_$cus.log; exit 1; }
  grep "frame(s)" $R/$O/run_$cus.log
  f=$(find /tmp/tr_$cus -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_lanes.py $f > $R/$O/timeline_$cus.log 2>&1; tail -30 $R/$O/timeline_$cus.log
done
