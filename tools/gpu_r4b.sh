#!/bin/bash
# round 4, call b: Infinity-Cache-resident vs HBM-streaming timing of each kernel of the middle of the chain; per-kernel trace of slab mode
set -o pipefail
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 300 python3 tools/mall_resident.py > $O/mall_resident.log 2>&1 || { echo FAILED; tail -20 $O/mall_resident.log; exit 1; }
cat $O/mall_resident.log
cd /tmp && export TMPDIR=/tmp
for cfg in "0 1" "64 1" "64 4" "32 8"; do
  set -- $cfg
  SARX_SLAB_MIB=$1 SARX_SLAB_STREAMS=$2 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_slab$1_s$2 -o p -- python3 $GRAFT_REPO_ROOT/tools/run_pass.py 0 16384 10 > $GRAFT_REPO_ROOT/$O/prof_slab$1_s$2.log 2>&1 || { echo "prof $cfg FAILED"; tail -5 $GRAFT_REPO_ROOT/$O/prof_slab$1_s$2.log; exit 1; }
  tail -1 $GRAFT_REPO_ROOT/$O/prof_slab$1_s$2.log
  f=$(find $GRAFT_REPO_ROOT/$O/prof_slab$1_s$2 -name "*kernel_stats.csv" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/$O/slab$1_s$2_kernel_stats.csv
  rm -rf $GRAFT_REPO_ROOT/$O/prof_slab$1_s$2
  head -8 $GRAFT_REPO_ROOT/$O/slab$1_s$2_kernel_stats.csv | cut -c1-200
done
