#!/bin/bash
# round 3, call w: whole GPU suite with the two-workgroup Rader and range_mixed forms as defaults; native frame, ABBA against the old forms
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3w; O=gpurun_out/r3w
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "tests rc $?"; tail -3 $O/gputests.log
for P in 0 1 1 0; do
  echo "== two-workgroup forms $P"
  SARX_RADER_TWO=$P SARX_MIXED_PLANES=$P timeout -k 10 300 python tools/bench_native.py || exit 1
  SARX_RADER_TWO=$P SARX_MIXED_PLANES=$P timeout -k 10 120 python3 tools/bench_twochannel.py native 10 fused || exit 1
done
timeout -k 10 120 python3 tools/bench_twochannel.py native 10 facade --json $O/native_twochannel_facade.json || exit 1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_native -- python3 $R/tools/bench_native.py > $R/$O/native_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find $O/prof_native -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/native_kernel_stats.csv && head -8 $f
rm -rf $O/prof_native
