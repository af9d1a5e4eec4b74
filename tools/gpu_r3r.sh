#!/bin/bash
# round 3, call r: Rader-313 launch with two workgroups per CU (SARX_RADER_TWO=1) vs one, parity + ABBA timing
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3r; O=gpurun_out/r3r
SARX_RADER_TWO=1 timeout -k 10 600 python -m pytest tests/test_gpu_anysize.py -x -q > $O/tests_two.log 2>&1; echo "tests(two) rc $?"; tail -3 $O/tests_two.log
for P in 0 1 1 0; do
  export SARX_RADER_TWO=$P
  echo "== two $P"
  timeout -k 10 120 python3 tools/run_pass.py 1 7199 30 13200 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 4 7199 30 13200 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 0 7199 30 13200 || exit 1
done
export SARX_RADER_TWO=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof --output-format csv -o two -- python3 $R/tools/run_pass.py 0 7199 10 13200 > $R/$O/prof.log 2>&1; echo "prof rc $?"
cat $R/$O/prof/*/two_kernel_stats.csv 2>/dev/null || find $R/$O/prof -name "*kernel_stats.csv" -exec cat {} \;
