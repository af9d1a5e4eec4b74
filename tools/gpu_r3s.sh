#!/bin/bash
# round 3, call s: 13200-sample fused range launch with re / im plane crossings and two workgroups per CU (SARX_MIXED_PLANES=1)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3s; O=gpurun_out/r3s
SARX_MIXED_PLANES=1 SARX_RADER_TWO=1 timeout -k 10 600 python -m pytest tests/test_gpu_anysize.py -x -q > $O/tests.log 2>&1; echo "tests(planes, two) rc $?"; tail -3 $O/tests.log
for P in 0 1 1 0; do
  export SARX_MIXED_PLANES=$P
  echo "== planes $P"
  timeout -k 10 120 python3 tools/run_pass.py 23 7199 30 13200 || exit 1
  SARX_RADER_TWO=1 timeout -k 10 120 python3 tools/run_pass.py 0 7199 30 13200 || exit 1
done
unset SARX_MIXED_PLANES
for P in 0 1 1 0; do
  export SARX_RADER_TWO=$P
  echo "== two $P"
  timeout -k 10 120 python3 tools/run_pass.py 1 7199 30 13200 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 4 7199 30 13200 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 0 7199 30 13200 || exit 1
done
