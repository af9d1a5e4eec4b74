#!/bin/bash
# round 3, call v: thread -> butterfly rotation of the second workgroup per CU in the two-workgroup form of the 13200-sample fused range launch
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3v
export SARX_MIXED_PLANES=1
SARX_MIXED_ROT=64 timeout -k 10 300 python -m pytest tests/test_gpu_anysize.py -x -q > gpurun_out/r3v/tests.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r3v/tests.log
for S in 0 64 128 192 320 0 64; do
  echo "== rot $S"; SARX_MIXED_ROT=$S timeout -k 10 120 python3 tools/run_pass.py 23 7199 30 13200 || exit 1
done
SARX_MIXED_PLANES=0 timeout -k 10 120 python3 tools/run_pass.py 23 7199 30 13200
