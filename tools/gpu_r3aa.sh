#!/bin/bash
# round 3, call aa: Phi_1 epilogues with their row constants through the scalar cache BEFORE the stores (az_tile_kernel W=32, pfa_dft23_kernel), ABBA against HEAD's library
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3aa; O=gpurun_out/r3aa
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchsize.py tests/test_gpu_anysize.py tests/test_gpu_batch64.py -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
for L in head new new head; do
  echo "== $L"
  if [ $L = head ]; then export SARX_LIB=$R/build/abl/libsarx_head.so; else unset SARX_LIB; fi
  timeout -k 10 120 python3 tools/run_pass.py 1 16384 20 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 0 16384 20 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 1 8192 40 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 0 8192 40 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 1 4096 80 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 1 7199 30 13200 || exit 1
done
