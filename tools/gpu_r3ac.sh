#!/bin/bash
# round 3, call ac: prime-factor (twiddle-free) split of the coprime composite butterflies (24 = 3*8, 22 = 11*2, 15 = 3*5) in fft_mixed.hpp, ABBA against HEAD's library
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3ac; O=gpurun_out/r3ac
./tools/dfttest.bin > $O/dfttest.log 2>&1; echo "dfttest rc $?"; tail -3 $O/dfttest.log
timeout -k 10 900 python -m pytest tests/test_gpu_anysize.py tests/test_gpu_rda.py -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
for L in head new new head; do
  echo "== $L"
  if [ $L = head ]; then export SARX_LIB=$R/build/abl/libsarx_head.so; else unset SARX_LIB; fi
  timeout -k 10 120 python3 tools/run_pass.py 23 7199 30 13200 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 1 7199 30 13200 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 0 7199 30 13200 || exit 1
done
