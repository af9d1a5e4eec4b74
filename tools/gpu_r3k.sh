#!/bin/bash
# round 3, call k: the 8192-sample fused range kernel on the wave-private structure (range_wp8_fused_kernel) against the Stockham kernel
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3k; O=gpurun_out/r3k
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchsize.py tests/test_gpu_batch64.py -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
for i in 1 2; do
  for impl in 1 0 0 1; do echo "SARX_RANGE_IMPL=$impl"; SARX_RANGE_IMPL=$impl timeout -k 10 100 python tools/run_pass.py 23 8192 30; done
  for impl in 0 1 1 0; do echo "SARX_RANGE_IMPL=$impl"; SARX_RANGE_IMPL=$impl timeout -k 10 200 python tools/bench_twochannel.py 8192 10 fused | head -1; done
done 2>&1 | tee $O/ab.log
for impl in 0 1 1 0; do echo "SARX_RANGE_IMPL=$impl"; SARX_RANGE_IMPL=$impl timeout -k 10 300 python tools/bench_batch64.py; done 2>&1 | tee -a $O/ab.log
