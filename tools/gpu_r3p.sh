#!/bin/bash
# round 3, call p: generic range kernel on persistent workgroups (lines of 4096+ samples) vs one workgroup per line, ABBA
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3p; O=gpurun_out/r3p
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchsize.py tests/test_gpu_batch64.py tests/test_gpu_rda.py -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
for P in 0 1 1 0; do
  export SARX_RANGE_PERSIST=$P
  echo "== persist $P"
  for pid in 23 2 3; do timeout -k 10 120 python3 tools/run_pass.py $pid 8192 40 || exit 1; done
  timeout -k 10 120 python3 tools/run_pass.py 23 4096 80 || exit 1
  timeout -k 10 120 python3 tools/run_pass.py 0 8192 30 || exit 1
  timeout -k 10 120 python3 tools/bench_twochannel.py 8192 20 fused || exit 1
  timeout -k 10 120 python3 tools/bench_twochannel.py 4096 40 fused || exit 1
  timeout -k 10 200 python3 tools/bench_batch64.py 2>&1 | tail -2 || exit 1
done
