#!/bin/bash
set -o pipefail
O=gpurun_out/r4q; mkdir -p $O
for lib in "" build/abl/libsarx_mixw5.so "" build/abl/libsarx_mixw5.so; do
  echo "== SARX_LIB=${lib:-default}" | tee -a $O/mixw5.log
  if [ -z "$lib" ]; then timeout -k 10 200 python3 tools/run_pass.py 23 7199 20 13200 2>&1 | tee -a $O/mixw5.log; timeout -k 10 200 python3 tools/bench_native.py 2>&1 | head -2 | tee -a $O/mixw5.log
  else SARX_LIB=$PWD/$lib timeout -k 10 200 python3 tools/run_pass.py 23 7199 20 13200 2>&1 | tee -a $O/mixw5.log; SARX_LIB=$PWD/$lib timeout -k 10 200 python3 tools/bench_native.py 2>&1 | head -2 | tee -a $O/mixw5.log; fi
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lanes" 2>&1 | tail -2
