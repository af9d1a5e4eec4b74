#!/bin/bash
set -o pipefail
O=gpurun_out/r4ag; mkdir -p $O /tmp/ex
timeout -k 10 600 python -m pytest tests/test_gpu_noise.py tests/test_gpu_tdbp.py tests/test_gpu_echo.py tests/test_gpu_example.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do timeout -k 10 400 python3 examples/sar_batch_gpu.py --headings 0 --outdir /tmp/ex/bo 2>&1 | grep -v "^\[" | tee -a $O/videosar_example.log; done
timeout -k 10 400 python3 examples/sar_batch_gpu.py --outdir /tmp/ex/bo4 2>&1 | grep -v "^\[" | tee -a $O/videosar_example_all_headings.log
