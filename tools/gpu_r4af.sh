#!/bin/bash
set -o pipefail
O=gpurun_out/r4af; mkdir -p $O /tmp/ex
timeout -k 10 600 python -m pytest tests/test_gpu_echo.py tests/test_gpu_tdbp.py tests/test_gpu_noise.py tests/test_gpu_example.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do timeout -k 10 400 python3 examples/sar_batch_gpu.py --headings 0 --outdir /tmp/ex/bo 2>&1 | grep -v "^\[" | tee -a $O/videosar_example.log; done
timeout -k 10 300 python3 tools/bench_videosar.py 2>&1 | tee $O/bench_videosar.log
timeout -k 10 300 python3 examples/sar_ati_dcpa_csa_gpu.py --out /tmp/ex/two.npz 2>&1 | grep "echo synthesis" | tee -a $O/videosar_example.log
