#!/bin/bash
set -o pipefail
O=gpurun_out/r4r; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_batch64.py tests/test_gpu_benchsize.py -x -q -m gpu -k "c3_scene or two_frames_in_flight or different_plan" --durations=5 > $O/tests.log 2>&1 || { echo FAILED; tail -40 $O/tests.log; exit 1; }
tail -12 $O/tests.log
