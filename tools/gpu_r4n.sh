#!/bin/bash
set -o pipefail
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 400 python3 tools/bench_pipeline.py 16384 60 > $O/pipeline.log 2>&1 || { echo FAILED; tail -20 $O/pipeline.log; exit 1; }
cat $O/pipeline.log
timeout -k 10 200 python3 tools/bench_native.py > $O/native_lanes.log 2>&1; cat $O/native_lanes.log
for p in 0 1; do SARX_CONV_PLANES=$p timeout -k 10 200 python3 tools/bench_rda.py 13200 7200 10 1 >> $O/rda.log 2>&1; done
timeout -k 10 200 python3 tools/bench_rda.py 13200 7200 10 2 >> $O/rda.log 2>&1; cat $O/rda.log
SARX_CONV_PLANES=1 timeout -k 10 300 python -m pytest tests/test_gpu_rda.py -x -q -m gpu 2>&1 | tail -2
