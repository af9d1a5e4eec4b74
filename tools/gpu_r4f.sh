#!/bin/bash
set -o pipefail
O=gpurun_out/r4f; mkdir -p $O
timeout -k 10 400 python3 tools/bench_two_streams.py 16384 40 > $O/two_streams_modes.log 2>&1 || { echo FAILED; tail -20 $O/two_streams_modes.log; exit 1; }
cat $O/two_streams_modes.log
