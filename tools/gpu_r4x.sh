#!/bin/bash
set -o pipefail
O=gpurun_out/r4x; mkdir -p $O
timeout -k 10 300 python3 tools/bench_twochannel_lanes.py 8192 20 2>&1 | tee $O/twochannel_lanes_8192.log
timeout -k 10 300 python3 tools/bench_twochannel_lanes.py 16384 10 2>&1 | tee $O/twochannel_lanes_16384.log
timeout -k 10 300 python3 tools/bench_twochannel_lanes.py 4096 40 2>&1 | tee $O/twochannel_lanes_4096.log
