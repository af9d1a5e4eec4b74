#!/bin/bash
set -o pipefail
O=gpurun_out/r4ah; mkdir -p $O
for i in 1 2 3; do echo "== process $i" | tee -a $O/probe.log; timeout -k 10 300 python3 tools/probe_lanes.py 16384 40 2>&1 | tee -a $O/probe.log; done
