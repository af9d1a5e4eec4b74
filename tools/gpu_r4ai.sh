#!/bin/bash
set -o pipefail
O=gpurun_out/r4ai; mkdir -p $O
echo "== box $(date +%H%M%S)" | tee -a $O/probe_boxes.log
timeout -k 10 300 python3 tools/probe_lanes.py 16384 40 2>&1 | tee -a $O/probe_boxes.log
