#!/bin/bash
set -o pipefail
O=gpurun_out/r4am; mkdir -p $O
timeout -k 10 300 python3 tools/az_pairs.py 16384 12 2>&1 | tee $O/az_pairs.log
