#!/bin/bash
# the three example scripts at their native sizes (wall times of whole Python runs), round 4 library
set -o pipefail
O=gpurun_out/r4ac; mkdir -p $O /tmp/ex
for i in 1 2; do timeout -k 10 300 python3 examples/sar_ati_dcpa_csa_gpu.py --out /tmp/ex/two.npz 2>&1 | grep -v "^\[" | tee -a $O/examples.log; done
for i in 1 2; do timeout -k 10 300 python3 examples/sar_satellite_rda_gpu.py --out /tmp/ex/sat.npz 2>&1 | grep -v "^\[" | tee -a $O/examples.log; done
timeout -k 10 400 python3 examples/sar_batch_gpu.py --headings 0 --outdir /tmp/ex/bo 2>&1 | grep -v "^\[" | tee -a $O/examples.log
