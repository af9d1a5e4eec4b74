#!/bin/bash
set -o pipefail
O=gpurun_out/r4ad; mkdir -p $O /tmp/ex
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "page_locked or two_channel" 2>&1 | tail -2
for i in 1 2; do timeout -k 10 300 python3 examples/sar_ati_dcpa_csa_gpu.py --out /tmp/ex/two.npz 2>&1 | grep -v "^\[" | tee -a $O/examples.log; done
for i in 1 2; do timeout -k 10 300 python3 examples/sar_satellite_rda_gpu.py --out /tmp/ex/sat.npz 2>&1 | grep -v "^\[" | tee -a $O/examples.log; done
timeout -k 10 400 python3 examples/sar_batch_gpu.py --headings 0 --outdir /tmp/ex/bo 2>&1 | grep -v "^\[" | tee -a $O/examples.log
timeout -k 10 300 python3 tools/bench_hostpath.py 16384 --json $O/hostpath_16384.json 2>&1 | grep "call\|parts\|out=" | tee $O/hostpath_16384.log
timeout -k 10 300 python3 tools/bench_hostpath.py 8192 --json $O/hostpath_8192.json 2>&1 | grep "call\|parts\|out=" | tee $O/hostpath_8192.log
