#!/bin/bash
# round 3, call y: counters + HBM traffic of the native frame's launches at HEAD (Rader two-workgroup form, range planes form, 23-point launches);
# the new A/B test alone
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3y
timeout -k 10 400 python -m pytest tests/test_gpu_anysize.py -x -q -k "one_workgroup" > gpurun_out/r3y/test.log 2>&1; echo "test rc $?"; tail -3 gpurun_out/r3y/test.log
rm -rf gpurun_out/pmc_nat_* gpurun_out/traffic_nat_*
bash tools/pmc.sh nat_rg 23 7199 13200 && bash tools/pmc_traffic.sh nat_rg 23 7199 13200 && bash tools/pmc.sh nat_az1 1 7199 13200 && bash tools/pmc_traffic.sh nat_az1 1 7199 13200 && bash tools/pmc.sh nat_az4 4 7199 13200 && bash tools/pmc_traffic.sh nat_az4 4 7199 13200
python3 tools/pmc_summary.py gpurun_out/r3y/pmc_native_kernels.json gpurun_out/pmc_nat_rg_A gpurun_out/pmc_nat_rg_B gpurun_out/traffic_nat_rg_FETCH_SIZE gpurun_out/traffic_nat_rg_WRITE_SIZE gpurun_out/pmc_nat_az1_A gpurun_out/pmc_nat_az1_B gpurun_out/traffic_nat_az1_FETCH_SIZE gpurun_out/traffic_nat_az1_WRITE_SIZE gpurun_out/pmc_nat_az4_A gpurun_out/pmc_nat_az4_B gpurun_out/traffic_nat_az4_FETCH_SIZE gpurun_out/traffic_nat_az4_WRITE_SIZE; echo "summary rc $?"
rm -rf gpurun_out/pmc_nat_* gpurun_out/traffic_nat_*
