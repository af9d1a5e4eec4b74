#!/bin/bash
# round 3, call ab: HBM traffic of the planes range kernel after the spill fix
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3ab
rm -rf gpurun_out/traffic_nat2_*
bash tools/pmc_traffic.sh nat2_rg 23 7199 13200
python3 tools/pmc_summary.py gpurun_out/r3ab/pmc_planes_traffic.json gpurun_out/traffic_nat2_rg_FETCH_SIZE gpurun_out/traffic_nat2_rg_WRITE_SIZE | grep -E "hbm_bytes|FETCH|WRITE|Name|range_mixed"
rm -rf gpurun_out/traffic_nat2_*
