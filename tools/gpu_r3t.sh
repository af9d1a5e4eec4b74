#!/bin/bash
# round 3, call t: counters of the 13200-sample fused range launch, one workgroup per CU vs re / im planes with two
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
SARX_MIXED_PLANES=0 bash tools/pmc.sh mixed_p0 23 7199 13200 && SARX_MIXED_PLANES=1 bash tools/pmc.sh mixed_p1 23 7199 13200
python3 tools/pmc_report.py gpurun_out/pmc_mixed_p0_A gpurun_out/pmc_mixed_p0_B gpurun_out/pmc_mixed_p1_A gpurun_out/pmc_mixed_p1_B
