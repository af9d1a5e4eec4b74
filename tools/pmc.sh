#!/bin/bash
# usage: tools/pmc.sh TAG PASS SIZE   (run on the GPU box from the repo root) -> gpurun_out/pmc_TAG_{A,B}/
R=$GRAFT_REPO_ROOT; TAG=$1; P=$2; N=${3:-16384}; NRG=${4:-}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_${TAG}_A -- python3 $R/tools/run_pass.py $P $N 3 $NRG > $R/gpurun_out/pmc_${TAG}_A.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc_${TAG}_B -- python3 $R/tools/run_pass.py $P $N 3 $NRG > $R/gpurun_out/pmc_${TAG}_B.log 2>&1
cd $R
