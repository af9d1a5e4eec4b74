#!/bin/bash
# HBM traffic per launch from rocprofv3 PMC counters, one counter per run (gfx950: FETCH_SIZE and
# WRITE_SIZE do not fit one pass).  usage (GPU box, repo root): tools/pmc_traffic.sh TAG PASS [SIZE]
R=$GRAFT_REPO_ROOT; TAG=$1; P=$2; N=${3:-16384}; NRG=${4:-}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/traffic_${TAG}_$C -- python3 $R/tools/run_pass.py $P $N 3 $NRG > $R/gpurun_out/traffic_${TAG}_$C.log 2>&1
done
cd $R
