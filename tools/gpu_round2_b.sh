#!/bin/bash
# second GPU call of round 2: whole GPU suite, SQ counters + HBM traffic of the range kernels, 2-rank rehearsal
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
echo "== gpu suite"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "rc $?"; tail -3 gpurun_out/t_all.log
echo "== 2-rank rehearsal on one GPU"; timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --batch-frames 8 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err; echo "rc $?"; tail -c 300 gpurun_out/bench_2rank.json
echo "== counters"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
for P in 23 2; do
  bash $R/tools/pmc.sh r02_$P $P 16384
  bash $R/tools/pmc_traffic.sh r02_$P $P 16384
done
cd $R
python3 tools/pmc_report.py gpurun_out/pmc_r02_23_A gpurun_out/pmc_r02_23_B gpurun_out/pmc_r02_2_A gpurun_out/pmc_r02_2_B gpurun_out/traffic_r02_23_FETCH_SIZE gpurun_out/traffic_r02_23_WRITE_SIZE gpurun_out/traffic_r02_2_FETCH_SIZE gpurun_out/traffic_r02_2_WRITE_SIZE > gpurun_out/pmc_r02_report.txt 2>&1
tail -30 gpurun_out/pmc_r02_report.txt
