#!/bin/bash
# round 3, call f1: whole GPU test suite, the bench line, its kernel profile, counters of the range kernels
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3f
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3f/gputests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r3f/gputests.log
timeout -k 10 900 python bench.py --steps 20 --warmup 3 --stack all > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err; echo "bench rc $?"; head -c 1500 gpurun_out/r3f/bench.json; echo
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f/prof_bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-batch > $R/gpurun_out/r3f/bench_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find gpurun_out/r3f/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r3f/bench_kernel_stats.csv && head -12 $f
for P in 12 13 23; do timeout -k 10 200 bash tools/pmc.sh r3_$P $P 16384; timeout -k 10 200 bash tools/pmc_traffic.sh r3_$P $P 16384; done
python3 tools/pmc_summary.py gpurun_out/r3f/pmc_range_kernels.json gpurun_out/pmc_r3_12_A gpurun_out/pmc_r3_12_B gpurun_out/traffic_r3_12_FETCH_SIZE gpurun_out/traffic_r3_12_WRITE_SIZE gpurun_out/pmc_r3_13_A gpurun_out/pmc_r3_13_B gpurun_out/traffic_r3_13_FETCH_SIZE gpurun_out/traffic_r3_13_WRITE_SIZE gpurun_out/pmc_r3_23_A gpurun_out/pmc_r3_23_B gpurun_out/traffic_r3_23_FETCH_SIZE gpurun_out/traffic_r3_23_WRITE_SIZE > /dev/null; echo "pmc summary rc $?"
rm -rf gpurun_out/pmc_r3_* gpurun_out/traffic_r3_* gpurun_out/r3f/prof_bench
