#!/bin/bash
# round 3, call j: tests + bench at the final range_wp defaults (prefetch 6, layout 1), RDA profile
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3j; O=gpurun_out/r3j
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "tests rc $?"; tail -4 $O/gputests.log
timeout -k 10 900 python bench.py --steps 20 --warmup 3 --stack all --passes > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; grep pass $O/bench.err
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-batch > $R/$O/bench_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -12 $f | cut -c1-130
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_videosar -- python3 $R/tools/bench_videosar.py > $R/$O/videosar_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find $O/prof_videosar -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/videosar_kernel_stats.csv && head -10 $f | cut -c1-130
timeout -k 10 100 tools/rgbench_def.bin > $O/rgbench_default.log 2>&1; echo "rgbench rc $?"; grep "wp  \|v2  \|wl  " $O/rgbench_default.log | tail -10 | cut -c1-90
for P in 12 13; do timeout -k 10 200 bash tools/pmc.sh r3_$P $P 16384; timeout -k 10 200 bash tools/pmc_traffic.sh r3_$P $P 16384; done
python3 tools/pmc_summary.py $O/pmc_wp_kernels.json gpurun_out/pmc_r3_12_A gpurun_out/pmc_r3_12_B gpurun_out/traffic_r3_12_FETCH_SIZE gpurun_out/traffic_r3_12_WRITE_SIZE gpurun_out/pmc_r3_13_A gpurun_out/pmc_r3_13_B gpurun_out/traffic_r3_13_FETCH_SIZE gpurun_out/traffic_r3_13_WRITE_SIZE > /dev/null; echo "pmc summary rc $?"
rm -rf gpurun_out/pmc_r3_* gpurun_out/traffic_r3_* $O/prof_bench $O/prof_videosar
