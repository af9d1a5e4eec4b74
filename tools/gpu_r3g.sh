#!/bin/bash
# round 3, call g: wl interleave A/B, native CSA with the prefetching mixed-radix kernel, bench passes, RDA profile, PMC of the range kernels
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3g
for v in wlI0 wlI1 wlI0 wlI1; do timeout -k 10 100 tools/rgbench_$v.bin >> gpurun_out/r3g/rgbench_$v.log 2>&1; echo "$v rc $?"; done
grep -h "wl  fused\|wp  FFT\|wp  IFFT" gpurun_out/r3g/rgbench_wlI0.log | tail -12; echo ---; grep -h "wl  fused\|wp  FFT\|wp  IFFT" gpurun_out/r3g/rgbench_wlI1.log | tail -12
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_anysize.py tests/test_gpu_rda.py tests/test_gpu_benchsize.py -x -q > gpurun_out/r3g/tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r3g/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-batch --passes > gpurun_out/r3g/bench_nocpu.json 2> gpurun_out/r3g/bench_nocpu.err; echo "bench rc $?"; cat gpurun_out/r3g/bench_nocpu.err | grep pass
timeout -k 10 300 python tools/bench_native.py > gpurun_out/r3g/native.log 2>&1; echo "rc $?"; cat gpurun_out/r3g/native.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3g/prof_native -- python3 $R/tools/bench_native.py > $R/gpurun_out/r3g/native_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find gpurun_out/r3g/prof_native -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r3g/native_kernel_stats.csv && head -8 $f
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3g/prof_videosar -- python3 $R/tools/bench_videosar.py > $R/gpurun_out/r3g/videosar_profiled.log 2>&1 ); echo "prof rc $?"
f=$(find gpurun_out/r3g/prof_videosar -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r3g/videosar_kernel_stats.csv && head -10 $f
for P in 12 13; do timeout -k 10 200 bash tools/pmc.sh r3_$P $P 16384; timeout -k 10 200 bash tools/pmc_traffic.sh r3_$P $P 16384; done
python3 tools/pmc_summary.py gpurun_out/r3g/pmc_wp_kernels.json gpurun_out/pmc_r3_12_A gpurun_out/pmc_r3_12_B gpurun_out/traffic_r3_12_FETCH_SIZE gpurun_out/traffic_r3_12_WRITE_SIZE gpurun_out/pmc_r3_13_A gpurun_out/pmc_r3_13_B gpurun_out/traffic_r3_13_FETCH_SIZE gpurun_out/traffic_r3_13_WRITE_SIZE > /dev/null; echo "pmc summary rc $?"
rm -rf gpurun_out/pmc_r3_* gpurun_out/traffic_r3_* gpurun_out/r3g/prof_native gpurun_out/r3g/prof_videosar
