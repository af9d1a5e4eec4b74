#!/bin/bash
# evidence for every number quoted in DESIGN.md / README.md: bench logs + rocprofv3 --kernel-trace --stats summaries
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/ev
prof() {  # tag, program args...
  local tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ev/prof_$tag -- python3 "$@" > $R/gpurun_out/ev/${tag}_profiled.log 2>&1 )
  echo "prof $tag rc $?"
  f=$(find gpurun_out/ev/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/ev/${tag}_kernel_stats.csv
}
echo "== plain runs"
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --stack both > gpurun_out/ev/bench.json 2> gpurun_out/ev/bench.err; echo "bench rc $?"
for m in fused masked two-pass; do timeout -k 10 300 python tools/bench_twochannel.py 8192 10 $m >> gpurun_out/ev/twochannel.log 2>&1; echo "rc $?"; timeout -k 10 300 python tools/bench_twochannel.py native 10 $m >> gpurun_out/ev/native_twochannel.log 2>&1; echo "rc $?"; done; cat gpurun_out/ev/twochannel.log gpurun_out/ev/native_twochannel.log
timeout -k 10 300 python tools/bench_native.py > gpurun_out/ev/native.log 2>&1; echo "rc $?"; cat gpurun_out/ev/native.log
timeout -k 10 300 python tools/bench_echo.py > gpurun_out/ev/echo.log 2>&1; echo "rc $?"; cat gpurun_out/ev/echo.log
timeout -k 10 300 python tools/bench_videosar.py > gpurun_out/ev/videosar.log 2>&1; echo "rc $?"; cat gpurun_out/ev/videosar.log
timeout -k 10 300 python tools/bench_batch64.py > gpurun_out/ev/batch64.log 2>&1; echo "rc $?"; cat gpurun_out/ev/batch64.log
timeout -k 10 300 python tools/bench_batch64.py --stack magnitude > gpurun_out/ev/batch64_mag.log 2>&1; echo "rc $?"; cat gpurun_out/ev/batch64_mag.log
for s in 4096 8192; do timeout -k 10 300 python bench.py --size $s --steps 40 --warmup 5 --no-cpu --no-batch > gpurun_out/ev/bench_$s.json 2>/dev/null; echo "rc $?"; done
timeout -k 10 300 tools/membench.bin > gpurun_out/ev/membench.log 2>&1; echo "membench rc $?"
echo "== profiled runs"
prof bench $R/bench.py --steps 20 --warmup 3 --no-cpu --no-batch
prof native $R/tools/bench_native.py
prof videosar $R/tools/bench_videosar.py
prof twochannel $R/tools/bench_twochannel.py 8192 6
prof batch64 $R/tools/bench_batch64.py --frames 16
ls gpurun_out/ev | head -40
