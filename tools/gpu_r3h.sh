#!/bin/bash
# round 3, call h: evidence for the numbers quoted in DESIGN.md / README.md at this commit
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3h; O=gpurun_out/r3h
prof() {  # tag, program args...
  local tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$tag -- python3 "$@" > $R/$O/${tag}_profiled.log 2>&1 )
  echo "prof $tag rc $?"
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
  rm -rf $O/prof_$tag
}
timeout -k 10 900 python bench.py --steps 20 --warmup 3 --stack all > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --size 4096 --steps 40 --warmup 5 --no-batch > $O/bench_4096.json 2>/dev/null; echo "bench 4096 rc $?"
timeout -k 10 400 python bench.py --size 8192 --steps 40 --warmup 5 --no-batch --config3 > $O/bench_8192.json 2>/dev/null; echo "bench 8192 rc $?"   # + config 3 with its CPU baseline
for m in fused facade masked two-pass; do timeout -k 10 400 python tools/bench_twochannel.py 8192 10 $m --json $O/twochannel_$m.json >> $O/twochannel.log 2>&1; echo "tc $m rc $?"; done
for m in fused facade; do timeout -k 10 300 python tools/bench_twochannel.py native 10 $m --json $O/native_twochannel_$m.json >> $O/native_twochannel.log 2>&1; echo "ntc $m rc $?"; done
cat $O/twochannel.log $O/native_twochannel.log | grep -v "^{"
timeout -k 10 300 python tools/bench_batch64.py > $O/batch64.log 2>&1; echo "rc $?"; cat $O/batch64.log
timeout -k 10 300 python tools/bench_batch64.py --stack products > $O/batch64_products.log 2>&1; echo "rc $?"; cat $O/batch64_products.log
timeout -k 10 400 python tools/bench_batch64.py --scene c3 > $O/batch64_c3.log 2>&1; echo "rc $?"; cat $O/batch64_c3.log
timeout -k 10 300 python tools/bench_echo.py > $O/echo.log 2>&1; echo "rc $?"; cat $O/echo.log
timeout -k 10 300 python tools/bench_native.py > $O/native.log 2>&1; echo "rc $?"; cat $O/native.log
timeout -k 10 100 tools/membench.bin > $O/membench.log 2>&1; echo "membench rc $?"
prof bench $R/bench.py --steps 20 --warmup 3 --no-cpu --no-batch
prof twochannel $R/tools/bench_twochannel.py 8192 6
prof batch64 $R/tools/bench_batch64.py --frames 16
timeout -k 10 500 bash tools/pmc_compute.sh > $O/pmc_compute.log 2>&1; echo "pmc compute rc $?"; tail -3 $O/pmc_compute.log
cp gpurun_out/pmc_compute/*_summary.json $O/ 2>/dev/null; rm -rf gpurun_out/pmc_compute
ls $O
