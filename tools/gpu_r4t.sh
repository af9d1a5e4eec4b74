#!/bin/bash
# round 4, call t (final validation, re-run at the last library): validation + evidence at this commit (whole GPU suite, smoke, bench lines, rocprof summaries, traffic counters)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r4t; mkdir -p $O
prof() {  # tag, program args...
  local tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 "$@" > $R/$O/${tag}_profiled.log 2>&1 )
  echo "prof $tag rc $?"
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
  rm -rf /tmp/prof_$tag
}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { echo "GPU TESTS FAILED"; tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "bench default FAILED"; tail -20 $O/bench_default.err; exit 1; }
timeout -k 10 900 python bench.py --steps 100 --warmup 5 --stack all --passes > $O/bench.json 2> $O/bench.err || { echo "bench FAILED"; tail -20 $O/bench.err; exit 1; }
grep "pass\]" $O/bench.err
timeout -k 10 300 python bench.py --steps 100 --warmup 5 --in-flight 1 --no-cpu --no-batch > $O/bench_inflight1.json 2>/dev/null; echo "bench inflight1 rc $?"
timeout -k 10 300 python bench.py --size 4096 --steps 200 --warmup 5 --no-batch > $O/bench_4096.json 2>/dev/null; echo "bench 4096 rc $?"
timeout -k 10 400 python bench.py --size 8192 --steps 100 --warmup 5 --no-batch --config3 > $O/bench_8192.json 2>/dev/null; echo "bench 8192 rc $?"
python3 - <<PY
import json
for f in ('bench_default','bench','bench_inflight1','bench_4096','bench_8192'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1])
    print(f, round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'fps  roofline', round(d['roofline']['frac'],3), 'shared', round(d.get('roofline_shared',{}).get('frac',0),3),
          'phi2', round(d.get('roofline_rg_fft_phi2_pass',{}).get('frac',0),3), {k:round(v['value'],1) for k,v in d.items() if k.startswith('batch64')}, 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
for m in fused facade; do timeout -k 10 400 python tools/bench_twochannel.py 8192 10 $m --json $O/twochannel_$m.json >> $O/twochannel.log 2>&1; echo "tc $m rc $?"; done
grep -v "^{" $O/twochannel.log
timeout -k 10 300 python tools/bench_native.py > $O/native.log 2>&1; echo "native rc $?"; cat $O/native.log
timeout -k 10 300 python3 tools/bench_hostpath.py 16384 --json $O/hostpath_16384.json > $O/hostpath_16384.log 2>&1; echo "hostpath rc $?"; grep "call [2-5]\|parts" $O/hostpath_16384.log
timeout -k 10 300 python3 tools/bench_hostpath.py 8192 --json $O/hostpath_8192.json > $O/hostpath_8192.log 2>&1; echo "hostpath rc $?"; grep "call [2-5]\|parts" $O/hostpath_8192.log
timeout -k 10 100 tools/membench.bin > $O/membench.log 2>&1; echo "membench rc $?"
prof bench $R/bench.py --steps 20 --warmup 3 --no-cpu --no-batch
prof bench_inflight1 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-batch --in-flight 1
prof batch64 $R/tools/bench_batch64.py --frames 16
for f in bench bench_inflight1 batch64; do echo "== $f"; head -7 $O/${f}_kernel_stats.csv | cut -c1-160; done
bash tools/pmc_traffic.sh r4fused 23 16384 > /dev/null 2>&1; echo "pmc rc $?"
python3 tools/pmc_summary.py $O/pmc_range_fused.json gpurun_out/traffic_r4fused_FETCH_SIZE gpurun_out/traffic_r4fused_WRITE_SIZE > /dev/null 2>&1; echo "pmc summary rc $?"
python3 -c "
import json; d=json.load(open('$O/pmc_range_fused.json'))
for k,v in d['kernels'].items(): print(k[:60], v.get('hbm_bytes_per_launch'), v['dispatch_ms_under_profiler'])"
rm -rf gpurun_out/traffic_r4fused_*
ls $O
