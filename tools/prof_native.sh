# per-kernel durations of the native 7199 x 13200 focus (tools/bench_native.py) under rocprofv3, for each SARX_PFA_PF given
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for pf in "$@"; do
  export SARX_PFA_PF=$pf
  rm -rf /tmp/natprof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/natprof -- python3 $R/tools/bench_native.py > /tmp/natprof.log 2>&1 || { tail -5 /tmp/natprof.log; exit 1; }
  echo "== SARX_PFA_PF=$pf: $(grep 'CSA focus' /tmp/natprof.log)"
  python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/natprof/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 2: print(f"  {r['Name'][:80]:80s} {r['Calls']:>4s} {float(r['AverageNs'])/1e6:.3f} ms")
PY
done
