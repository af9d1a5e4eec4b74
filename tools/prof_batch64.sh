cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python3 $R/tools/bench_batch64.py --frames 16 > /tmp/pb.log 2>&1 || { tail -5 /tmp/pb.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/pb/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 1: print(f"  {r['Name'][:80]:80s} {r['Calls']:>4s} {float(r['AverageNs'])/1e6:.3f} ms")
PY
