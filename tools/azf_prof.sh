# per-kernel durations of the 16384^2 frame with the XCD-scheduled azimuth launches, for a few scheduler settings
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$R
for cfg in "$@"; do
  set -- $(echo $cfg | tr ':' ' ')
  export SARX_AZF_WGS=$1 SARX_AZF_CHUNK=$2 SARX_AZF_LOOK=$3
  rm -rf /tmp/azf_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/azf_prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-batch > /tmp/azf_prof.log 2>&1 || { tail -5 /tmp/azf_prof.log; exit 1; }
  echo "== wgs=$1 chunk=$2 look=$3"
  python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/azf_prof/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 2: print(f"  {r['Name'][:90]:90s} {r['Calls']:>4s} {float(r['AverageNs'])/1e6:.3f} ms")
PY
done
