#!/bin/bash
# One parametrised runner for every call sent to the GPU box (replaces the per-call gpu_r3*/gpu_r4* transcripts).
#
#   gpurun --timeout 900 -- 'bash tools/gpu_call.sh <tag> <step> [args...] [-- <step> [args...]]...'
#
# Output goes to gpurun_out/<tag>/ (merged back by gpurun); every step appends one line to gpurun_out/<tag>/MANIFEST.txt
# (step, arguments, exit code, seconds, files it wrote), which is what profiles/MANIFEST.md is assembled from.
# A failing step stops the call (no further GPU step after a failure or a timeout).
#
# Steps:
#   tests [pytest args]          whole GPU suite (pytest tests -m gpu -x -q ...)            -> gputests.log
#   smoke                        __graft_entry__.smoke()                                    -> smoke.log
#   bench <name> [bench args]    python bench.py ...                                        -> bench_<name>.json / .err
#   prof <name> <script> [args]  rocprofv3 --kernel-trace --stats -- python3 <script> ...   -> <name>_kernel_stats.csv, <name>_profiled.log
#   trace <name> <script> [args] rocprofv3 --kernel-trace -- python3 <script> ...           -> <name>_kernel_trace.csv.gz, <name>_busy.log
#   py <name> <script> [args]    python3 <script> ...                                       -> <name>.log
#   bin <name> <source.hip> [args]  hipcc -O3 the micro-benchmark if its .bin is missing, run it  -> <name>.log
#   pmc <name> <counters> <script> [args]   one rocprofv3 --pmc pass (counters comma separated, kernel trace only) -> pmc_<name>/
#   sh <name> <script.sh> [args]  bash <script.sh> args (the PMC scripts under tools/)               -> <name>.log
#   env K=V ...                  exported for the steps that follow
#   summary                      one line per bench_*.json written so far
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 9
TAG=$1; shift
[ -n "$TAG" ] || { echo "usage: gpu_call.sh <tag> <step> [args] [-- <step> [args]]..."; exit 9; }
O=gpurun_out/$TAG
mkdir -p "$O"
MAN=$O/MANIFEST.txt
echo "# call $TAG  $(date -u +%FT%TZ)  head $(cat .git_head 2>/dev/null || echo '?')" >> "$MAN"

step_tests() { timeout -k 10 1100 python -m pytest tests -m gpu -x -q "$@" > "$O/gputests.log" 2>&1; local rc=$?; tail -3 "$O/gputests.log"; [ $rc -eq 0 ] || tail -40 "$O/gputests.log"; return $rc; }
step_smoke() { timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; local rc=$?; tail -2 "$O/smoke.log"; return $rc; }
step_bench() {
  local name=$1; shift
  timeout -k 10 900 python bench.py "$@" > "$O/bench_$name.json" 2> "$O/bench_$name.err"; local rc=$?
  [ $rc -eq 0 ] || tail -25 "$O/bench_$name.err"
  grep "pass\]" "$O/bench_$name.err" || true
  return $rc
}
step_prof() {
  local name=$1; shift
  local script=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$name" -- python3 "$R/$script" "$@" > "$R/$O/${name}_profiled.log" 2>&1 ); local rc=$?
  local f; f=$(find "/tmp/prof_$name" -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$O/${name}_kernel_stats.csv" && head -8 "$O/${name}_kernel_stats.csv" | cut -c1-170
  rm -rf "/tmp/prof_$name"
  [ $rc -eq 0 ] || tail -15 "$O/${name}_profiled.log"
  return $rc
}
step_trace() {   # kernel trace kept as a CSV (gzip) + tools/trace_busy.py summary: where the GPU idles inside a frame loop
  local name=$1; shift
  local script=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "/tmp/trace_$name" -- python3 "$R/$script" "$@" > "$R/$O/${name}_traced.log" 2>&1 ); local rc=$?
  local f; f=$(find "/tmp/trace_$name" -name "*kernel_trace.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python3 "$R/tools/trace_busy.py" "$f" > "$O/${name}_busy.log" 2>&1; tail -${GPU_CALL_TAIL:-30} "$O/${name}_busy.log"
    gzip -c "$f" > "$O/${name}_kernel_trace.csv.gz"
  fi
  rm -rf "/tmp/trace_$name"
  [ $rc -eq 0 ] || tail -15 "$O/${name}_traced.log"
  return $rc
}
step_py() {
  local name=$1; shift
  timeout -k 10 900 python3 "$@" > "$O/$name.log" 2>&1; local rc=$?
  tail -${GPU_CALL_TAIL:-40} "$O/$name.log"
  return $rc
}
step_bin() {
  local name=$1; shift
  local src=$1; shift
  local bin=${src%.hip}.bin
  if [ ! -x "$bin" ] || [ "$src" -nt "$bin" ]; then
    hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -I nis-sar-amtigmti-video_amd/csrc "$src" -o "$bin" > "$O/${name}_build.log" 2>&1 || { tail -20 "$O/${name}_build.log"; return 8; }
  fi
  timeout -k 10 600 "$bin" "$@" > "$O/$name.log" 2>&1; local rc=$?
  tail -${GPU_CALL_TAIL:-60} "$O/$name.log"
  return $rc
}
step_pmc() {   # counters in their own run: kernel trace only, never with --stats / sys-trace domains
  local name=$1; shift
  local counters=$1; shift
  local script=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc ${counters//,/ } --output-format csv -d "/tmp/pmc_$name" -- python3 "$R/$script" "$@" > "$R/$O/pmc_${name}.log" 2>&1 ); local rc=$?
  mkdir -p "$O/pmc_$name"
  find "/tmp/pmc_$name" -name "*counter_collection.csv" -exec cp {} "$O/pmc_$name/" \; 2>/dev/null
  rm -rf "/tmp/pmc_$name"
  [ $rc -eq 0 ] || tail -15 "$O/pmc_${name}.log"
  return $rc
}
step_sh() {      # a script of this repo (tools/*.sh) that does its own profiling calls: bash <script> args
  local name=$1; shift
  timeout -k 10 900 bash "$@" > "$O/$name.log" 2>&1; local rc=$?
  tail -${GPU_CALL_TAIL:-20} "$O/$name.log"
  return $rc
}
step_summary() {
  python3 - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as exc:
        print(f, "unreadable:", exc); continue
    rf = d.get("roofline", {})
    print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms", round(d["value"], 1), d["unit"], "| roofline", rf.get("bound"), round(rf.get("frac", 0), 3),
          "alone", round(d.get("roofline_kernel_alone", {}).get("frac", 0), 3), "| phi2", round(d.get("roofline_rg_fft_phi2_pass", {}).get("frac", 0), 3),
          "|", {k: round(v["value"], 1) for k, v in d.items() if isinstance(v, dict) and "value" in v and k.startswith(("batch64", "config"))},
          "| parity", d.get("parity"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
}

run_step() {
  local name=$1; shift
  local t0=$SECONDS rc
  if [ "$name" = env ]; then
    for kv in "$@"; do export "$kv"; done
    echo "env $*" >> "$MAN"; return 0
  fi
  declare -F "step_$name" > /dev/null || { echo "unknown step '$name'"; return 9; }
  local before; before=$(ls -1 "$O" | sort)
  echo "== $name $*"
  "step_$name" "$@"; rc=$?
  local wrote; wrote=$(comm -13 <(echo "$before") <(ls -1 "$O" | sort) | tr '\n' ' ')
  echo "$name $* | rc $rc | $((SECONDS - t0)) s | wrote: $wrote" >> "$MAN"
  return $rc
}

args=()
for w in "$@" --; do
  if [ "$w" = "--" ]; then
    if [ ${#args[@]} -gt 0 ]; then
      run_step "${args[@]}" || { echo "STEP FAILED: ${args[*]} (no further GPU step in this call)"; exit 1; }
    fi
    args=()
  else
    args+=("$w")
  fi
done
echo "call $TAG done"; ls "$O"
