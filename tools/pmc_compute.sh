#!/bin/bash
# SQ counters of the compute-bound kernels (echo synthesis, back-projection): vector instructions per dispatch against the ISA count of
# tools/isa_slots.py, and the share of wave lifetime spent issuing them.  One --pmc pass (SQ block: 8 slots), kernel trace only.
#   bash tools/pmc_compute.sh    -> gpurun_out/pmc_compute/{videosar,echo}_counters.csv
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/pmc_compute
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_compute/videosar -- python3 $R/tools/bench_videosar.py > $R/gpurun_out/pmc_compute/videosar.log 2>&1; echo "videosar rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_compute/echo -- python3 $R/tools/bench_echo.py 5000 1200 > $R/gpurun_out/pmc_compute/echo.log 2>&1; echo "echo rc $?"
cd $R
for t in videosar echo; do f=$(find gpurun_out/pmc_compute/$t -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/pmc_compute/${t}_counters.csv; done
python3 - <<'PY'
import csv, collections, json
for t, kern in (("videosar", ("tdbp_tile_kernel", "tdbp_kernel", "echo_synth_kernel")), ("echo", ("echo_synth_kernel",))):
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/pmc_compute/{t}_counters.csv")))
    except OSError:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        for k in kern:
            if ("::" + k + "(") in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
    for k, d in out.items():
        if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d:
            d["valu_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        if "SQ_ACTIVE_INST_VALU" in d and "SQ_WAVE_CYCLES" in d:
            d["share_of_wave_lifetime_issuing_valu"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
    json.dump(out, open(f"gpurun_out/pmc_compute/{t}_summary.json", "w"), indent=1)
    print(t, json.dumps(out))
PY
