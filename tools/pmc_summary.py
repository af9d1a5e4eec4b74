#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output directories into one JSON: per kernel the mean of every counter over its dispatches,
the dispatch durations of the same runs, and a few ratios (per wave, per SIMD).
    python3 tools/pmc_summary.py OUT.json DIR [DIR ...]"""
import collections
import csv
import glob
import json
import sys

out, dirs = sys.argv[1], sys.argv[2:]
kern = collections.defaultdict(lambda: {"counters": {}, "dur_ms": []})
for d in dirs:
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "fill_noise" in k:
                continue
            for c, v in cs.items():
                kern[k]["counters"][c] = sum(v) / len(v)
    for f in glob.glob(f"{d}/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "fill_noise" not in r["Kernel_Name"]:
                kern[r["Kernel_Name"]]["dur_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
res = {"units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles summed over all waves (MI355X_MICROARCH.md); "
                "FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE reports half of a wide coalesced read on gfx950", "kernels": {}}
for k, v in kern.items():
    c = v["counters"]
    d = {"counters": c, "dispatch_ms_under_profiler": {"n": len(v["dur_ms"]), "mean": sum(v["dur_ms"]) / max(len(v["dur_ms"]), 1)}}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        share = lambda name: round(c[name] / wc, 4) if name in c else None
        d["share_of_wave_lifetime"] = {n: share(n) for n in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                                                             "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY")}
        if "SQ_WAVES" in c:
            d["per_wave"] = {"quad_cycles": wc / c["SQ_WAVES"], "valu_insts": c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"],
                             "lds_insts": c.get("SQ_INSTS_LDS", 0) / c["SQ_WAVES"]}
        if "SQ_INSTS_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
            d["quad_cycles_per_valu_inst"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_share_of_lds_cycles"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    res["kernels"][k] = d
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
