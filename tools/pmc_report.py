#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output dirs: per-kernel mean of each counter, per wave where useful."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "fill_noise" in k:
                continue
            print(d, k)
            print("   ", {c: f"{sum(v) / len(v):.4g}" for c, v in cs.items()})
    for f in glob.glob(f"{d}/*/*_kernel_trace.csv"):
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        for k, v in dur.items():
            if "fill_noise" not in k:
                print("    dur ms", k[:50], [round(x, 3) for x in v[-3:]])
