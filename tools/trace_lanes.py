#!/usr/bin/env python3
"""Timeline of two frames in flight from a rocprofv3 --kernel-trace CSV: which launches overlap which, and for how long.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/bench_two_streams.py 16384 24 lanes
    python3 tools/trace_lanes.py DIR/*/*_kernel_trace.csv"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "fill_noise" in n:
        continue
    kind = "R" if "range" in n else "a"
    if kind == "a":
        i0, i1 = n.find("<"), n.find(">")
        args = n[i0 + 1:i1].replace(" ", "").split(",")
        kind = {("false", "1"): "A", ("false", "2"): "B", ("true", "1"): "C", ("true", "3"): "D"}.get((args[2], args[3]), "a")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
t0 = rows[0][0]
qs = sorted({q for *_, q in rows})
print("queues:", qs, " launches:", len(rows))
# steady state: the last 40 launches
tail = rows[-45:-5]
for s, e, k, q in tail:
    lane = qs.index(q)
    others = [(k2, max(0, min(e, e2) - max(s, s2))) for s2, e2, k2, q2 in rows if q2 != q and s2 < e and e2 > s]
    ov = " ".join(f"{k2}:{o / 1e3:.0f}us" for k2, o in others)
    print(f"{(s - t0) / 1e6:9.3f} ms  lane {lane}  {k}  {(e - s) / 1e3:7.1f} us   overlaps {ov}")
# per-kind mean duration and mean overlap with each other kind
import collections
dur, ovl = collections.defaultdict(list), collections.defaultdict(float)
span = tail[-1][1] - tail[0][0]
for s, e, k, q in tail:
    dur[k].append(e - s)
    for s2, e2, k2, q2 in rows:
        if q2 != q and s2 < e and e2 > s:
            ovl[(k, k2)] += max(0, min(e, e2) - max(s, s2))
print("mean durations (us):", {k: round(sum(v) / len(v) / 1e3, 1) for k, v in sorted(dur.items())})
print("share of the window in which X (one lane) runs beside Y (other lane):", {f"{a}|{b}": round(v / span, 3) for (a, b), v in sorted(ovl.items())})
print(f"window {span / 1e6:.3f} ms for {len(tail)} launches = {span / 1e6 / (len(tail) / 5):.3f} ms per frame")
