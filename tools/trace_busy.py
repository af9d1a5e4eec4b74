#!/usr/bin/env python3
"""How busy is the GPU inside a frame loop?  From a rocprofv3 --kernel-trace CSV: the union of all kernel intervals against the
window they span (steady state: the middle 60 % of the launches; gaps longer than 2 ms separate phases of the script - file output
between two loops - and are reported but not counted as idle time of a loop), the largest idle gaps and what ran on either side of
them, and per-kernel time inside the window.
    python3 tools/trace_busy.py DIR/*/*_kernel_trace.csv"""
import collections
import csv
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
n = len(rows)
mid = rows[n // 5: n - n // 5] if n > 50 else rows
t0, t1 = mid[0][0], max(e for _, e, _ in mid)
busy, cur_s, cur_e, gaps = 0, mid[0][0], mid[0][1], []
last_name = mid[0][2]
for s, e, k in mid[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, k))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = k
busy += cur_e - cur_s
PHASE = 2_000_000
between = sum(g for g, _, _ in gaps if g > PHASE)
gaps = [g for g in gaps if g[0] <= PHASE]
span = t1 - t0 - between
print(f"{len(mid)} launches in {span / 1e6:.3f} ms of loops (+ {between / 1e6:.1f} ms between phases): GPU busy {busy / span:.3f} of the loops, "
      f"idle {(span - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
per = collections.defaultdict(lambda: [0, 0])
for s, e, k in mid:
    per[k][0] += e - s
    per[k][1] += 1
print("kernel time inside the window (sum of durations; overlapping launches count twice):")
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:10]:
    print(f"  {t / 1e6:8.3f} ms  {c:5d} x {t / c / 1e3:8.1f} us  {k}")
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    agg[(a, b)][0] += g
    agg[(a, b)][1] += 1
print("idle time by (kernel before -> kernel after):")
for (a, b), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
    print(f"  {t / 1e6:8.3f} ms in {c:4d} gaps (mean {t / c / 1e3:7.1f} us)  {a}  ->  {b}")
