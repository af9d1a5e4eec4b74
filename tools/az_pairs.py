#!/usr/bin/env python3
"""Why do two azimuth launches of different frames, side by side, reach 6.0 TB/s where one alone stays at 5.4?  The forward azimuth
pass (two launches) on two lanes at once, with the two lanes' data (A) entirely distinct, (B) sharing the input image, (C) sharing
input, scratch and output (a race on the values, same addresses: timing only), against one lane alone.
    python3 tools/az_pairs.py [size=16384] [reps=12]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import _ffi, radar  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = sarx.Context(0)
lanes = ctx.concurrent_lanes(2)
plans = [sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE) for _ in range(2)]
x = [ctx.alloc(n * n * 8) for _ in range(2)]
y = [ctx.alloc(n * n * 8) for _ in range(2)]
for i in range(2):
    ctx.fill_noise(x[i], n * n, 10 + i)


def run(jobs, pid):
    """jobs: list of (lane index, plan, in, out); every job repeated reps times, interleaved"""
    for lane, p, a, b in jobs:
        ctx.select_lane(lanes[lane])
        p.run_pass(pid, a, b)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        for lane, p, a, b in jobs:
            ctx.select_lane(lanes[lane])
            p.run_pass(pid, a, b)
    ctx.sync()
    ctx.select_lane(0)
    return (time.perf_counter() - t0) / reps / len(jobs) * 1e3


for pid, name in ((_ffi.PASS_AZ_FFT_PHI1, "azimuth FFT + Phi1"), (_ffi.PASS_AZ_IFFT, "azimuth IFFT")):
    for rep in range(2):
        one = run([(0, plans[0], x[0], y[0])], pid)
        a = run([(0, plans[0], x[0], y[0]), (1, plans[1], x[1], y[1])], pid)
        b = run([(0, plans[0], x[0], y[0]), (1, plans[1], x[0], y[1])], pid)
        c = run([(0, plans[0], x[0], y[0]), (1, plans[0], x[0], y[0])], pid)
        print(f"{name} (two launches) per pass: alone {one:.3f} ms | two lanes, distinct data {a:.3f} | shared input {b:.3f} | "
              f"shared input, scratch and output {c:.3f}", flush=True)
