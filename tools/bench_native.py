#!/usr/bin/env python3
"""Time the reference's native scene size (7199 x 13200, sar_ati_dcpa_sim_csa.py:47,111,402): direct mixed-radix range lines and
prime-factor azimuth transforms, one frame at a time and with two frames in flight on two lanes of the context.
    python3 tools/bench_native.py [n_az=7199] [n_rg=13200] [frames=20]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import radar  # noqa: E402

n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 7199
n_rg = int(sys.argv[2]) if len(sys.argv) > 2 else 13200
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = sarx.Context(0)
px = n_az * n_rg
plans = [sarx.CsaPlan(ctx, n_az, n_rg, *radar.focus_args()) for _ in range(2)]
bufs = [(ctx.alloc(px * 8), ctx.alloc(px * 8)) for _ in range(2)]
for i, (d_in, _) in enumerate(bufs):
    ctx.fill_noise(d_in, px, 3 + i)
plans[0].focus_dev(*bufs[0])
ctx.sync()
ctx.record(0)
for _ in range(5):
    plans[0].focus_dev(*bufs[0])
ctx.record(1)
ms = ctx.elapsed_ms(0, 1) / 5
print(f"CSA focus {n_az} x {n_rg}: {ms:.2f} ms/frame = {1e3 / ms:.1f} frames/s; plan scratch {plans[0].scratch_bytes() / 2**30:.2f} GiB")
for lanes in (1, 2, 1, 2):
    for f in range(lanes):
        ctx.select_lane(f)
        plans[f].focus_dev(*bufs[f])
    ctx.sync()
    t0 = time.perf_counter()
    for f in range(frames):
        ctx.select_lane(f % lanes)
        plans[f % lanes].focus_dev(*bufs[f % lanes])
    ctx.sync()
    ms = (time.perf_counter() - t0) / frames * 1e3
    ctx.select_lane(0)
    print(f"   {lanes} frame(s) in flight: {ms:.3f} ms/frame = {1e3 / ms:.1f} frames/s")
