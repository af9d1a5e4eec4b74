#!/usr/bin/env python3
"""Time the reference's native scene size (7199 x 13200, sar_ati_dcpa_sim_csa.py:47,111,402) on the chirp-z path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import radar  # noqa: E402

n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 7199
n_rg = int(sys.argv[2]) if len(sys.argv) > 2 else 13200
ctx = sarx.Context(0)
plan = sarx.CsaPlan(ctx, n_az, n_rg, *radar.focus_args())
px = n_az * n_rg
d_in, d_out = ctx.alloc(px * 8), ctx.alloc(px * 8)
ctx.fill_noise(d_in, px, 3)
plan.focus_dev(d_in, d_out)
ctx.sync()
ctx.record(0)
for _ in range(5):
    plan.focus_dev(d_in, d_out)
ctx.record(1)
ms = ctx.elapsed_ms(0, 1) / 5
print(f"CSA focus {n_az} x {n_rg}: {ms:.2f} ms/frame = {1e3 / ms:.1f} frames/s; plan scratch {plan.scratch_bytes() / 2**30:.2f} GiB")
