#!/usr/bin/env python3
"""The LDS corner-turn kernel (BASELINE north_star: "a corner-turn kernel stages tiles through LDS for the range<->azimuth
transpose"; sarx_corner_turn_dev, SARX_OUT_RG_MAJOR) alone: HIP-event time and GB/s at the reference's and BASELINE's sizes.
Algorithmic bytes: 16 B per sample (one complex64 read + one write).
    python3 tools/bench_corner_turn.py [iters=20]
(profiles/r05_bf_corner_turn.log was taken with a build that still held the kernel of rounds 1-4, SARX_CT_V0=1, and a
nontemporal form, SARX_CT_NT=1; the library keeps neither.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = sarx.default_context()
tag = "plain accesses, loads issued together"
for rows, cols in ((4096, 4096), (8192, 8192), (7199, 13200), (32768, 2048), (16384, 16384)):
    n = rows * cols
    d_in, d_out = ctx.alloc(n * 8), ctx.alloc(n * 8)
    ctx.fill_noise(d_in, n, 11)
    for _ in range(3):
        ctx.corner_turn(d_in, d_out, rows, cols)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        ctx.corner_turn(d_in, d_out, rows, cols)
    ctx.sync()
    ms = (time.perf_counter() - t0) / iters * 1e3
    # spot check: a few rows of the result against the source
    src = d_in.download(np.complex64, (rows, cols))[:3, :5]
    dst = d_out.download(np.complex64, (cols, rows))[:5, :3]
    assert np.array_equal(src.T, dst)
    print(f"corner turn {rows} x {cols} [{tag}]: {ms:.3f} ms = {16.0 * n / ms / 1e9:.2f} TB/s = {16.0 * n / ms / 1e9 / 8.0:.3f} of 8 TB/s", flush=True)
    d_in.release(); d_out.release()
