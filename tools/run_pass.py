#!/usr/bin/env python3
"""Run one CSA pass repeatedly on device-resident noise (for rocprofv3 --pmc / --kernel-trace).
    python3 tools/run_pass.py PASS_ID [size] [iters] [n_rg]     PASS_ID: 1,2,3,4,23,100,101 or 0 = whole focus
size = n_az (= n_rg unless given); 7199 13200 selects the reference's native scene with its own radar constants
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import _ffi, radar  # noqa: E402

pid = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n_rg = int(sys.argv[4]) if len(sys.argv) > 4 else n
ctx = sarx.Context(0)
plan = sarx.CsaPlan(ctx, n, n_rg, *(radar.focus_args() if n_rg == 13200 else radar.focus_args(n_rg)), flags=_ffi.FUSE_RANGE)
d_in, d_out = ctx.alloc(n * n_rg * 8), ctx.alloc(n * n_rg * 8)
ctx.fill_noise(d_in, n * n_rg, 7)
for _ in range(2):
    plan.focus_dev(d_in, d_out) if pid == 0 else plan.run_pass(pid, d_in, d_out)
ctx.sync()
ctx.record(0)
for _ in range(iters):
    if pid == 0:
        plan.focus_dev(d_in, d_out)
    else:
        plan.run_pass(pid, d_in, d_out)
ctx.record(1)
ms = ctx.elapsed_ms(0, 1) / iters
print(f"pass {pid} size {n} x {n_rg}: {ms:.3f} ms/iter, {16.0 * n * n_rg / ms / 1e6:.1f} GB/s per 16 B/sample")
