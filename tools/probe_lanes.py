#!/usr/bin/env python3
"""Do two lanes of a context run side by side or take turns (their HIP streams sharing one hardware pipe)?  Probe every lane pair,
then time two frames in flight on each pair: does the probe predict the frames-in-flight gain?
    python3 tools/probe_lanes.py [size=16384] [frames=40]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import _ffi, radar  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = sarx.Context(0)
lib = ctx.lib
lib.sarx_probe_lanes.restype = C.c_int
lib.sarx_probe_lanes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
plans = [sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE) for _ in range(2)]
bufs = [(ctx.alloc(n * n * 8), ctx.alloc(n * n * 8)) for _ in range(2)]
for i in range(2):
    ctx.fill_noise(bufs[i][0], n * n, 1000 + i)


def in_flight(a, b):
    lanes = (a, b)
    ctx.set_range_cus(192)
    for f in range(2):
        ctx.select_lane(lanes[f])
        plans[f].focus_dev(*bufs[f])
    ctx.sync()
    t0 = time.perf_counter()
    for f in range(frames):
        ctx.select_lane(lanes[f & 1])
        plans[f & 1].focus_dev(*bufs[f & 1])
    ctx.sync()
    ctx.select_lane(0)
    ctx.set_range_cus(0)
    return (time.perf_counter() - t0) / frames * 1e3


for a in range(4):
    for b in range(a + 1, 4):
        r = C.c_double()
        _ffi.check(lib.sarx_probe_lanes(ctx.h, a, b, 300, C.byref(r)), ctx.h)
        ms = in_flight(a, b)
        print(f"lanes {a},{b}: probe ratio {r.value:.2f} (1 = side by side, 2 = taking turns)   two frames in flight {ms:.3f} ms per frame", flush=True)
