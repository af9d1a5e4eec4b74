#!/usr/bin/env python3
"""BASELINE config 3 with the two channels of ONE scene in flight together: channel 1's focus on lane 0, channel 2's first three
passes on lane 1 at the same time; channel 2's last azimuth pass - the one that reads slc1 and emits the ATI/DPCA products -
after the lanes have joined.  Against the plain fused chain (one channel after the other) of tools/bench_twochannel.py.
    python3 tools/bench_twochannel_lanes.py [size=8192] [frames=20]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import _ffi, radar  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = sarx.Context(0)
px = n * n
plans = [sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE) for _ in range(2)]
raw1, raw2, s1, s2, tmp = (ctx.alloc(px * 8) for _ in range(5))
planes = [[ctx.alloc(px * 4) for _ in range(3)] for _ in range(2)]
d_max = ctx.alloc(_ffi.MAX_SLOT_BYTES)
ctx.fill_noise(raw1, px, 1)
ctx.fill_noise(raw2, px, 2)


def serial(out):
    p = plans[0]
    p.set_max_slot(d_max)
    p.focus_dev(raw1, s1)
    p.set_max_slot(None)
    p.set_ati(s1, d_max, 0.05, 0.0, *out)
    p.focus_dev(raw2, s2)
    p.set_ati(None)


def lanes(out):
    p0, p1 = plans
    ctx.select_lane(0)
    p0.set_max_slot(d_max)
    p0.focus_dev(raw1, s1)
    p0.set_max_slot(None)
    ctx.select_lane(1)
    p1.run_pass(_ffi.PASS_AZ_FFT_PHI1, raw2, tmp)
    p1.run_pass(_ffi.PASS_RG_FUSED_23, tmp, tmp)
    ctx.select_lane(0)
    ctx.lanes_join()                                   # lane 0 now also waits for channel 2's range pass; lane 1 for channel 1's image
    p1.set_ati(s1, d_max, 0.05, 0.0, *out)
    p1.run_pass(_ffi.PASS_AZ_IFFT, tmp, s2)
    p1.set_ati(None)


for fn, out in ((serial, planes[0]), (lanes, planes[1])):
    fn(out)
ctx.sync()
same = all(np.array_equal(a.download(np.float32, (4, n)), b.download(np.float32, (4, n))) for a, b in zip(*planes))
for rep in range(2):
    for name, fn, out in (("one channel after the other", serial, planes[0]), ("two channels in flight  ", lanes, planes[1])):
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            fn(out)
        ctx.sync()
        ms = (time.perf_counter() - t0) / frames * 1e3
        print(f"two-channel {n}x{n}, {name}: {ms:.3f} ms per scene = {1e3 / ms:.1f} scenes/s   planes bit-identical: {same}", flush=True)
