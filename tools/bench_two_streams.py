#!/usr/bin/env python3
"""Do two frames in flight finish a stream of independent frames sooner than one frame at a time?  The range launch is
issue-bound and leaves half the HBM bandwidth unused, the azimuth launches are bandwidth-bound and leave the vector pipes idle,
so kernels of neighbouring frames can fill each other's gaps when the hardware scheduler mixes them.
    python3 tools/bench_two_streams.py [size=16384] [frames=40] [mode ...]
modes: ctx    one sarx.Context (= one compute stream) per frame in flight
       lanes  one context, sarx_select_lane per frame
       mark   lanes + HIP events recorded around every range launch (what bench.py's timed region does)
(round 4 also measured an explicit stagger - frame f+1's first launch waiting for the event before frame f's range launch - with a
 temporary sarx_event_wait: 4.24 ms against 4.09 ms free-running on the same box, profiles/r04_i_stagger.log; removed)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import _ffi, radar  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
modes = sys.argv[3:] or ["ctx", "lanes", "mark"]
KMAX = 3


def setup(mode):
    ctxs = [sarx.Context(0) for _ in range(KMAX)] if mode == "ctx" else [sarx.Context(0)] * KMAX
    plans, bufs = [], []
    for i, c in enumerate(ctxs):
        plans.append(sarx.CsaPlan(c, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE))
        d_in, d_img = c.alloc(n * n * 8), c.alloc(n * n * 8)
        c.fill_noise(d_in, n * n, 1000 + i)
        bufs.append((d_in, d_img))
    return ctxs, plans, bufs


def run(mode, ctxs, plans, bufs, k):
    def frame(f):
        i = f % k
        if mode != "ctx":
            ctxs[0].select_lane(i)
        if mode == "mark":
            plans[i].mark_range(2 * (f % 100), 2 * (f % 100) + 1)
        plans[i].focus_dev(*bufs[i])
    for c in ctxs[:k]:
        c.sync()
    for f in range(k):
        frame(f)
    for c in ctxs[:k]:
        c.sync()
    t0 = time.perf_counter()
    nfill = int(os.environ.get("TWO_STREAMS_DELAY_FILLS", "0"))      # one-time head start for lane 0: lane 1 first fills a scratch image n times
    if nfill and k > 1 and mode != "ctx":
        ctxs[0].select_lane(1)
        for _ in range(nfill):
            ctxs[0].fill_noise(bufs[2][1], n * n, 5)
    for f in range(frames):
        frame(f)
    for c in ctxs[:k]:
        c.sync()
    if mode != "ctx":
        ctxs[0].select_lane(0)
    return (time.perf_counter() - t0) / frames * 1e3


for mode in modes:
    ctxs, plans, bufs = setup(mode)
    for rep in range(2):
        for k in [int(x) for x in os.environ.get("TWO_STREAMS_K", "1,2,3,1").split(",")]:
            ms = run(mode, ctxs, plans, bufs, k)
            print(f"{n}x{n} [{mode:5s}]: {k} frame(s) in flight  {ms:.3f} ms per frame  {1e3 / ms:.1f} frames/s", flush=True)
    for p in plans:
        p.close()
    for a, b in bufs:
        a.release(); b.release()
    for c in set(ctxs):
        c.close()
