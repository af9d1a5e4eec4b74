#!/usr/bin/env python3
"""How far do an upload and a download overlap on this box?  (The floor of the host-frame pipeline, sarx.focus_stream.)
    python3 tools/bench_duplex.py [MiB=2048]
Times, for one buffer size: the staged upload of a pageable array alone, the asynchronous download into page-locked memory alone, both
together (download in flight on the download stream while the copy threads upload), and the same with a page-locked SOURCE (both
directions plain DMA, no host memcpy)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nbytes = mib << 20
ctx = sarx.Context(0)
src = np.ones(nbytes // 8, dtype=np.complex64)                       # pageable, touched
d_up, d_dn = ctx.alloc(nbytes), ctx.alloc(nbytes)
ctx.fill_noise(d_dn, nbytes // 8, 1)
ctx.reserve_pinned((nbytes // 8,), np.complex64, 2)
pin_src = ctx.pinned_empty((nbytes // 8,), np.complex64, force=True)
pin_src[:] = 1
ctx.sync()


def t(f, reps=4):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def both(source):
    p = d_dn.download_begin(np.complex64, (nbytes // 8,))
    d_up.upload_unordered(source)
    p.result()


up = t(lambda: d_up.upload_unordered(src))
dn = t(lambda: d_dn.download_begin(np.complex64, (nbytes // 8,)).result())
bo = t(lambda: both(src))
up_p = t(lambda: d_up.upload_unordered(pin_src))
bo_p = t(lambda: both(pin_src))
g = lambda ms: nbytes / ms / 1e6
print(f"{mib} MiB each way, env HSA_ENABLE_SDMA={os.environ.get('HSA_ENABLE_SDMA')} SARX_COPY_THREADS={os.environ.get('SARX_COPY_THREADS')}")
print(f"  staged upload of a pageable array alone      {up:7.1f} ms  {g(up):5.1f} GB/s")
print(f"  download into page-locked memory alone       {dn:7.1f} ms  {g(dn):5.1f} GB/s")
print(f"  both together (pageable source)              {bo:7.1f} ms  = {bo / max(up, dn):.2f} x the slower one alone, {2 * nbytes / bo / 1e6:5.1f} GB/s over both directions")
print(f"  upload of a page-locked array alone          {up_p:7.1f} ms  {g(up_p):5.1f} GB/s")
print(f"  both together (page-locked source: two DMAs) {bo_p:7.1f} ms  = {bo_p / max(up_p, dn):.2f} x the slower one alone")
