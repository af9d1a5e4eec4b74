#!/usr/bin/env python3
"""Wall time of the drop-in host-array call (sar_focus_csa on NumPy in, NumPy out): what a maintainer of the reference sees
at sar_ati_dcpa_sim_csa.py:410-411, PCIe included.
    python3 tools/bench_hostpath.py [size=8192] [--json FILE]
Reports the first calls (plan creation, twiddles; the first three results of a size are ordinary pageable arrays, the fourth and fifth pay
hipHostMalloc for the two page-locked blocks the loop then alternates between), the steady state with the result
allocated per call from the page-locked pool (results alternate between two blocks, as `img = f(raw)` in a loop does), the
`out=` form, and the complex128 input of the reference's own arrays.  The floor is 2 x bytes / PCIe rate + the focus itself."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import radar  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 8192
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
rng = np.random.default_rng(0)
raw = np.empty((n, n), np.complex64)
for i0 in range(0, n, 1024):
    blk = rng.standard_normal((min(1024, n - i0), n, 2), dtype=np.float32)
    raw[i0:i0 + blk.shape[0]] = blk[..., 0] + 1j * blk[..., 1]
args = radar.focus_args(n)
res = {"size": n, "bytes_each_way": raw.nbytes, "calls_ms": [], "out_ms": [], "c128_ms": []}


def timed(f):
    t0 = time.perf_counter()
    r = f()
    return r, (time.perf_counter() - t0) * 1e3


img = None
for rep in range(10):
    (img, rax, cax), ms = timed(lambda: sarx.sar_focus_csa(raw, *args))
    res["calls_ms"].append(round(ms, 2))
    print(f"sar_focus_csa {n}x{n} host in / host out, call {rep}: {ms:.1f} ms wall ({2 * raw.nbytes / ms / 1e6:.1f} GB/s over both directions)", flush=True)
ref = img.copy() if n <= 8192 else None
for rep in range(3):
    (img2, _, _), ms = timed(lambda: sarx.sar_focus_csa(raw, *args, out=img))
    res["out_ms"].append(round(ms, 2))
    print(f"   ... out=<previous result>: {ms:.1f} ms", flush=True)
if ref is not None:
    assert np.array_equal(img2, ref)
# device-resident focus alone, for the floor
ctx = sarx.default_context()
d_in, d_out = ctx.to_device(raw), ctx.alloc(raw.nbytes)
plan = sarx.CsaPlan(ctx, n, n, *args, flags=sarx._ffi.FUSE_RANGE)
plan.focus_dev(d_in, d_out)
ctx.sync()
ctx.record(0)
for _ in range(5):
    plan.focus_dev(d_in, d_out)
ctx.record(1)
res["focus_dev_ms"] = round(ctx.elapsed_ms(0, 1) / 5, 3)
_, up = timed(lambda: d_in.upload(raw))
pinned = ctx.pinned_empty(raw.shape, np.complex64)
_, down = timed(lambda: sarx._ffi.check(ctx.lib.sarx_memcpy_d2h(ctx.h, pinned.ctypes.data, d_out.ptr, raw.nbytes), ctx.h))
res["upload_pageable_ms"], res["download_pinned_ms"] = round(up, 2), round(down, 2)
print(f"   parts: staged upload of the pageable input {up:.1f} ms ({raw.nbytes / up / 1e6:.1f} GB/s), focus {res['focus_dev_ms']:.2f} ms, "
      f"download into a pooled page-locked result {down:.1f} ms ({raw.nbytes / down / 1e6:.1f} GB/s)", flush=True)
plan.close(); d_in.release(); d_out.release()
del pinned
if n <= 8192:
    raw128 = raw.astype(np.complex128)
    for rep in range(2):
        (img3, _, _), ms = timed(lambda: sarx.sar_focus_csa(raw128, *args))
        res["c128_ms"].append(round(ms, 2))
        print(f"sar_focus_csa {n}x{n} complex128 in (the reference's dtype) / complex64 out: {ms:.1f} ms wall", flush=True)
    assert np.array_equal(img3, ref)
# ---- the same frames as a pipeline (round 5): frame i+1 uploads while frame i focuses and downloads -------------------------------
frames = [raw, raw[::-1].copy()] if n <= 8192 else [raw, raw]          # two different inputs where memory allows
sync_ref = [sarx.sar_focus_csa(f, *args)[0].copy() for f in frames] if n <= 8192 else None
t_first = time.perf_counter()
got, stamps = [], []
t_prev = time.perf_counter()
for k, (img_k, _, _) in enumerate(sarx.focus_stream((frames[i % 2] for i in range(12)), *args)):
    stamps.append((time.perf_counter() - t_prev) * 1e3)
    if sync_ref is not None:                    # the comparison is the consumer's work, not the pipeline's: outside the interval
        assert np.array_equal(img_k, sync_ref[k % 2]), f"pipelined frame {k} differs from the synchronous call"
    del img_k                                   # the consumer drops the result: its block goes back to the pool
    t_prev = time.perf_counter()
res["stream_ms"] = [round(x, 2) for x in stamps]
steady = sorted(stamps[3:])
res["stream_steady_ms"] = round(steady[len(steady) // 2], 2)
res["stream_worst_after_first_ms"] = round(max(stamps[1:]), 2)
print(f"focus_stream, 12 frames: per-frame intervals {res['stream_ms']} ms; median of frames 3.. {res['stream_steady_ms']} ms, worst after the "
      f"first {res['stream_worst_after_first_ms']} ms ({res['stream_worst_after_first_ms'] / res['stream_steady_ms']:.2f} x the median)" +
      ("; every frame bit-identical to the synchronous call" if sync_ref is not None else ""), flush=True)
# the same pipeline on PAGE-LOCKED frames (a caller that reads its frames into Context.pinned_empty arrays): both directions are plain
# DMAs, no staging memcpy on the host
pctx = sarx.default_context()
pctx.reserve_pinned(raw.shape, np.complex64, 5)
pframes = [pctx.pinned_empty(raw.shape, np.complex64, force=True) for _ in range(2)]
for pf_, f_ in zip(pframes, frames):
    pf_[...] = f_
stamps_p = []
t_prev = time.perf_counter()
for k, (img_k, _, _) in enumerate(sarx.focus_stream((pframes[i % 2] for i in range(12)), *args)):
    stamps_p.append((time.perf_counter() - t_prev) * 1e3)
    if sync_ref is not None:
        assert np.array_equal(img_k, sync_ref[k % 2])
    del img_k
    t_prev = time.perf_counter()
res["stream_pinned_ms"] = [round(x, 2) for x in stamps_p]
steady_p = sorted(stamps_p[3:])
res["stream_pinned_steady_ms"] = round(steady_p[len(steady_p) // 2], 2)
print(f"focus_stream on page-locked frames: {res['stream_pinned_ms']} ms; median of frames 3.. {res['stream_pinned_steady_ms']} ms", flush=True)
del pframes
# the future form, driven by hand: begin(i+1) before result(i)
fut, ivals = None, []
t_prev = time.perf_counter()
for i in range(8):
    nxt = sarx.sar_focus_csa_async(frames[i % 2], *args)
    if fut is not None:
        fut.result()
        now = time.perf_counter(); ivals.append((now - t_prev) * 1e3); t_prev = now
    fut = nxt
fut.result()
res["async_ms"] = [round(x, 2) for x in ivals]
print(f"sar_focus_csa_async, begin(i+1) before result(i): {res['async_ms']} ms per frame", flush=True)
# the whole two-channel processing section on host arrays (sar_ati_dcpa_sim_csa.py:402-419,447-449): two uploads, two focuses, five planes
# back - channel 2 uploads while channel 1 focuses, slc1 downloads meanwhile, every plane's download in flight at once
if n <= 8192:
    raw_b = frames[1]
    tc = []
    for rep in range(7):
        _, ms = timed(lambda: sarx.focus_ati_dpca(raw, raw_b, *args, pulse_shift=False))
        tc.append(round(ms, 2))
    res["two_channel_host_ms"] = tc
    moved = 2 * raw.nbytes + 2 * raw.nbytes + 3 * raw.nbytes // 2
    print(f"focus_ati_dpca {n}x{n} on host arrays (2 echoes up, slc1 + slc2 + 3 planes down = {moved / 2**30:.2f} GiB): {tc} ms per call "
          f"(one direction after the other at 55 GB/s: {moved / 55e6:.0f} ms)", flush=True)
res["steady_ms"] = min(res["calls_ms"][6:])
res["floor_ms"] = round(res["upload_pageable_ms"] + res["focus_dev_ms"] + res["download_pinned_ms"], 2)
print(json.dumps(res))
if out_json:
    with open(out_json, "w") as fh:
        json.dump(res, fh, indent=1)
