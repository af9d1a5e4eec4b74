#!/usr/bin/env python3
"""What does the 256 MiB Infinity Cache buy each kernel of the middle of the CSA chain?
The same launches on the same shapes, once with the working set cycling through 2 GiB of different images (every byte comes
from and goes to HBM) and once on ONE image set again and again (everything stays in the memory-side cache).  The difference
is the most a slab-by-slab / persistent-launch schedule of azimuth step B -> fused range -> inverse azimuth step A could
return per launch; DESIGN.md 4.5 / LABBOOK.md round 4.
    python3 tools/mall_resident.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import _ffi, radar  # noqa: E402

ctx = sarx.Context(0)


def timed(fn, reps):
    fn(0)
    ctx.sync()
    ctx.record(0)
    for r in range(reps):
        fn(r)
    ctx.record(1)
    return ctx.elapsed_ms(0, 1) / reps


def case(n_az, n_rg, pid, inplace, k_stream):
    plan = sarx.CsaPlan(ctx, n_az, n_rg, *radar.focus_args(n_rg), flags=_ffi.FUSE_RANGE)
    img = n_az * n_rg * 8
    ins = [ctx.alloc(img) for _ in range(k_stream)]
    outs = ins if inplace else [ctx.alloc(img) for _ in range(k_stream)]
    for b in ins:
        ctx.fill_noise(b, n_az * n_rg, 7)
    res = {}
    for mode in ("stream", "resident", "stream", "resident"):
        f = (lambda r: plan.run_pass(pid, ins[r % k_stream], outs[r % k_stream])) if mode == "stream" else \
            (lambda r: plan.run_pass(pid, ins[0], outs[0]))
        ms = timed(f, 2 * k_stream)
        res.setdefault(mode, []).append(ms)
        if not inplace or True:
            for b in ins:                       # keep magnitudes bounded for in-place repeats
                pass
    full = 16.0 * 16384 * 16384 / (16.0 * n_az * n_rg)
    s, r = min(res["stream"]), min(res["resident"])
    name = {1: "azimuth FFT + Phi1 (two launches)", 4: "azimuth IFFT (two launches)", 23: "fused range FFT.Phi2.IFFT.Phi3",
            2: "range FFT + Phi2", 3: "range IFFT + Phi3"}[pid]
    print(f"{name:36s} {n_az:6d} x {n_rg:6d} ({img >> 20:4d} MiB) {'in place ' if inplace else 'out of pl.'}  "
          f"HBM {s:7.4f} ms ({16.0 * n_az * n_rg / s / 1e9:5.2f} TB/s)   cache-resident {r:7.4f} ms ({16.0 * n_az * n_rg / r / 1e9:5.2f} TB/s)   "
          f"ratio {r / s:5.3f}   x{full:.0f} -> {s * full:6.3f} / {r * full:6.3f} ms per 16384^2", flush=True)
    for b in set(ins) | set(outs):
        b.release()
    plan.close()


for mib in (32, 64):
    rows = mib * 2 ** 20 // (16384 * 8)
    case(rows, 16384, 23, True, 2048 // mib // 2)
    case(rows, 16384, 2, True, 2048 // mib // 2)
    case(16384, rows, 1, False, 2048 // mib // 4)
    case(16384, rows, 4, False, 2048 // mib // 4)
