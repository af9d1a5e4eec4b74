#!/usr/bin/env python3
"""BASELINE config 3: two-channel n x n scene, CSA focus of both channels + ATI/DPCA on one MI355X.
    python3 tools/bench_twochannel.py [size=8192 | native] [frames=10] [fused | facade | masked | two-pass] [--json FILE]
Echoes are device-resident noise; prints ms per frame, frames/s, and the ATI/DPCA kernel's GB/s
against its 28 B/pixel algorithmic traffic.  --json writes one JSON object with the numbers.  (The CPU baseline
of this configuration is `bench.py --config3`: only bench.py's cpu_baseline leg runs the oracle.)"""
import json
import os
import sys
import time

json_path = None
if "--json" in sys.argv:
    i = sys.argv.index("--json")
    json_path = sys.argv[i + 1]
    del sys.argv[i:i + 2]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import _ffi, radar  # noqa: E402

native = len(sys.argv) > 1 and sys.argv[1] == "native"            # the reference's own scene: 7199 pulses x 13200 samples per channel
n = 8192 if native or len(sys.argv) < 2 else int(sys.argv[1])
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = sarx.Context(0)
if native:
    n_az, n_rg = 7199, 13200
    plan = sarx.CsaPlan(ctx, n_az, n_rg, *radar.focus_args())
else:
    n_az = n_rg = n
    plan = sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE)
px = n_az * n_rg
raw1, raw2, s1, s2 = (ctx.alloc(px * 8) for _ in range(4))
outs = {k: ctx.alloc(px * 4) for k in ("ati_phase", "slc1_mag", "dpca_mag")}
masked = ctx.alloc(px * 4)
ctx.fill_noise(raw1, px, 1)
ctx.fill_noise(raw2, px, 2)


mode = sys.argv[3] if len(sys.argv) > 3 else "fused"             # fused: products out of channel 2's last azimuth launch; facade: the same through sarx.focus_ati_dpca;
two_pass = mode == "two-pass"                                    # masked: mask inside the ATI launch; two-pass: mask as a further launch
d_max = ctx.alloc(_ffi.MAX_SLOT_BYTES)


facade_ws = None
if mode == "facade":             # the public function a reference maintainer calls (sar_ati_dcpa_sim_csa.py:402-419,447-449), device-resident
    from sarx.engine import DeviceArray
    facade_ws = sarx.two_channel_workspace(ctx, n_az, n_rg)
    dev1, dev2 = DeviceArray(raw1, (n_az, n_rg), owner=False), DeviceArray(raw2, (n_az, n_rg), owner=False)
    fargs = radar.focus_args() if native else radar.focus_args(n)


def frame():
    if mode == "facade":
        ctx.record(10)
        sarx.focus_ati_dpca(dev1, dev2, *fargs, ctx=ctx, pulse_shift=False, return_slc2=False, device_output=True,
                            workspace=facade_ws, fetch_stats=False)
        ctx.record(11)
        return
    if mode == "fused":
        plan.set_max_slot(d_max)
        plan.focus_dev(raw1, s1)
        plan.set_max_slot(None)
        ctx.record(10)
        plan.set_ati(s1, d_max, 0.05, 0.0, masked, outs["slc1_mag"], outs["dpca_mag"])
        plan.focus_dev(raw2, s2)                                 # s2 is scratch: slc2 is never written
        plan.set_ati(None)
        ctx.record(11)
        return
    if two_pass:
        plan.focus_dev(raw1, s1)
        plan.focus_dev(raw2, s2)
        ctx.record(10)
        ctx.ati_dpca(s1, s2, px, 0.0, outs, want_stats=False)   # enqueue only; max|slc1| stays on the device
        ctx.record(11)
        ctx.mask_phase_frac(outs["ati_phase"], outs["slc1_mag"], px, 0.05, masked)
        return
    plan.set_max_slot(d_max)                                    # channel 1's focus leaves max|slc1| on the device
    plan.focus_dev(raw1, s1)
    plan.set_max_slot(None)
    plan.focus_dev(raw2, s2)
    ctx.record(10)
    ctx.ati_dpca_masked(s1, s2, px, 0.0, d_max, 0.05, dict(outs, ati_phase=masked))   # the phase plane comes out masked
    ctx.record(11)



for _ in range(2):
    frame()
ctx.sync()
ctx.record(0)
ati_ms = 0.0
for _ in range(frames):
    frame()
    ati_ms += ctx.elapsed_ms(10, 11)
ctx.record(1)
ms = ctx.elapsed_ms(0, 1) / frames
ati_ms /= frames
how = {"fused": "products out of channel 2's last azimuth launch", "masked": "mask inside the ATI launch", "two-pass": "mask as a further launch",
       "facade": "sarx.focus_ati_dpca on device arrays (device_output, reused workspace): products out of channel 2's last azimuth launch"}[mode]
print(f"two-channel {n_az}x{n_rg}: {ms:.3f} ms/frame = {1e3 / ms:.1f} frames/s (2 x CSA focus + ATI/DPCA + mask; {how})")
if mode == "facade":
    print(f"  the facade call, HIP events around it: {ati_ms:.3f} ms")
elif mode == "fused":
    print(f"  channel 2's whole focus with the products in its last launch: {ati_ms:.3f} ms")
else:
    print(f"  ATI/DPCA launch + reduction: {ati_ms:.3f} ms -> {28.0 * px / ati_ms / 1e6:.1f} GB/s at 28 B/pixel "
          f"= {28.0 * px / ati_ms / 1e6 / 8000.0:.3f} of the 8 TB/s HBM peak")


if json_path:
    rec = {"metric": "two-channel frames/sec (2 x CSA focus + ATI/DPCA + 5 % mask)", "value": 1e3 / ms, "unit": "frames/s", "ms_per_frame": ms,
           "config": {"workload": f"two-channel {n_az}x{n_rg} complex64, both echoes resident in HBM (BASELINE config 3)", "form": how},
           "dtype": "c64 (phase arguments f64)", "data": "synthetic"}
    if mode == "facade":
        rec["facade_call_ms_hip_events"] = ati_ms
    elif mode == "fused":
        rec["channel2_focus_with_products_ms"] = ati_ms
    else:
        rec["ati_dpca_launch"] = {"ms": ati_ms, "GBps": 28.0 * px / ati_ms / 1e6, "frac_of_8TBps": 28.0 * px / ati_ms / 1e6 / 8000.0}
    with open(json_path, "w") as fh:
        json.dump(rec, fh)
    print(json.dumps(rec))
