#!/usr/bin/env python3
"""Range-Doppler focus (sar_focus_rda, sar_satellite_sim.py:356-448) at the satellite scripts' own 13200 x 7200, pulses resident on the
device: HIP-event time of sarx_rda_focus_dev (six launches), optionally with two focuses in flight on two lanes.
    python3 tools/bench_rda.py [n_ranges=13200] [n_pulses=7200] [iters=10] [lanes=1] [radar=satellite|vehicle]
radar=vehicle: the airborne script's radar (sar_vehicle_sim.py:21-40,166-168,283-287: 10 GHz, 300 MHz over 1 us, 360 MHz sampling,
2 kHz PRF, 150 m/s, R0 = 20 km / cos 45 deg), whose own size is 2048 ranges x 32768 pulses."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import _ffi, rda  # noqa: E402
from sarx._ffi import check  # noqa: E402

n_r = int(sys.argv[1]) if len(sys.argv) > 1 else 13200
n_p = int(sys.argv[2]) if len(sys.argv) > 2 else 7200
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
which = sys.argv[5] if len(sys.argv) > 5 else "satellite"
from sarx import radar  # noqa: E402
ctx = sarx.default_context()
if which == "vehicle":
    prm = _ffi.RadarParams(radar.C / 10e9, 1.0e-6, 300e6 / 1.0e-6, 360e6, 2000.0, 150.0, 20000.0 / np.cos(np.radians(45.0)), 0.0)
else:
    kk = radar.reference_constants()                       # the satellite scripts' radar (sar_satellite_sim.py:18-38)
    prm = _ffi.RadarParams(kk["Lambda"], kk["T_p"], kk["Kr"], kk["FS"], kk["PRF"], kk["V_eff"], kk["R0"], 0.0)
plans = [rda.RdaPlan(ctx, n_r, n_p, prm) for _ in range(lanes)]
d_in = ctx.alloc(n_p * n_r * 8)
ctx.fill_noise(d_in, n_p * n_r, 3)
d_mag = [ctx.alloc(n_p * n_r * 4) for _ in range(lanes)]


def run(i):
    ctx.select_lane(i % lanes)
    check(ctx.lib.sarx_rda_focus_dev(plans[i % lanes].h, d_in.ptr, d_mag[i % lanes].ptr, None, None, None), ctx.h)


for i in range(2 * lanes):
    run(i)
ctx.sync()
t0 = time.perf_counter()
for i in range(iters):
    run(i)
ctx.sync()
ms = (time.perf_counter() - t0) / iters * 1e3
ctx.select_lane(0)
m = d_mag[0].download(np.float32, (8, n_r))
assert np.isfinite(m).all() and m.max() > 0
print(f"sar_focus_rda {n_r} x {n_p} ({which} radar), device resident, {lanes} focus(es) in flight, SARX_CONV_PLANES={os.environ.get('SARX_CONV_PLANES', '0')}: "
      f"{ms:.3f} ms per focus ({1e3 / ms:.1f} frames/s)")
