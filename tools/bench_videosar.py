#!/usr/bin/env python3
"""Time the reference's VideoSAR frame (sar_batch_sim.py:303-331) at its native size on one GPU:
destroyer target, 2500-pulse CPI (0.5 s at PRF 5 kHz), 22004 samples, 512 x 512 back-projection image,
and the Range-Doppler focuser at sar_satellite_sim.py's native 13200 x 7200.
    python3 tools/bench_videosar.py [n_pulses=2500] [nx=512]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx.targets import generate_destroyer  # noqa: E402

n_p = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 512
k = sarx.batch_constants()
tg = generate_destroyer(center_pos=(0, 0, 0))
t_vec = np.linspace(-2.5, 2.5, 25000)[:n_p]
pos, vel = sarx.orbit_arc(t_vec, k)
l_ant = k["Lambda"] * k["R0"] / 500.0
ctx = sarx.default_context()
for rep in range(2):
    t0 = time.perf_counter()
    d_raw, t_start, n_s, v_tgt = sarx.run_physics_spotlight(tg, t_vec, pos, vel, 45.0, 15.0, l_ant, consts=k, device=True)
    ctx.sync()
    t1 = time.perf_counter()
    img = sarx.tdbp_gpu(d_raw, pos, vel, t_start, n_s, v_tgt, t_vec, 500.0, nx, nx, consts=k)
    t2 = time.perf_counter()
    d_raw.release()
print(f"spotlight echo {len(tg)} targets x {n_p} x {n_s}: {1e3 * (t1 - t0):.1f} ms wall; "
      f"tdbp {nx}x{nx} (range compression + back-projection + download): {1e3 * (t2 - t1):.1f} ms wall; "
      f"peak {np.abs(img).max():.4g}")

# Range-Doppler focus, native sar_satellite_sim.py size
from sarx import radar  # noqa: E402
kk = radar.reference_constants()
n_r, n_az = int(22e-6 * kk["FS"]), 7200
rng = np.random.default_rng(0)
raw = (rng.standard_normal((n_az, n_r), dtype=np.float32) + 1j * rng.standard_normal((n_az, n_r), dtype=np.float32)).astype(np.complex64)
args = (kk["Lambda"], kk["T_p"], kk["Kr"], kk["FS"], kk["PRF"], kk["V_eff"], kk["R0"])
for rep in range(6):          # steady state: results of a size come from the page-locked pool from its fourth request on
    t0 = time.perf_counter()
    out = sarx.sar_focus_rda(raw.T, *args, intermediates=False)
    t1 = time.perf_counter()
    del out
print(f"sar_focus_rda {n_r} x {n_az} host in / magnitude out: {1e3 * (t1 - t0):.1f} ms wall (sixth call)")

# the same with the pulses already on the GPU (as the echo kernels leave them): focus + magnitude download only
d = sarx.DeviceArray(ctx.to_device(raw), raw.shape)
for rep in range(6):
    t0 = time.perf_counter()
    out = sarx.sar_focus_rda(d.T, *args, intermediates=False)
    t1 = time.perf_counter()
    del out
d.release()
print(f"sar_focus_rda {n_r} x {n_az} device in / magnitude out: {1e3 * (t1 - t0):.1f} ms wall (sixth call)")
