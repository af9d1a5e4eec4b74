#!/usr/bin/env python3
"""Time the echo-synthesis kernel at the reference's native size: n_targets x 7200 pulses x 13200 samples.
    python3 tools/bench_echo.py [n_targets=5000] [n_pulses=7200]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import radar  # noqa: E402

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
npulse = int(sys.argv[2]) if len(sys.argv) > 2 else 7200
k = radar.reference_constants()
ns = int(22e-6 * k["FS"])
ctx = sarx.Context(0)
rng = np.random.default_rng(0)
tau = 2 * k["R0"] / k["C"] + rng.uniform(-5e-7, 5e-7, (npulse, nt))
tp = np.stack([tau, -k["FC"] * tau], axis=2)
d_tp = ctx.to_device(tp)
d_amp = ctx.to_device(rng.uniform(1, 30, nt).astype(np.float32))
t0 = 2 * k["R0"] / k["C"] - k["T_p"] / 2 - 1e-6
d_tf = ctx.to_device(t0 + np.linspace(0, ns / k["FS"], ns))
d_raw = ctx.alloc(npulse * ns * 8)
ctx.echo_synth(d_tp, d_amp, d_tf, npulse, nt, ns, k["Kr"], k["T_p"], d_raw)
ctx.sync()
ctx.record(0)
ctx.echo_synth(d_tp, d_amp, d_tf, npulse, nt, ns, k["Kr"], k["T_p"], d_raw)
ctx.record(1)
ms = ctx.elapsed_ms(0, 1)
ts = float(nt) * npulse * ns
print(f"echo synth {nt} targets x {npulse} pulses x {ns} samples: {ms:.1f} ms = {ts / ms / 1e6:.1f} G target-samples/s")
# compute-bound: the roofline is the vector issue rate.  cycles per target-sample from the ISA (tools/isa_slots.py -> profiles/r05_isa_slots.json)
import json  # noqa: E402
try:
    isa = json.load(open(os.path.join(ROOT, "profiles", "r05_isa_slots.json")))
    kk = isa["kernels"]["echo_synth_kernel"]
    need = ts / 64.0 * kk["cycles_per_unit"]                    # SIMD cycles the launch's wave-instructions occupy
    have = isa["n_simd"] * isa["clock_hz"] * ms * 1e-3
    print(json.dumps({"roofline": {"bound": "valu_issue", "kernel": "echo_synth_kernel", "unit": "SIMD issue cycles",
                                   "cycles_per_target_sample_wave": kk["cycles_per_unit"], "valu_mix_per_4_targets": kk["loop"]["valu"],
                                   "achieved": need / (ms * 1e-3), "peak": isa["n_simd"] * isa["clock_hz"], "frac": need / have}}))
except (OSError, KeyError) as exc:
    print("no ISA table:", exc)
