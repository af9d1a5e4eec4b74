#!/usr/bin/env python3
"""Wall time of the drop-in host-array call (sar_focus_csa on NumPy in, NumPy out): what a maintainer of the reference sees,
PCIe included.  python3 tools/bench_hostpath.py [size=8192]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import radar  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(0)
raw = (rng.standard_normal((n, n), dtype=np.float32) + 1j * rng.standard_normal((n, n), dtype=np.float32)).astype(np.complex64)
args = radar.focus_args(n)
for rep in range(3):
    t0 = time.perf_counter()
    img, rax, cax = sarx.sar_focus_csa(raw, *args)
    dt = time.perf_counter() - t0
    print(f"sar_focus_csa {n}x{n} host in / host out: {dt * 1e3:.1f} ms wall ({2 * raw.nbytes / dt / 1e9:.1f} GB/s over both directions)")
raw128 = raw.astype(np.complex128)
for rep in range(2):
    t0 = time.perf_counter()
    img2, _, _ = sarx.sar_focus_csa(raw128, *args)
    dt = time.perf_counter() - t0
    print(f"sar_focus_csa {n}x{n} complex128 in (the reference's dtype) / complex64 out: {dt * 1e3:.1f} ms wall")
assert np.array_equal(img, img2)
t0 = time.perf_counter()
raw128.astype(np.complex64)
print(f"  (numpy astype(complex64) of that input alone: {(time.perf_counter() - t0) * 1e3:.1f} ms)")
