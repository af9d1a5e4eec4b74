#!/usr/bin/env python3
"""Print GPU-vs-oracle relative L2 errors for a list of scene sizes (GPU box only).
    python tests/parity_report.py [size ...]      sizes as NAZxNRG, default a small ladder
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: F401,E402  (sets sys.path)
from oracle import csa_oracle as orc  # noqa: E402
import sarx  # noqa: E402

sizes = sys.argv[1:] or ["256x256", "1024x1024", "2048x2048", "4096x4096"]
for s in sizes:
    n_az, n_rg = (int(x) for x in s.split("x"))
    raw, k = orc.point_scene(n_az, n_rg, seed=1, clutter_db=-20.0, n_targets=7)
    args = orc.focus_args(k)
    t = time.time()
    ref = orc.sar_focus_csa_lean(raw, *args, workers=8)[0]
    tc = time.time() - t
    for fuse in (True, False):
        img = sarx.sar_focus_csa(raw, *args, fuse_range=fuse)[0]
        print(f"{s:>12s} fuse={int(fuse)}  |img| rel-L2 {orc.rel_l2(np.abs(img), np.abs(ref)):.3e}  "
              f"complex rel-L2 {orc.rel_l2(img, ref):.3e}  (oracle {tc:.1f}s)", flush=True)
