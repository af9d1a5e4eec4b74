#!/usr/bin/env python3
"""The two Range-Doppler example scripts added in round 5 at the reference's own sizes, each run twice in one process (the second
run shows the steady state: plans and page-locked blocks exist), with cProfile of the second run.
    python3 tools/run_examples_fullsize.py [vehicle|moving|satellite]..."""
import cProfile
import os
import pstats
import runpy
import sys
import tempfile
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1:] or ["vehicle", "moving"]
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
runs = {"vehicle": ("sar_vehicle_rda_gpu.py", ["--out", os.path.join(tmp, "v.npz")]),
        "moving": ("sar_satellite_moving_gpu.py", ["--outdir", tmp, "--scenarios", "stationary,moving_45deg"]),
        "satellite": ("sar_satellite_rda_gpu.py", ["--out", os.path.join(tmp, "s.npz")])}
for name in which:
    script, args = runs[name]
    path = os.path.join(ROOT, "examples", script)
    for rep in range(2):
        sys.argv = [script] + args
        pr = cProfile.Profile()
        t0 = time.time()
        pr.enable()
        runpy.run_path(path, run_name="__main__")
        pr.disable()
        print(f"== {name} run {rep}: {time.time() - t0:.2f} s wall", flush=True)
        if rep == 1:
            pstats.Stats(pr).sort_stats("tottime").print_stats(12)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
