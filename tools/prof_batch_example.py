#!/usr/bin/env python3
"""Host-side profile of the VideoSAR example's frame loop (examples/sar_batch_gpu.py, one heading, both focus velocities): where does the host spend a frame once the kernels of a frame take 2.4 ms?  (It found the seven per-frame table uploads, each of which
waited for EVERY lane: 0.33 ms per upload.)   python3 tools/prof_batch_example.py"""
import cProfile
import os
import pstats
import runpy
import sys
import tempfile

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = tempfile.mkdtemp()
sys.argv = ["sar_batch_gpu.py", "--headings", "90", "--outdir", out]
path = os.path.join(ROOT, "examples", "sar_batch_gpu.py")
runpy.run_path(path, run_name="__main__")          # warm: plans, page-locked blocks
pr = cProfile.Profile()
pr.enable()
runpy.run_path(path, run_name="__main__")
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(25)
