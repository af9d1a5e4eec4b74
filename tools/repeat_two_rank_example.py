#!/usr/bin/env python3
"""The two-rank run of the VideoSAR example (what tests/test_gpu_example.py::test_batch_tdbp_example_two_ranks_equals_one starts) a few
times in a row, stopping at the first failure and printing the head of its stderr: a check of the clean shutdown added after rank 1
aborted once at the end of the script (no barrier / destroy_process_group).   python3 tools/repeat_two_rank_example.py [n=6]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
common = ["--frames", "3", "--cpi-pulses", "600", "--nx", "64", "--headings", "90"]
for i in range(n):
    out = tempfile.mkdtemp()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29540 + i), os.path.join(ROOT, "examples", "sar_batch_gpu.py"), *common, "--outdir", out],
                       capture_output=True, text=True, timeout=300)
    print(f"run {i}: exit {r.returncode}", flush=True)
    if r.returncode != 0:
        print(r.stderr[:4000])
        sys.exit(1)
print("all clean")
