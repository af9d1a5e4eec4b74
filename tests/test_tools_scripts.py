"""Every shell script under tools/ must at least parse (`bash -n`) before it is sent to the GPU box: in round 4 two calls were lost
to an unterminated quote and an empty path in one-shot call scripts.  The one runner that replaced them is also run with a
harmless step list in a scratch directory."""
import glob
import os
import subprocess

import pytest

from conftest import ROOT

SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.basename(p) for p in SCRIPTS])
def test_script_parses(path):
    r = subprocess.run(["bash", "-n", path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_runner_is_the_only_call_script():
    assert os.path.join(ROOT, "tools", "gpu_call.sh") in SCRIPTS
    assert not glob.glob(os.path.join(ROOT, "tools", "gpu_r[0-9]*.sh")), "per-call transcripts belong in profiles/MANIFEST.md, not tools/"


def test_runner_steps_manifest_and_failure_stop(tmp_path):
    """`env`, an unknown step and the failure stop, without a GPU: the runner writes its manifest, runs steps in order, and a failing
    step ends the call (no further step after a failure)."""
    env = dict(os.environ, GRAFT_REPO_ROOT=str(tmp_path))
    os.makedirs(tmp_path / "tools")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "gpu_call.sh"), "t1", "env", "A=1", "B=2", "--", "summary"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    man = (tmp_path / "gpurun_out" / "t1" / "MANIFEST.txt").read_text()
    assert "env A=1 B=2" in man and "summary" in man and "rc 0" in man
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "gpu_call.sh"), "t2", "nosuchstep", "--", "summary"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "STEP FAILED" in r.stdout
    assert "summary" not in (tmp_path / "gpurun_out" / "t2" / "MANIFEST.txt").read_text()
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "gpu_call.sh")], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 9


def test_corner_turn_issues_its_loads_together():
    """Regression guard read off the ISA (tools/isa_load_waits.py, CPU only: hipcc cross-compiles): the corner-turn kernel's sixteen
    loads per thread are issued before the first wait.  Written as `if (inside) tile[..] = in[..]` the compiler waited for every load
    inside its own predicated block (16 of 16): 0.986 against 0.833-0.864 ms at 16384^2."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_load_waits
    rows = isa_load_waits.census(os.path.join(ROOT, "nis-sar-amtigmti-video_amd", "csrc", "products.hip"))
    ct = [r for r in rows if "corner_turn_kernel" in r[0]]
    assert len(ct) == 1
    name, loads, stores, waits, waits0, serial = ct[0]
    assert loads == 16 and stores == 16
    assert serial <= 2 and waits <= 3, ct[0]


def test_trace_busy_summarises_a_kernel_trace(tmp_path):
    """tools/trace_busy.py on a synthetic rocprofv3 kernel trace: two overlapping lanes, one idle gap inside the loop and one long gap
    between two phases of the script (not counted as idle time of a loop)."""
    rows = ["Kernel_Name,Start_Timestamp,End_Timestamp"]
    t = 1_000_000
    for phase in range(2):
        for i in range(60):
            rows.append(f"\"void k_a(int)\",{t},{t + 900}")
            rows.append(f"\"void k_b(int)\",{t + 500},{t + 1500}")
            t += 1500 + (100 if i % 10 == 0 else 0)
        t += 5_000_000
    p = tmp_path / "trace.csv"
    p.write_text("\n".join(rows) + "\n")
    r = subprocess.run([os.environ.get("PYTHON", "python3"), os.path.join(ROOT, "tools", "trace_busy.py"), str(p)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "between phases" in r.stdout and "GPU busy 0.9" in r.stdout
    assert "k_b" in r.stdout and "->" in r.stdout
