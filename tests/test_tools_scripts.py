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
