"""The C ABI from a C program (tests/c_client/client.c, built by `make` with gcc against include/sarx.h and
libsarx.so only): same image as the oracle, error codes instead of fallbacks."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import csa_oracle as orc

CLIENT = os.path.join(ROOT, "tests", "c_client", "client")


def test_c_client_builds_and_links():
    """not gpu: the client exists (make built it) and resolves every symbol it uses (usage exit, no GPU call)."""
    assert os.path.exists(CLIENT), "run `make -C nis-sar-amtigmti-video_amd/csrc` (or __graft_entry__.build())"
    r = subprocess.run([CLIENT], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "usage" in r.stderr


@pytest.mark.gpu
def test_c_client_focus_matches_oracle(tmp_path):
    raw, k = orc.point_scene(128, 256, seed=21, clutter_db=-25.0)
    args = orc.focus_args(k)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    np.ascontiguousarray(raw, dtype=np.complex64).tofile(fin)
    r = subprocess.run([CLIENT, str(fin), "128", "256"] + [repr(float(v)) for v in args] + [str(fout)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().endswith("ok")
    img = np.fromfile(fout, dtype=np.complex64).reshape(128, 256)
    ref, rax, cax = orc.sar_focus_csa(raw, *args)
    assert orc.rel_l2(img.T, ref) < 1e-4
    m = re.search(r"bad_plan_rc (-?\d+) msg \"(.*)\"", r.stdout)
    assert m and int(m.group(1)) < 0 and len(m.group(2)) > 5
    m = re.search(r"lanes_bit_identical (\d) lane_out_of_range_rc (-?\d+)", r.stdout)      # two frames in flight, page-locked download
    assert m and m.group(1) == "1" and int(m.group(2)) < 0
    m = re.search(r"pipeline_bit_identical (\d) third_pending_rc (-?\d+) end_twice_rc (-?\d+)", r.stdout)    # sarx_csa_focus_host_begin / _end
    assert m and m.group(1) == "1" and int(m.group(2)) < 0 and int(m.group(3)) < 0
    m = re.search(r"range_axis (\S+) (\S+) cross_range (\S+) (\S+)", r.stdout)
    got = [float(x) for x in m.groups()]
    np.testing.assert_allclose(got, [rax[0], rax[-1], cax[0], cax[-1]], rtol=1e-9, atol=1e-6)
