"""CPU-side checks of the drop-in boundary: libsarx.so loads, exports every
symbol include/sarx.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _header_symbols():
    with open(os.path.join(ROOT, "include", "sarx.h")) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sarx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from sarx import _ffi
    syms = _header_symbols()
    assert len(syms) >= 30
    assert set(syms) == set(_ffi.SIGNATURES), set(syms) ^ set(_ffi.SIGNATURES)


def test_library_exports_every_symbol():
    from sarx import _ffi
    lib = _ffi.load()
    for s in _header_symbols():
        assert hasattr(lib, s), s
    assert lib.sarx_version() == 206


def test_struct_layouts():
    from sarx import _ffi
    assert C.sizeof(_ffi.RadarParams) == 8 * 8
    assert C.sizeof(_ffi.AtiOutputs) == 9 * C.sizeof(C.c_void_p)


def test_persistent_grid_is_per_device_arithmetic():
    """The persistent range kernels size their grid from the compute-unit count of the ctx being launched on
    (sarx_persistent_grid), not from a process-wide cache: two devices with different CU counts get different grids."""
    from sarx import _ffi
    g = _ffi.load().sarx_persistent_grid
    assert g(1, 256, 16384) == 256 and g(2, 256, 16384) == 512       # MI355X: fused range kernel, 32-point kernel
    assert g(1, 304, 16384) == 304 and g(1, 64, 16384) == 64         # another device in the same process
    assert g(2, 256, 64) == 64 and g(1, 256, 1) == 1                 # never more workgroups than lines
    assert g(0, 0, 0) == 1 and g(-3, 256, 8) == 8                    # degenerate arguments still launch one workgroup


def test_rccl_is_taken_from_the_hip_runtime_in_use():
    """libsarx resolves librccl next to the HIP runtime the process runs on (torch's bundle after `import torch`,
    /opt/rocm otherwise) and reports which file and version it got."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); {pre}import sarx, json; "
            "print(json.dumps(sarx.Context.rccl_info()))" % os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
    import json
    plain = json.loads(subprocess.run([sys.executable, "-c", code.format(pre="")], capture_output=True, text=True,
                                      check=True).stdout.strip().splitlines()[-1])
    assert os.path.exists(plain["path"]) and "torch" not in plain["path"]
    assert plain["version"] == plain["header_version"] > 20000         # /opt/rocm's RCCL is what the headers describe
    with_torch = json.loads(subprocess.run([sys.executable, "-c", code.format(pre="import torch; ")],
                                           capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    assert os.path.dirname(with_torch["path"]).endswith(os.path.join("torch", "lib"))
    assert with_torch["version"] // 10000 == with_torch["header_version"] // 10000    # same major: the calls used are ABI-stable


def _have_gpu():
    from sarx import _ffi
    n = C.c_int()
    return _ffi.load().sarx_device_count(C.byref(n)) == 0 and n.value > 0


def test_no_cpu_fallback():
    """Without a GPU the product path raises; it never routes to the oracle."""
    if _have_gpu():
        pytest.skip("GPU present")
    import sarx
    with pytest.raises(sarx.SarxError) as e:
        sarx.sar_focus_csa(np.zeros((64, 64), np.complex64), 0.03, 1e-6, 1e12, 6e8, 6e3, 7e3, 5e5, 3e-3)
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)
    with pytest.raises(sarx.SarxError):
        sarx.ati_dpca(np.zeros((4, 4), np.complex64), np.zeros((4, 4), np.complex64))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "nis-sar-amtigmti-video_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                with open(os.path.join(dirpath, f)) as fh:
                    src = fh.read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def test_target_model_and_orbit_match_reference():
    """Host helpers the example script needs: destroyer scatterers against the reference's own
    vehicle_targets.generate_destroyer output, orbit track against the oracle."""
    from conftest import load_golden
    from oracle import csa_oracle as orc
    from sarx import radar, targets
    g = load_golden("destroyer.npz")
    for model in (targets.generate_destroyer(tuple(g["center"])), orc.destroyer_targets(tuple(g["center"]))):
        assert len(model) == 35
        np.testing.assert_array_equal(np.array([t["position"] for t in model], dtype=np.float64), g["pos"])
        np.testing.assert_array_equal(np.array([t["rcs"] for t in model], dtype=np.float64), g["rcs"])
    t = np.linspace(-0.6, 0.6, 13)
    a, b = radar.orbit_track(t), orc.orbit_track(t)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    k1, k2 = radar.reference_constants(), orc.reference_radar_constants()
    for key in ("V_sat", "R0", "Lambda", "V_eff", "Kr", "d_rx"):
        assert abs(k1[key] - k2[key]) <= 1e-12 * abs(k2[key])
