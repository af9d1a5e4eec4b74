"""The reference script's flow on the GPU (examples/sar_ati_dcpa_csa_gpu.py) at a reduced pulse count:
output schema of sar_ati_dcpa_sim_csa.py:457-461 and the physics the reference demonstrates."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_example_script_schema_and_physics(tmp_path):
    out = tmp_path / "sar_ati_dpca_data_csa.npz"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "sar_ati_dcpa_csa_gpu.py"), "--pulses", "385",
                        "--clutter", "200", "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    with np.load(out, allow_pickle=False) as z:                       # what the reference's viewer reads (:25-29)
        assert set(z.files) == {"slc1", "slc2", "range_axis", "cross_range"}
        s1, s2, rax, cax = z["slc1"], z["slc2"], z["range_axis"], z["cross_range"]
    assert s1.shape == s2.shape == (13200, 384) and rax.shape == (13200,) and cax.shape == (384,)
    assert np.isfinite(s1).all() and np.isfinite(s2).all()
    mag = np.abs(s1)
    bright = mag > 0.05 * mag.max()
    # DPCA cancels the stationary clutter far better than it cancels the 15 m/s ship (:418-419)
    resid = np.abs(s1 - s2)[bright] / mag[bright]
    assert np.median(resid) < 0.5
    # the two channels are co-registered: mean interferometric phase over the scene is small (viewer :249-250)
    assert abs(np.angle(np.sum(s1 * np.conj(s2)))) < 0.2


def test_satellite_rda_example_schema(tmp_path):
    """examples/sar_satellite_rda_gpu.py: the .npz keys and shapes of sar_satellite_sim.py:483-500; the destroyer
    focuses (image peak far above the noise/clutter floor)."""
    out = tmp_path / "sar_satellite_data.npz"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "sar_satellite_rda_gpu.py"), "--pulses", "1024",
                        "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    with np.load(out, allow_pickle=False) as z:
        assert set(z.files) == {"raw_phist", "range_comp", "rd_map", "rd_rcmc", "final_image", "range_axis", "cross_range",
                                "doppler_axis", "orbit_alt", "orbit_vel", "look_ang", "inc_ang", "bw", "r0", "fc", "v_eff"}
        img = z["final_image"]
        assert z["raw_phist"].shape == z["range_comp"].shape == z["rd_map"].shape == z["rd_rcmc"].shape == (1024, 13200)
        assert img.shape == (1024, 13200) and z["range_axis"].shape == (13200,) and z["cross_range"].shape == (1024,)
        assert z["doppler_axis"].shape == (1024,)
    assert np.isfinite(img).all()
    assert img.max() > 5 * np.median(img)


def test_vehicle_rda_example_schema(tmp_path):
    """examples/sar_vehicle_rda_gpu.py: the .npz keys and shapes of sar_vehicle_sim.py:291-307 (rd_az_comp = the eighth output of
    that script's sar_focus_rda); 4096 pulses of the airborne geometry; the destroyer focuses."""
    out = tmp_path / "sar_simulation_data.npz"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "sar_vehicle_rda_gpu.py"), "--pulses", "4096",
                        "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    with np.load(out, allow_pickle=False) as z:
        assert set(z.files) == {"raw_phist", "range_comp", "rd_map", "rd_rcmc", "rd_az_comp", "final_image", "range_axis",
                                "cross_range", "doppler_axis", "platform_alt", "platform_vel", "look_ang", "inc_ang", "r0", "prf"}
        img = z["final_image"]
        for key in ("raw_phist", "range_comp", "rd_map", "rd_rcmc", "rd_az_comp"):
            assert z[key].shape == (2048, 4096), key                 # [ranges x pulses], as the script saves them
        assert img.shape == (4096, 2048) and z["range_axis"].shape == (2048,) and z["cross_range"].shape == (4096,)
        assert float(z["prf"]) == 2000.0 and float(z["platform_vel"]) == 150.0
        filt = z["rd_az_comp"]
    assert np.isfinite(img).all() and np.isfinite(filt).all()
    assert img.max() > 5 * np.median(img)


def test_moving_ship_example_schema(tmp_path):
    """examples/sar_satellite_moving_gpu.py: one .npz per scenario with the keys of sar_satellite_moving_sim.py:337-353; the moving
    ship's image differs from the stationary one (azimuth displacement / smear of a radial mover)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "sar_satellite_moving_gpu.py"), "--pulses", "1024",
                        "--outdir", str(tmp_path), "--scenarios", "stationary,moving_0deg"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    imgs = {}
    for tag, speed in (("stationary", 0.0), ("0deg", 15.0)):
        with np.load(tmp_path / f"sar_satellite_moving_scen_{tag}.npz", allow_pickle=False) as z:
            assert set(z.files) == {"final_image", "range_axis", "cross_range", "orbit_alt", "orbit_vel", "look_ang", "inc_ang",
                                    "r0", "v_eff", "prf", "scen_name", "ship_speed", "ship_heading", "ship_vel"}
            assert z["final_image"].shape == (1024, 13200) and z["range_axis"].shape == (13200,) and z["cross_range"].shape == (1024,)
            assert float(z["ship_speed"]) == speed and z["ship_vel"].shape == (3,)
            imgs[tag] = z["final_image"]
    assert not (tmp_path / "sar_satellite_moving_scen_45deg.npz").exists()
    for im in imgs.values():
        assert np.isfinite(im).all() and im.max() > 5 * np.median(im)
    assert np.abs(imgs["stationary"] - imgs["0deg"]).max() > 0.1 * imgs["stationary"].max()


def test_batch_tdbp_example(tmp_path):
    """examples/sar_batch_gpu.py at a reduced CPI: frame stacks for both algorithms; focusing at the target's
    velocity (mBP) gives a sharper ship than the static focus (StdBP) (sar_batch_sim.py:283-286)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "sar_batch_gpu.py"), "--frames", "2", "--cpi-pulses",
                        "1000", "--nx", "128", "--headings", "45", "--outdir", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    peaks = {}
    for algo in ("mBP", "StdBP"):
        with np.load(tmp_path / f"Destroyer_15_45_{algo}.npz", allow_pickle=False) as z:
            assert z["frames"].shape == (2, 128, 128) and np.isfinite(z["frames"]).all()
            peaks[algo] = float(z["g_max"])
    assert peaks["mBP"] > 1.2 * peaks["StdBP"]


def test_batch_tdbp_example_two_ranks_equals_one(tmp_path):
    """The same batch sharded over 2 ranks (frame f -> rank f mod 2, stack reassembled by the per-round all-gather)
    writes the same stack as one rank: same kernels, same per-frame seeds."""
    common = ["--frames", "3", "--cpi-pulses", "600", "--nx", "64", "--headings", "90"]
    one, two = tmp_path / "one", tmp_path / "two"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "sar_batch_gpu.py"), *common, "--outdir", str(one)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "examples", "sar_batch_gpu.py"), *common,
                        "--outdir", str(two)], capture_output=True, text=True, timeout=600)
    # the launcher's own summary fills the tail of stderr; what a rank printed before it died is further up
    assert r.returncode == 0, r.stderr[:3000] + "\n...\n" + r.stderr[-1500:]
    for algo in ("mBP", "StdBP"):
        with np.load(one / f"Destroyer_15_90_{algo}.npz") as a, np.load(two / f"Destroyer_15_90_{algo}.npz") as b:
            assert a["frames"].shape == (3, 64, 64)
            np.testing.assert_array_equal(a["frames"], b["frames"])
