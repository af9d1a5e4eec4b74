"""GPU echo synthesis (drop-ins for run_physics_engine / run_bistatic_physics_gpu) against the
fixtures produced by the reference's own functions, and against the oracle on a focusable scene."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import csa_oracle as orc

pytestmark = pytest.mark.gpu


def _targets(g):
    return [{"position": p, "rcs": r} for p, r in zip(g["tgt_pos"], g["tgt_rcs"])]


def test_monostatic_matches_reference_fixture():
    import sarx
    g = load_golden("echo_mono.npz")
    raw, t0, fs = sarx.run_physics_engine(_targets(g), g["pos_sat"], np.zeros(len(g["pos_sat"])),
                                          BW=float(g["BW"]), T_p=float(g["T_p"]), FC=float(g["FC"]))
    assert raw.shape == g["raw"].shape and raw.dtype == np.complex64
    assert t0 == float(g["t_start_fast"]) and fs == float(g["fs"])
    assert orc.rel_l2(raw, g["raw"]) < 2e-6
    np.testing.assert_array_equal(raw == 0, g["raw"] == 0)          # the pulse gate switches on the same samples


def test_bistatic_matches_reference_fixture():
    import sarx
    g = load_golden("echo_bistatic.npz")
    raw, t0 = sarx.run_bistatic_physics_gpu(_targets(g), g["t_vec"], g["pos_tx"], g["vel_tx"], float(g["rx_offset"]),
                                            g["vel_target"], FS=float(g["fs"]), BW=float(g["BW"]), T_p=float(g["T_p"]),
                                            FC=float(g["FC"]))
    assert raw.shape == g["raw"].shape
    assert t0 == float(g["t_start_fast"])
    assert orc.rel_l2(raw, g["raw"]) < 2e-6


def test_moving_and_vehicle_variants_match_reference_fixtures():
    """run_moving_physics (sar_satellite_moving_sim.py:111-159) and run_custom_physics (sar_vehicle_sim.py:83-128)."""
    import sarx
    g = load_golden("echo_moving.npz")
    raw, t0, fs = sarx.run_moving_physics(_targets(g), g["t_vec"], g["pos_sat"], g["vel_target"], BW=float(g["BW"]),
                                          T_p=float(g["T_p"]), FC=float(g["FC"]), R0=float(g["R0"]))
    assert raw.shape == g["raw"].shape and t0 == float(g["t_start_fast"]) and fs == float(g["fs"])
    assert orc.rel_l2(raw, g["raw"]) < 2e-6
    np.testing.assert_array_equal(raw == 0, g["raw"] == 0)
    g = load_golden("echo_vehicle.npz")
    raw = sarx.run_custom_physics(_targets(g), np.arange(len(g["pos"])) / 1000.0, g["pos"], 1e-3, float(g["t_p"]), float(g["fc"]),
                                  float(g["bw"]), R0=float(g["R0"]))
    assert raw.shape == g["raw"].shape and raw.dtype == np.complex64
    assert orc.rel_l2(raw, g["raw"]) < 2e-6


def test_many_targets_and_focus_chain():
    """600 scatterers (more than one LDS chunk) x 256 pulses; then GPU echo -> GPU focus equals
    oracle echo -> oracle focus."""
    import sarx
    k = orc.scaled_radar(256, 512)
    rng = np.random.default_rng(8)
    tg = [{"position": [rng.uniform(-20, 20), rng.uniform(-20, 20), 0.0], "rcs": float(rng.uniform(1, 50))}
          for _ in range(600)]
    t_vec = np.linspace(-128 / k["PRF"], 128 / k["PRF"], 256)
    pos, vel = orc.orbit_track(t_vec, k)
    n_rg = 512
    ref = orc.echo_monostatic(tg, pos, n_rg, k["FS"], 2 * k["R0"] / k["C"] - k["T_p"] / 2 - 1e-6, k["FC"], k["Kr"], k["T_p"])
    raw, t0, fs = sarx.run_physics_engine(tg, pos, t_vec, BW=k["BW"], T_p=k["T_p"], fs=k["FS"], window_sec=n_rg / k["FS"])
    assert raw.shape == (256, 512)
    assert orc.rel_l2(raw, ref) < 5e-6
    args = (k["Lambda"], k["T_p"], k["Kr"], fs, k["PRF"], k["V_eff"], k["R0"], t0)
    img = sarx.sar_focus_csa(raw, *args)[0]
    oimg = orc.sar_focus_csa(ref.astype(np.complex64), *args)[0]
    assert orc.rel_l2(np.abs(img), np.abs(oimg)) < 1e-4


def test_device_resident_two_channel_chain():
    """Echoes synthesised with device=True, a second target set added in place (add_to), the DPCA pulse shift as two
    views and focus_ati_dpca on DeviceArrays: identical to the same chain through host arrays."""
    import sarx
    k = orc.scaled_radar(129, 2048)
    rng = np.random.default_rng(3)
    ship = [{"position": [rng.uniform(-10, 10), rng.uniform(-10, 10), 0.0], "rcs": float(rng.uniform(5, 50))} for _ in range(5)]
    sea = [{"position": [rng.uniform(-30, 30), rng.uniform(-30, 30), 0.0], "rcs": float(rng.uniform(0.1, 2))} for _ in range(300)]
    t_vec = np.linspace(-64 / k["PRF"], 64 / k["PRF"], 129)
    pos, vel = orc.orbit_track(t_vec, k)
    kw = dict(FS=k["FS"], BW=k["BW"], T_p=k["T_p"], R0=k["R0"], FC=k["FC"], window_sec=2048 / k["FS"])    # the 1 us lead-in must fit
    dev, host = [], []
    for off in (-1.0, 1.0):
        d, t0 = sarx.run_bistatic_physics_gpu(ship, t_vec, pos, vel, off, [12.0, 0.0, 0.0], device=True, **kw)
        same, _ = sarx.run_bistatic_physics_gpu(sea, t_vec, pos, vel, off, [0.0, 0.0, 0.0], add_to=d, **kw)
        assert same is d and d.shape == (129, 2048)
        h = (sarx.run_bistatic_physics_gpu(ship, t_vec, pos, vel, off, [12.0, 0.0, 0.0], **kw)[0]
             + sarx.run_bistatic_physics_gpu(sea, t_vec, pos, vel, off, [0.0, 0.0, 0.0], **kw)[0])
        assert orc.rel_l2(d.numpy(), h) < 1e-6
        dev.append(d)
        host.append(d.numpy())
    args = (k["Lambda"], k["T_p"], k["Kr"], k["FS"], k["PRF"], k["V_eff"], k["R0"], t0)
    one = sarx.sar_focus_csa(dev[0], *args)[0]               # single channel, device input = host input
    np.testing.assert_array_equal(one, sarx.sar_focus_csa(host[0], *args)[0])
    a = sarx.focus_ati_dpca(dev[0], dev[1], *args)
    b = sarx.focus_ati_dpca(host[0], host[1], *args)
    for d in dev:
        d.release()
    assert a["slc1"].shape == (2048, 128) and np.abs(a["slc1"]).max() > 0
    for key in ("slc1", "slc2", "slc1_mag", "dpca_mag", "ati_phase_masked"):
        np.testing.assert_array_equal(a[key], b[key])
