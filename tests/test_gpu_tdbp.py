"""VideoSAR back-projection path on the GPU: spotlight echo synthesis and TDBP against the
reference's own outputs (tests/golden/spot_*.npz, tdbp_*.npz, written by oracle/make_golden.py from
sar_batch_sim.py:85-238) and against oracle/tdbp_oracle.py where the fixtures do not reach (native 22004-sample
pulses: two overlap-save blocks).  Tolerance: relative L2 <= 1e-4 (BASELINE.json's bar for the hot path)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import tdbp_oracle as tb
from oracle.csa_oracle import rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("tag", ["a", "b"])
def test_spotlight_echo_golden(tag):
    import sarx
    s = load_golden(f"spot_{tag}.npz")
    k = tb.constants_from_fixture(s["consts"])
    tg = [{"position": p, "rcs": r} for p, r in zip(s["tgt_pos"], s["tgt_rcs"])]
    raw, t0, n, v = sarx.run_physics_spotlight(tg, s["t_vec"], s["pos"], s["vel"], float(s["heading_deg"]),
                                               float(s["speed"]), float(s["l_ant"]), consts=k)
    assert raw.dtype == np.complex64 and raw.shape == s["raw"].shape
    assert n == int(s["num_samples"]) and t0 == float(s["t_start"])
    np.testing.assert_allclose(v, s["v_tgt"], rtol=1e-15, atol=1e-15)
    assert rel_l2(raw, s["raw"]) < TOL
    assert abs(sarx.calculate_raw_snr_db(k["R0"], 5000.0, k["Lambda"], k["BW"], float(s["l_ant"]), consts=k) - float(s["snr_db"])) < 1e-9


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("mode", ["mbp", "stdbp"])
def test_tdbp_golden(tag, mode):
    import sarx
    g = load_golden(f"tdbp_{tag}.npz")
    k = tb.constants_from_fixture(g["consts"])
    vf = g["v_tgt"] if mode == "mbp" else np.zeros(3)
    img = sarx.tdbp_gpu(g["raw"], g["pos"], g["vel"], float(g["t_start"]), int(g["num_samples"]), vf, g["t_vec"],
                        float(g["swath"]), int(g["nx"]), int(g["ny"]), consts=k)
    assert img.dtype == np.complex128 and img.shape == g["img_" + mode].shape
    assert rel_l2(img, g["img_" + mode]) < TOL


def test_range_compression_matches_oracle():
    """rc_data of sar_batch_sim.py:185 (circular correlation) read back from the plan."""
    import sarx
    g = load_golden("tdbp_a.npz")
    k = tb.constants_from_fixture(g["consts"])
    n_p, n_s = g["raw"].shape
    plan = sarx.TdbpPlan(sarx.default_context(), n_p, n_s, 8, 8, k)
    _, rc = plan.focus(g["raw"], g["pos"], g["vel"], float(g["t_start"]), np.zeros(3), g["t_vec"], 100.0, want_rc=True)
    plan.close()
    assert rel_l2(rc, tb.range_compress(g["raw"], n_s, k)) < 1e-5


def test_tdbp_native_pulse_length_two_blocks():
    """The script's own waveform: 22004 samples, 12000-tap chirp -> two overlap-save blocks of the 32768-point FFT."""
    import sarx
    k = tb.batch_constants()
    sc = tb.tdbp_scene(n_pulses=24, seed=11, k=k, speed=15.0, heading_deg=45.0, swath=200.0, n_targets=3)
    assert sc["num_samples"] == 22004
    raw = sc["raw"].astype(np.complex64)
    img = sarx.tdbp_gpu(raw, sc["pos"], sc["vel"], sc["t_start"], sc["num_samples"], sc["v_tgt"], sc["t_vec"], sc["swath"],
                        nx=20, ny=18, consts=k)
    ref = tb.tdbp(raw, sc["pos"], sc["vel"], sc["t_start"], sc["num_samples"], sc["v_tgt"], sc["t_vec"], sc["swath"], 20, 18, k)
    assert rel_l2(img, ref) < TOL


def test_windowed_compression_equals_full():
    """Only the samples the scene can touch are compressed (a geometric bound); the image equals the one from
    fully compressed pulses (forced by asking for rc_data), and the native window is a small part of the pulse."""
    import sarx
    k = tb.batch_constants()
    sc = tb.tdbp_scene(n_pulses=40, seed=4, k=k, speed=15.0, heading_deg=120.0, swath=500.0, n_targets=4)
    raw = sc["raw"].astype(np.complex64)
    plan = sarx.TdbpPlan(sarx.default_context(), 40, sc["num_samples"], 32, 32, k)
    args = (sc["pos"], sc["vel"], sc["t_start"], sc["v_tgt"], sc["t_vec"], sc["swath"])
    win = plan.focus(raw, *args)
    lo, hi = plan.last_window()
    full, rc = plan.focus(raw, *args, want_rc=True)
    assert plan.last_window() == (0, sc["num_samples"])
    plan.close()
    assert 0 < lo < hi < sc["num_samples"] and hi - lo < 2500
    assert rel_l2(win, full) < 1e-5
    # every sample the reference's interpolation reads lies inside the window: outside it the pulse energy is irrelevant
    ref = tb.tdbp(raw, sc["pos"], sc["vel"], sc["t_start"], sc["num_samples"], sc["v_tgt"], sc["t_vec"], sc["swath"], 32, 32, k)
    assert rel_l2(win, ref) < TOL


def test_device_resident_chain_and_chunking():
    """run_physics_spotlight(device=True) -> tdbp_gpu without a host round trip equals the host-buffer chain;
    160 pulses x 40 x 40 pixels exercises several pulse chunks."""
    import sarx
    k = tb.scaled_constants()
    sc = tb.tdbp_scene(n_pulses=160, seed=2, k=k)
    args = (sc["targets"], sc["t_vec"], sc["pos"], sc["vel"], sc["heading_deg"], sc["speed"], sc["l_ant"])
    d_raw, t0, n, v = sarx.run_physics_spotlight(*args, consts=k, device=True)
    raw, _, _, _ = sarx.run_physics_spotlight(*args, consts=k)
    a = sarx.tdbp_gpu(d_raw, sc["pos"], sc["vel"], t0, n, v, sc["t_vec"], sc["swath"], 40, 40, consts=k)
    b = sarx.tdbp_gpu(raw, sc["pos"], sc["vel"], t0, n, v, sc["t_vec"], sc["swath"], 40, 40, consts=k)
    d_raw.release()
    assert np.array_equal(a, b)
    ref = tb.tdbp(raw, sc["pos"], sc["vel"], t0, n, v, sc["t_vec"], sc["swath"], 40, 40, k)
    assert rel_l2(a, ref) < TOL
    # the mover is focused at its own velocity: the peak stands well above a static-scene focus
    c = sarx.tdbp_gpu(raw, sc["pos"], sc["vel"], t0, n, np.zeros(3), sc["t_vec"], sc["swath"], 40, 40, consts=k)
    assert np.abs(a).max() > 1.5 * np.abs(c).max()


@pytest.mark.parametrize("speed,heading", [(15.0, 45.0), (0.0, 0.0)])
def test_tile_expansion_equals_exact_kernel(speed, heading, monkeypatch):
    """Metre-sized pixels at the script's own geometry (sar_batch_sim.py:12-50: 500 km range) take the kernel that expands the
    geometry about each 16 x 16 tile's reference pixel; SARX_TDBP_TILE=0 forces the exact per-pixel kernel.  320 pulses cross a
    256-pulse batch and several chunks; 90 x 70 pixels leave ragged tiles.  Both against each other and against the oracle."""
    import sarx
    k = tb.batch_constants()
    sc = tb.tdbp_scene(n_pulses=320, seed=21, k=k, speed=speed, heading_deg=heading, swath=90.0, n_targets=4)
    raw = sc["raw"].astype(np.complex64)
    args = (raw, sc["pos"], sc["vel"], sc["t_start"], sc["num_samples"], sc["v_tgt"], sc["t_vec"], sc["swath"])
    tile = sarx.tdbp_gpu(*args, nx=90, ny=70, consts=k)
    monkeypatch.setenv("SARX_TDBP_TILE", "0")
    exact = sarx.tdbp_gpu(*args, nx=90, ny=70, consts=k)
    assert not np.array_equal(tile, exact)                       # two different kernels ran
    assert rel_l2(tile, exact) < 2e-6
    ref = tb.tdbp(raw, sc["pos"], sc["vel"], sc["t_start"], sc["num_samples"], sc["v_tgt"], sc["t_vec"], sc["swath"], 90, 70, k)
    assert rel_l2(tile, ref) < TOL and rel_l2(exact, ref) < TOL


def test_range_compression_one_launch_equals_three(monkeypatch):
    """SARX_TDBP_RC_FUSED=0 keeps the form of rounds 2-4 (copy-in, forward transform with the reference spectrum, inverse transform,
    copy-out per overlap-save block) for A/B; the one-launch form (wrapped segment in, FFT . spectrum . IFFT in registers / LDS) must
    give the same range-compressed pulses and image."""
    import sarx
    k = tb.batch_constants()
    sc = tb.tdbp_scene(n_pulses=48, seed=8, k=k, speed=15.0, heading_deg=60.0, swath=300.0, n_targets=3)
    raw = sc["raw"].astype(np.complex64)
    args = (sc["pos"], sc["vel"], sc["t_start"], sc["v_tgt"], sc["t_vec"], sc["swath"])
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SARX_TDBP_RC_FUSED", mode)
        plan = sarx.TdbpPlan(sarx.default_context(), 48, sc["num_samples"], 24, 24, k)      # the switch is read when a plan is created
        out[mode] = plan.focus(raw, *args)
        plan.close()
    assert rel_l2(out["1"], out["0"]) < 2e-6
    ref = tb.tdbp(raw, sc["pos"], sc["vel"], sc["t_start"], sc["num_samples"], sc["v_tgt"], sc["t_vec"], sc["swath"], 24, 24, k)
    assert rel_l2(out["1"], ref) < TOL


def test_tdbp_errors():
    import sarx
    k = tb.scaled_constants()
    ctx = sarx.default_context()
    with pytest.raises(sarx.SarxError):
        sarx.TdbpPlan(ctx, 0, 128, 8, 8, k)
    bad = dict(k); bad["FS"] = 0.0
    with pytest.raises(sarx.SarxError):
        sarx.TdbpPlan(ctx, 8, 128, 8, 8, bad)
    plan = sarx.TdbpPlan(ctx, 8, 128, 8, 8, k)
    with pytest.raises(ValueError):
        plan.focus(np.zeros((8, 64), np.complex64), np.zeros((8, 3)), np.zeros((8, 3)), 0.0, np.zeros(3), np.zeros(8), 10.0)
    plan.close()
