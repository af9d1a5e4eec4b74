"""Pin the CPU oracle to the reference: every fixture under tests/golden was
produced by the reference's own functions (oracle/make_golden.py)."""
import numpy as np
import pytest

from oracle import csa_oracle as orc
from conftest import load_golden

CSA_FIXTURES = ["csa_64x64", "csa_96x80", "csa_128x128", "csa_256x256", "csa_128x512", "csa_512x128",
                "csa_refconst_128x256"]


@pytest.mark.parametrize("tag", CSA_FIXTURES)
def test_focus_matches_reference(tag):
    g = load_golden(tag + ".npz")
    img_t, rax, cax = orc.sar_focus_csa(g["phist"], *g["args"])
    assert img_t.shape == g["img_T"].shape
    assert orc.rel_l2(img_t, g["img_T"]) < 1e-12
    np.testing.assert_allclose(rax, g["range_axis"], rtol=1e-15)
    np.testing.assert_allclose(cax, g["cross_range_axis"], rtol=1e-13, atol=1e-12)


@pytest.mark.parametrize("tag", ["csa_96x80", "csa_256x256", "csa_refconst_128x256"])
def test_lean_path_equals_literal(tag):
    g = load_golden(tag + ".npz")
    a = orc.sar_focus_csa(g["phist"], *g["args"])[0]
    b = orc.sar_focus_csa_lean(g["phist"], *g["args"], block=40)[0]
    assert orc.rel_l2(b, a) < 1e-12
    assert orc.rel_l2(b, g["img_T"]) < 1e-12


def test_stages_are_consistent():
    g = load_golden("csa_128x128.npz")
    img_t, _, _, (s1, s2, s3, s4) = orc.sar_focus_csa(g["phist"], *g["args"], return_stages=True)
    assert orc.rel_l2(s4.T, g["img_T"]) < 1e-12
    # stage 3 -> 4 is a plain inverse FFT along azimuth in natural order
    assert orc.rel_l2(np.fft.ifft(s3, axis=0), s4) < 1e-13
    assert s1.shape == s2.shape == s3.shape == (128, 128)


@pytest.mark.parametrize("tag", ["csa_96x80", "csa_128x512", "csa_refconst_128x256"])
def test_sampled_row_and_column_helpers_compose_to_the_reference(tag):
    """The per-row / per-column restatements used at 16384^2 and 8192^2, chained over every row and column (in
    shuffled order, in pieces), give the reference's own image."""
    g = load_golden(tag + ".npz")
    raw, args = g["phist"], g["args"]
    n_az, n_rg = raw.shape
    rng = np.random.default_rng(0)
    s1 = np.empty((n_az, n_rg), np.complex128)
    for cols in np.array_split(rng.permutation(n_rg), 3):
        s1[:, cols] = orc.azimuth_fft_cols(raw[:, cols], cols, n_rg, *args)
    s3 = np.empty_like(s1)
    for bins in np.array_split(rng.permutation(n_az), 4):
        s3[bins] = orc.range_chain_rows(s1[bins], bins, n_az, *args)[1]
    img = np.empty_like(s1)
    for cols in np.array_split(rng.permutation(n_rg), 2):
        img[:, cols] = orc.azimuth_ifft_cols(s3[:, cols])
    assert orc.rel_l2(img.T, g["img_T"]) < 1e-12
    _, _, _, (t1, t2, t3, _) = orc.sar_focus_csa(raw, *args, return_stages=True)
    assert orc.rel_l2(s1, t1) < 1e-13 and orc.rel_l2(s3, t3) < 1e-13
    bins = np.array([0, 1, n_az // 2 - 1, n_az // 2, n_az - 1])
    assert orc.rel_l2(orc.range_chain_rows(t1[bins], bins, n_az, *args)[0], t2[bins]) < 1e-13


def test_digest_1024():
    g = load_golden("csa_digest_1024.npz")
    raw, k = orc.point_scene(1024, 1024, seed=int(g["seed"]), clutter_db=float(g["clutter_db"]))
    np.testing.assert_array_equal(np.array(orc.focus_args(k)), g["args"])
    img_t, rax, cax = orc.sar_focus_csa_lean(raw, *g["args"])
    pk = np.unravel_index(np.argmax(np.abs(img_t)), img_t.shape)
    assert tuple(pk) == tuple(g["peak_index"])
    assert abs(img_t[pk] - g["peak_value"]) < 1e-9 * abs(g["peak_value"])
    assert orc.rel_l2(img_t[g["rows"], :], g["row_values"]) < 1e-12
    assert orc.rel_l2(img_t[:, g["rows"]], g["col_values"]) < 1e-12
    assert abs(np.linalg.norm(img_t) - g["l2"]) < 1e-10 * g["l2"]


def test_ati_dpca_products():
    g = load_golden("ati_128x128.npz")
    p = orc.ati_dpca(g["slc1"], g["slc2"], mask_frac=0.05, cal_phase=0.0)
    np.testing.assert_allclose(p["ati_phase"], g["ati_phase"], atol=1e-12)
    np.testing.assert_allclose(p["slc1_mag"], g["slc1_mag"], rtol=1e-14)
    np.testing.assert_allclose(p["dpca_mag"], g["dpca_mag"], rtol=1e-12, atol=1e-14)
    np.testing.assert_array_equal(p["mask"], g["mask"])
    np.testing.assert_allclose(p["ati_phase_masked"], g["ati_phase_masked"], atol=1e-12)
    assert abs(orc.phase_balance(g["slc1"], g["slc2"]) - float(g["cal_phase"])) < 1e-12
    # focusing the fixture's raw channels reproduces its SLCs
    s1 = orc.sar_focus_csa(g["raw1"], *g["args"])[0]
    assert orc.rel_l2(s1, g["slc1"]) < 1e-12


def test_echo_models_match_reference():
    g = load_golden("echo_mono.npz")
    tg = [{"position": p, "rcs": r} for p, r in zip(g["tgt_pos"], g["tgt_rcs"])]
    raw = orc.echo_monostatic(tg, g["pos_sat"], g["raw"].shape[1], float(g["fs"]), float(g["t_start_fast"]),
                              float(g["FC"]), float(g["BW"]) / float(g["T_p"]), float(g["T_p"]))
    assert orc.rel_l2(raw, g["raw"]) < 1e-9      # phases ~3e8 rad: fp64 ulp there is 6e-8
    g = load_golden("echo_bistatic.npz")
    tg = [{"position": p, "rcs": r} for p, r in zip(g["tgt_pos"], g["tgt_rcs"])]
    raw = orc.echo_bistatic(tg, g["t_vec"], g["pos_tx"], g["vel_tx"], float(g["rx_offset"]), g["vel_target"],
                            g["raw"].shape[1], float(g["fs"]), float(g["t_start_fast"]), float(g["FC"]),
                            float(g["BW"]) / float(g["T_p"]), float(g["T_p"]))
    assert orc.rel_l2(raw, g["raw"]) < 1e-6


def test_moving_and_vehicle_echo_oracle_matches_reference():
    """run_moving_physics (sar_satellite_moving_sim.py:111-159), run_custom_physics (sar_vehicle_sim.py:83-128)."""
    g = load_golden("echo_moving.npz")
    tg = [{"position": p, "rcs": r} for p, r in zip(g["tgt_pos"], g["tgt_rcs"])]
    o = orc.echo_monostatic(tg, g["pos_sat"], g["raw"].shape[1], float(g["fs"]), float(g["t_start_fast"]), float(g["FC"]),
                            float(g["BW"]) / float(g["T_p"]), float(g["T_p"]), t_vec=g["t_vec"], vel_target=g["vel_target"])
    assert orc.rel_l2(o, g["raw"]) < 1e-12
    g = load_golden("echo_vehicle.npz")
    n, fs = 2048, 360e6
    o = orc.echo_monostatic(tg, g["pos"], n, fs, (2 * float(g["R0"]) / orc.C_LIGHT) - (n / fs) / 2, float(g["fc"]),
                            float(g["bw"]) / float(g["t_p"]), float(g["t_p"]))
    assert orc.rel_l2(o, g["raw"]) < 1e-12


def test_reference_constants():
    k = orc.reference_radar_constants()
    assert abs(k["V_sat"] - 7701.0) < 5 and abs(k["R0"] - 509.4e3) < 200
    assert abs(k["d_rx"] - 2.567) < 2e-3 and abs(k["Kr"] - 2.5e13) < 1.0


@pytest.mark.parametrize("tag", ["rda_200x96", "rda_257x101", "rda_128x64"])
def test_rda_oracle_matches_reference(tag):
    from oracle import rda_oracle as rda
    g = load_golden(tag + ".npz")
    o = rda.sar_focus_rda(g["phist"], *g["args"])
    assert o[0].shape == g["image_mag_T"].shape
    assert orc.rel_l2(o[3], g["phist_compressed"]) < 1e-11
    assert orc.rel_l2(o[4], g["range_doppler"]) < 1e-11
    assert orc.rel_l2(o[5], g["range_doppler_rcmc"]) < 1e-11
    assert orc.rel_l2(o[0], g["image_mag_T"]) < 1e-11
    np.testing.assert_allclose(o[1], g["range_axis_centered"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(o[2], g["cross_range_m"], rtol=1e-13, atol=1e-9)
    np.testing.assert_allclose(o[6], g["doppler_freq"], rtol=1e-14, atol=1e-12)


def test_rda_oracle_matches_the_other_two_copies():
    """sar_focus_rda as pasted into sar_satellite_moving_sim.py:208-285 (three outputs) and sar_vehicle_sim.py:182-273
    (eight outputs, range_doppler_filtered :268 among them): fixtures written by each script's own copy."""
    from oracle import rda_oracle as rda
    g = load_golden("rda_moving_144x80.npz")
    o = rda.sar_focus_rda(g["phist"], *g["args"], variant="moving")
    assert len(o) == 3 and o[0].shape == g["image_mag_T"].shape
    assert orc.rel_l2(o[0], g["image_mag_T"]) < 1e-11
    np.testing.assert_allclose(o[1], g["range_axis_centered"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(o[2], g["cross_range_m"], rtol=1e-13, atol=1e-9)
    g = load_golden("rda_vehicle_96x256.npz")
    o = rda.sar_focus_rda(g["phist"], *g["args"], variant="vehicle")
    assert len(o) == 8
    for i, key in ((0, "image_mag_T"), (3, "phist_compressed"), (4, "range_doppler"), (5, "range_doppler_rcmc"),
                   (6, "range_doppler_filtered")):
        assert orc.rel_l2(o[i], g[key]) < 1e-11, key
    np.testing.assert_allclose(o[7], g["doppler_freq"], rtol=1e-14, atol=1e-12)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_spotlight_echo_oracle_matches_reference(tag):
    """run_physics_spotlight + calculate_raw_snr_db (sar_batch_sim.py:85-169, :54-64); 2e8 rad carrier phase in fp64."""
    from oracle import tdbp_oracle as tb
    s = load_golden(f"spot_{tag}.npz")
    k = tb.constants_from_fixture(s["consts"])
    tg = [{"position": p, "rcs": r} for p, r in zip(s["tgt_pos"], s["tgt_rcs"])]
    raw, t0, n, v = tb.run_physics_spotlight(tg, s["t_vec"], s["pos"], s["vel"], float(s["heading_deg"]),
                                             float(s["speed"]), float(s["l_ant"]), k)
    assert n == int(s["num_samples"]) and t0 == float(s["t_start"])
    np.testing.assert_allclose(v, s["v_tgt"], rtol=1e-15, atol=1e-15)
    assert orc.rel_l2(raw, s["raw"]) < 1e-6
    assert abs(tb.calculate_raw_snr_db(k["R0"], 5000.0, k["Lambda"], k["BW"], float(s["l_ant"]), k) - float(s["snr_db"])) < 1e-9


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("mode", ["mbp", "stdbp"])
def test_tdbp_oracle_matches_reference(tag, mode):
    """tdbp_gpu (sar_batch_sim.py:171-238), moving-target and static focus velocity."""
    from oracle import tdbp_oracle as tb
    g = load_golden(f"tdbp_{tag}.npz")
    k = tb.constants_from_fixture(g["consts"])
    vf = g["v_tgt"] if mode == "mbp" else np.zeros(3)
    img = tb.tdbp(g["raw"], g["pos"], g["vel"], float(g["t_start"]), int(g["num_samples"]), vf, g["t_vec"],
                  float(g["swath"]), int(g["nx"]), int(g["ny"]), k)
    assert img.shape == g["img_" + mode].shape
    assert orc.rel_l2(img, g["img_" + mode]) < 1e-6
