"""Sizes that are not powers of two: chirp-z path against the reference-generated
fixture and the oracle, including the reference's native extents 7199 (azimuth) and 13200 (range)."""
import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import csa_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_reference_fixture_96x80():
    import sarx
    g = load_golden("csa_96x80.npz")
    img_t, rax, cax = sarx.sar_focus_csa(g["phist"], *g["args"])
    assert img_t.shape == g["img_T"].shape == (80, 96)
    assert orc.rel_l2(img_t, g["img_T"]) < TOL
    assert orc.rel_l2(np.abs(img_t), np.abs(g["img_T"])) < TOL
    np.testing.assert_allclose(rax, g["range_axis"], rtol=1e-15)
    np.testing.assert_allclose(cax, g["cross_range_axis"], rtol=1e-12, atol=1e-9)
    c = sarx.sar_focus_csa(g["phist"], *g["args"], materialize_transpose=True)[0]
    assert c.flags.c_contiguous
    np.testing.assert_array_equal(c, img_t)


@pytest.mark.parametrize("n_az,n_rg", [(255, 257), (300, 200), (5, 7), (64, 100), (100, 64), (33, 1024),
                                       (48, 13200),      # native range extent: direct 24 x 22 x 25 lines; azimuth chirp-z in one step
                                       (300, 13200),     # ... with the three-launch azimuth chirp-z (1024 rows) around it
                                       (256, 13200),     # ... with direct two-step azimuth transforms
                                       (64, 13200),      # ... with direct one-step azimuth transforms
                                       (7199, 48),       # native azimuth extent: chirp-z over 16384 rows
                                       (9001, 40),       # chirp-z over 32768 rows (128 x 256 column transform)
                                       (32768, 32),      # the largest azimuth extent, direct
                                       (24, 20000),      # chirp-z over a 65536-point line (256 x 256)
                                       (16, 32768),      # the largest range extent
                                       (1000, 3000)])
def test_any_size_vs_oracle(n_az, n_rg):
    import sarx
    r = np.random.default_rng(n_az * 131 + n_rg)
    raw = (r.standard_normal((n_az, n_rg)) + 1j * r.standard_normal((n_az, n_rg))).astype(np.complex64)
    k = orc.scaled_radar(max(n_az, 16), max(n_rg, 16))
    args = orc.focus_args(k)
    ref = orc.sar_focus_csa(raw, *args)[0]
    img = sarx.sar_focus_csa(raw, *args)[0]
    assert img.shape == (n_rg, n_az)
    assert orc.rel_l2(img, ref) < TOL
    assert orc.rel_l2(np.abs(img), np.abs(ref)) < TOL


def test_point_targets_focus_at_native_like_size():
    """A focusable non-power-of-two scene: peak position and magnitude agree with the oracle."""
    import sarx
    raw, k = orc.point_scene(720, 1320, seed=77, n_targets=4, clutter_db=-25.0)
    args = orc.focus_args(k)
    ref = orc.sar_focus_csa(raw, *args)[0]
    img = sarx.sar_focus_csa(raw, *args)[0]
    assert np.unravel_index(np.argmax(np.abs(img)), img.shape) == np.unravel_index(np.argmax(np.abs(ref)), ref.shape)
    assert orc.rel_l2(np.abs(img), np.abs(ref)) < TOL


def test_mixed_radix_13200_range_passes():
    """The direct line kernels of the reference's native range extent (13200 = 24 x 22 x 25, range_mixed.hip), every mode,
    more lines than workgroups so the persistent loop runs more than once."""
    import sarx
    from sarx import _ffi
    n_az, n_rg = 600, 13200
    ctx = sarx.default_context()
    k = orc.scaled_radar(n_az, n_rg)
    args = orc.focus_args(k)
    plan = sarx.CsaPlan(ctx, n_az, n_rg, *args)
    r = np.random.default_rng(13200)
    x = (r.standard_normal((n_az, n_rg)) + 1j * r.standard_normal((n_az, n_rg))).astype(np.complex64)
    d_a, d_b = ctx.to_device(x), ctx.alloc(x.nbytes)

    def run(pid, src=None):
        if src is not None:
            d_a.upload(src.astype(np.complex64))
        plan.run_pass(pid, d_a, d_b)
        return d_b.download(np.complex64, x.shape)

    f = np.fft.fft(x.astype(np.complex128), axis=1)
    assert orc.rel_l2(run(_ffi.PASS_TEST_RG_FFT), f) < 3e-6
    assert orc.rel_l2(run(_ffi.PASS_TEST_RG_IFFT, f / 100.0), x / 100.0) < 3e-6
    bins = np.arange(n_az)
    o2, o3 = orc.range_chain_rows(x, bins, n_az, *args)
    assert orc.rel_l2(run(_ffi.PASS_RG_FFT_PHI2, x), o2) < 5e-6
    assert orc.rel_l2(run(_ffi.PASS_RG_IFFT_PHI3, o2), o3) < 5e-6
    g3 = run(_ffi.PASS_RG_FUSED_23, x)
    assert orc.rel_l2(g3, o3) < 5e-6
    per_row = np.linalg.norm(g3 - o3, axis=1) / np.linalg.norm(o3, axis=1)
    assert per_row.max() < 1e-5, (int(per_row.argmax()), per_row.max())
    d_a.upload(x)
    plan.run_pass(_ffi.PASS_RG_FUSED_23, d_a, d_a)                 # in place, as the focus runs it
    np.testing.assert_array_equal(d_a.download(np.complex64, x.shape), g3)
    with pytest.raises(sarx.SarxError):
        plan.run_pass(_ffi.PASS_AZ_IFFT, d_a, d_b)                 # azimuth passes of any-size plans have no per-pass entry
    plan.close()


def test_direct_lines_equal_chirp_z_lines(monkeypatch):
    """SARX_RANGE_MIXED=0 keeps the chirp-z range route: both routes give the same image."""
    import sarx
    ctx = sarx.default_context()
    n_az, n_rg = 300, 13200
    k = orc.scaled_radar(n_az, n_rg)
    args = orc.focus_args(k)
    r = np.random.default_rng(5)
    x = (r.standard_normal((n_az, n_rg)) + 1j * r.standard_normal((n_az, n_rg))).astype(np.complex64)
    d_in, d_a, d_b = ctx.to_device(x), ctx.alloc(x.nbytes), ctx.alloc(x.nbytes)
    direct = sarx.CsaPlan(ctx, n_az, n_rg, *args)
    monkeypatch.setenv("SARX_RANGE_MIXED", "0")
    chirpz = sarx.CsaPlan(ctx, n_az, n_rg, *args)
    monkeypatch.delenv("SARX_RANGE_MIXED")
    direct.focus_dev(d_in, d_a)
    chirpz.focus_dev(d_in, d_b)
    a, b = d_a.download(np.complex64, x.shape), d_b.download(np.complex64, x.shape)
    assert orc.rel_l2(a, b) < 1e-5
    assert chirpz.scratch_bytes() > direct.scratch_bytes()         # no per-row convolution spectra on the direct route
    direct.close()
    chirpz.close()


def test_native_scene_size_vs_oracle():
    """The reference's own scene, 7199 pulses x 13200 samples (sar_ati_dcpa_sim_csa.py:47,111,402), with its literal radar
    constants: the whole image against the oracle (row-blocked complex128)."""
    import sarx
    from sarx import radar
    n_az, n_rg = 7199, 13200
    ctx = sarx.default_context()
    args = radar.focus_args()                                       # the reference's constants and 22 us window (:18-38,112)
    plan = sarx.CsaPlan(ctx, n_az, n_rg, *args)
    px = n_az * n_rg
    d_in, d_out = ctx.alloc(px * 8), ctx.alloc(px * 8)
    ctx.fill_noise(d_in, px, 7199)
    from sarx import _ffi
    d_max = ctx.alloc(_ffi.MAX_SLOT_BYTES)                           # sarx_csa_plan_set_max_slot on the prime-factor route
    plan.set_max_slot(d_max)
    plan.focus_dev(d_in, d_out)
    plan.set_max_slot(None)
    raw = d_in.download(np.complex64, (n_az, n_rg))
    img = d_out.download(np.complex64, (n_az, n_rg))
    shards = d_max.download(np.float32, (256, 32))
    assert not shards[:, 1:].any()
    assert abs(float(shards[:, 0].max()) - float(np.abs(img).max())) <= 2e-7 * float(np.abs(img).max())
    ref = orc.sar_focus_csa_lean(raw, *args, workers=8, block=64)[0]
    assert orc.rel_l2(img.T, ref) < TOL
    assert orc.rel_l2(np.abs(img.T), np.abs(ref)) < TOL
    # the product stage out of a second focus's last launch (sarx_csa_plan_set_ati) on the prime-factor route: bit for bit the
    # separate launch's planes
    d_in2, s2 = ctx.alloc(px * 8), ctx.alloc(px * 8)
    ctx.fill_noise(d_in2, px, 7200)
    plan.focus_dev(d_in2, s2)
    ref_p = {kk: ctx.alloc(px * 4) for kk in ("ati_phase", "slc1_mag", "dpca_mag")}
    ctx.ati_dpca_masked(d_out, s2, px, 0.2, d_max, 0.05, ref_p)
    mx, sm = ctx.ati_stats()
    got_p = {kk: ctx.alloc(px * 4) for kk in ("ati_phase", "slc1_mag", "dpca_mag")}
    s2f = ctx.alloc(px * 8)
    plan.set_ati(d_out, d_max, 0.05, 0.2, got_p["ati_phase"], got_p["slc1_mag"], got_p["dpca_mag"])
    plan.focus_dev(d_in2, s2f)
    plan.set_ati(None)
    mx2, sm2 = ctx.ati_stats()
    for kk in ref_p:
        np.testing.assert_array_equal(got_p[kk].download(np.float32, (px,)), ref_p[kk].download(np.float32, (px,)))
    assert mx2 == mx and abs(sm2 - sm) <= 1e-12 * abs(sm)
    plan.focus_dev(d_in2, s2f)                                       # switched off: an ordinary focus again
    np.testing.assert_array_equal(s2f.download(np.complex64, (px,)), s2.download(np.complex64, (px,)))
    for b in (d_in2, s2, s2f, *ref_p.values(), *got_p.values()):
        b.release()
    plan2 = sarx.CsaPlan(ctx, 100, 13200, *args)                     # chirp-z azimuth route: no such epilogue, and it says so
    with pytest.raises(sarx.SarxError):
        plan2.set_max_slot(d_max)
    with pytest.raises(sarx.SarxError):
        plan2.set_ati(d_out, d_max, 0.05, 0.0, d_max, d_max, d_max)
    plan2.close()
    for b in (d_in, d_out, d_max):
        b.release()
    plan.close()


def test_prime_factor_azimuth_passes_at_native_size():
    """7199 = 23 x 313 pulses: Good-Thomas map + Rader's algorithm for the 313 (az_pfa.hip), forward with Phi_1 and
    inverse, columns sampled against numpy.fft through the oracle's per-column restatement (:233-274, :385)."""
    import sarx
    from sarx import _ffi, radar
    from sarx.engine import download_block
    n_az, n_rg = 7199, 13200
    ctx = sarx.default_context()
    args = radar.focus_args()
    plan = sarx.CsaPlan(ctx, n_az, n_rg, *args)
    px = n_az * n_rg
    x, y, z = ctx.alloc(px * 8), ctx.alloc(px * 8), ctx.alloc(px * 8)
    ctx.fill_noise(x, px, 23313)
    cols = np.array([0, 1, 31, 32, 33, 63, 64, 6599, 6600, 13167, 13168, 13199])      # incl. both sides of tile edges and the ragged last tile
    get = lambda buf: np.concatenate([download_block(ctx, buf.ptr, n_rg, 0, n_az, c, 1) for c in cols], axis=1)
    plan.run_pass(_ffi.PASS_AZ_FFT_PHI1, x, y)
    xin = get(x)
    o1 = orc.azimuth_fft_cols(xin, cols, n_rg, *args)
    g1 = get(y)
    assert orc.rel_l2(g1, o1) < 5e-6
    per_row = np.abs(g1 - o1).max(axis=1) / np.abs(o1).max()
    assert per_row.max() < 1e-4, int(per_row.argmax())          # no single azimuth bin is misplaced
    np.testing.assert_array_equal(get(x), xin)                  # the source is not modified
    plan.run_pass(_ffi.PASS_AZ_IFFT, y, z)
    assert orc.rel_l2(get(z), orc.azimuth_ifft_cols(g1)) < 5e-6
    for b in (x, y, z):
        b.release()
    plan.close()


def test_prime_factor_route_equals_chirp_z_route(monkeypatch):
    import sarx
    from sarx import radar
    n_az, n_rg = 7199, 13200
    ctx = sarx.default_context()
    args = radar.focus_args()
    px = n_az * n_rg
    d_in, d_a, d_b = ctx.alloc(px * 8), ctx.alloc(px * 8), ctx.alloc(px * 8)
    ctx.fill_noise(d_in, px, 99)
    pfa = sarx.CsaPlan(ctx, n_az, n_rg, *args)
    monkeypatch.setenv("SARX_AZ_PFA", "0")
    czt = sarx.CsaPlan(ctx, n_az, n_rg, *args)
    monkeypatch.delenv("SARX_AZ_PFA")
    pfa.focus_dev(d_in, d_a)
    czt.focus_dev(d_in, d_b)
    a, b = d_a.download(np.complex64, (n_az, n_rg)), d_b.download(np.complex64, (n_az, n_rg))
    assert orc.rel_l2(a, b) < 1e-5
    for buf in (d_in, d_a, d_b):
        buf.release()
    pfa.close()
    czt.close()


def test_one_workgroup_forms_kept_for_ab_still_match(tmp_path):
    """SARX_RADER_TWO=0 / SARX_MIXED_PLANES=0 select the one-workgroup-per-CU forms of the Rader and the 13200-sample fused range
    launches (the A/B partners of DESIGN.md 4.6).  The switches are read once per process, so a child process focuses the same
    seeded native-size echo with them and the two images are compared (different twiddle arithmetic: not bit-identical)."""
    import os
    import subprocess
    import sys
    import sarx
    from sarx import radar
    n_az, n_rg = 7199, 13200
    out = tmp_path / "old_forms.npy"
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'nis-sar-amtigmti-video_amd')!r})\n"
        "import sarx\n"
        "from sarx import radar, _ffi\n"
        f"n_az, n_rg = {n_az}, {n_rg}\n"
        "ctx = sarx.Context(0)\n"
        "plan = sarx.CsaPlan(ctx, n_az, n_rg, *radar.focus_args(), flags=_ffi.FUSE_RANGE)\n"
        "d_in, d_out = ctx.alloc(n_az * n_rg * 8), ctx.alloc(n_az * n_rg * 8)\n"
        "ctx.fill_noise(d_in, n_az * n_rg, 4242)\n"
        "plan.focus_dev(d_in, d_out)\n"
        f"np.save({str(out)!r}, d_out.download(np.complex64, (n_az, n_rg))[::37])\n"
    )
    env = dict(os.environ, SARX_RADER_TWO="0", SARX_MIXED_PLANES="0")
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=300)
    old = np.load(out)
    from sarx import _ffi
    ctx = sarx.default_context()
    plan = sarx.CsaPlan(ctx, n_az, n_rg, *radar.focus_args(), flags=_ffi.FUSE_RANGE)
    d_in, d_out = ctx.alloc(n_az * n_rg * 8), ctx.alloc(n_az * n_rg * 8)
    ctx.fill_noise(d_in, n_az * n_rg, 4242)
    plan.focus_dev(d_in, d_out)
    new = d_out.download(np.complex64, (n_az, n_rg))[::37]
    assert np.isfinite(new).all() and np.abs(new).max() > 0
    assert orc.rel_l2(new, old) < 2e-6
    for b in (d_in, d_out):
        b.release()
    plan.close()
