"""GPU parity: HIP path (through the C ABI) vs the CPU oracle and the golden
fixtures.  Tolerance from BASELINE.json's north_star: <= 1e-4 relative L2 on
|img| and on the masked ATI phase; we also hold the complex image to 1e-4."""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import load_golden
from oracle import csa_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def sx():
    import sarx
    return sarx


@pytest.fixture(scope="module")
def ctx(sx):
    return sx.default_context()


def _plan(sx, ctx, n_az, n_rg, args, flags=0):
    return sx.CsaPlan(ctx, n_az, n_rg, *args, flags=flags)


def _rand(shape, seed):
    r = np.random.default_rng(seed)
    return (r.standard_normal(shape) + 1j * r.standard_normal(shape)).astype(np.complex64)


# ---- plain FFT kernels -------------------------------------------------------------
@pytest.mark.parametrize("n", [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_range_fft_all_sizes(sx, ctx, n):
    from sarx import _ffi
    k = orc.scaled_radar(16, n)
    plan = _plan(sx, ctx, 16, n, orc.focus_args(k))
    x = _rand((16, n), n)
    d_in, d_out = ctx.to_device(x), ctx.alloc(x.nbytes)
    plan.run_pass(_ffi.PASS_TEST_RG_FFT, d_in, d_out)
    y = d_out.download(np.complex64, x.shape)
    assert orc.rel_l2(y, np.fft.fft(x.astype(np.complex128), axis=1)) < 2e-6
    plan.run_pass(_ffi.PASS_TEST_RG_IFFT, d_out, d_in)          # round trip
    assert orc.rel_l2(d_in.download(np.complex64, x.shape), x) < 3e-6
    plan.close()


# ---- per-pass parity against the oracle's intermediates ---------------------------------
@pytest.mark.parametrize("n_az,n_rg", [(64, 64), (128, 128), (16, 256), (256, 16), (256, 256), (512, 1024),
                                       (2048, 512), (1024, 2048)])
def test_each_pass(sx, ctx, n_az, n_rg):
    from sarx import _ffi
    raw, k = orc.point_scene(n_az, n_rg, seed=n_az + n_rg, clutter_db=-15.0)
    args = orc.focus_args(k)
    _, _, _, (s1, s2, s3, s4) = orc.sar_focus_csa(raw, *args, return_stages=True)
    plan = _plan(sx, ctx, n_az, n_rg, args)
    nb = raw.nbytes
    d_a, d_b = ctx.alloc(nb), ctx.alloc(nb)

    def run(pid, src):
        d_a.upload(src.astype(np.complex64))
        plan.run_pass(pid, d_a, d_b)
        return d_b.download(np.complex64, (n_az, n_rg))

    assert orc.rel_l2(run(_ffi.PASS_AZ_FFT_PHI1, raw), s1) < 5e-6
    assert orc.rel_l2(run(_ffi.PASS_RG_FFT_PHI2, s1), s2) < 5e-6
    assert orc.rel_l2(run(_ffi.PASS_RG_IFFT_PHI3, s2), s3) < 5e-6
    assert orc.rel_l2(run(_ffi.PASS_AZ_IFFT, s3), s4) < 5e-6
    assert orc.rel_l2(run(_ffi.PASS_RG_FUSED_23, s1), s3) < 5e-6
    # range passes may run in place
    d_a.upload(s1.astype(np.complex64))
    plan.run_pass(_ffi.PASS_RG_FUSED_23, d_a, d_a)
    assert orc.rel_l2(d_a.download(np.complex64, (n_az, n_rg)), s3) < 5e-6
    plan.close()


# ---- end to end against the reference-generated fixtures ---------------------------------
@pytest.mark.parametrize("tag", ["csa_64x64", "csa_128x128", "csa_256x256", "csa_128x512", "csa_512x128",
                                 "csa_refconst_128x256"])
@pytest.mark.parametrize("fuse", [True, False])
def test_focus_matches_reference_fixture(sx, tag, fuse):
    g = load_golden(tag + ".npz")
    img_t, rax, cax = sx.sar_focus_csa(g["phist"], *g["args"], fuse_range=fuse)
    assert img_t.shape == g["img_T"].shape and img_t.dtype == np.complex64
    assert orc.rel_l2(np.abs(img_t), np.abs(g["img_T"])) < TOL
    assert orc.rel_l2(img_t, g["img_T"]) < TOL
    np.testing.assert_allclose(rax, g["range_axis"], rtol=1e-15)
    np.testing.assert_allclose(cax, g["cross_range_axis"], rtol=1e-12, atol=1e-9)


def test_focus_digest_1024(sx):
    g = load_golden("csa_digest_1024.npz")
    raw, k = orc.point_scene(1024, 1024, seed=int(g["seed"]), clutter_db=float(g["clutter_db"]))
    img_t, _, _ = sx.sar_focus_csa(raw, *g["args"])
    pk = np.unravel_index(np.argmax(np.abs(img_t)), img_t.shape)
    assert tuple(pk) == tuple(g["peak_index"])
    assert orc.rel_l2(img_t[g["rows"], :], g["row_values"]) < TOL
    assert orc.rel_l2(img_t[:, g["rows"]], g["col_values"]) < TOL
    assert abs(np.linalg.norm(img_t.astype(np.complex128)) - g["l2"]) < TOL * g["l2"]


@pytest.mark.parametrize("n_az,n_rg", [(2048, 2048), (4096, 1024), (512, 8192), (4096, 4096)])
def test_focus_vs_oracle_large(sx, n_az, n_rg):
    raw, k = orc.point_scene(n_az, n_rg, seed=n_az ^ n_rg, clutter_db=-20.0, n_targets=7)
    args = orc.focus_args(k)
    ref = orc.sar_focus_csa_lean(raw, *args, workers=8)[0]
    img_t = sx.sar_focus_csa(raw, *args)[0]
    assert orc.rel_l2(np.abs(img_t), np.abs(ref)) < TOL
    assert orc.rel_l2(img_t, ref) < TOL
    assert np.unravel_index(np.argmax(np.abs(img_t)), img_t.shape) == np.unravel_index(np.argmax(np.abs(ref)), ref.shape)


def test_layouts_and_fusion_agree(sx):
    raw, k = orc.point_scene(512, 256, seed=9)
    args = orc.focus_args(k)
    a = sx.sar_focus_csa(raw, *args, fuse_range=True)[0]
    b = sx.sar_focus_csa(raw, *args, fuse_range=False)[0]
    c = sx.sar_focus_csa(raw, *args, materialize_transpose=True)[0]
    assert a.shape == (256, 512) and not a.flags.c_contiguous      # a view, like the reference's img.T
    assert c.shape == (256, 512) and c.flags.c_contiguous
    np.testing.assert_array_equal(a, c)                          # corner turn moves bits only
    assert orc.rel_l2(a, b) < 2e-6
    # complex128 input is accepted (reference dtype) and rounded once to complex64
    d = sx.sar_focus_csa(raw.astype(np.complex128), *args)[0]
    np.testing.assert_array_equal(a, d)
    # ... by the library's copy threads when contiguous (sarx_csa_focus_host_c128), by NumPy otherwise: the same rounding
    wide = raw.astype(np.complex128) * (1 + 1e-9) + 1e-12j
    e = sx.sar_focus_csa(wide, *args)[0]
    np.testing.assert_array_equal(e, sx.sar_focus_csa(wide.astype(np.complex64), *args)[0])
    strided = np.asfortranarray(wide)
    np.testing.assert_array_equal(e, sx.sar_focus_csa(strided, *args)[0])


def test_linearity_and_determinism(sx):
    raw, k = orc.point_scene(256, 512, seed=3)
    args = orc.focus_args(k)
    x2 = _rand(raw.shape, 77)
    fa = sx.sar_focus_csa(raw, *args)[0]
    fb = sx.sar_focus_csa(x2, *args)[0]
    fab = sx.sar_focus_csa(raw + 2 * x2, *args)[0]
    assert orc.rel_l2(fab, fa + 2 * fb) < 5e-6
    np.testing.assert_array_equal(fa, sx.sar_focus_csa(raw, *args)[0])


# ---- ATI / DPCA --------------------------------------------------------------------------
def _masked_phase_err(p, ref_phase, mask):
    d = np.angle(np.exp(1j * (p[mask].astype(np.float64) - ref_phase[mask])))
    return float(np.linalg.norm(d) / max(np.linalg.norm(ref_phase[mask]), 1e-30))


def test_ati_dpca_fixture(sx):
    g = load_golden("ati_128x128.npz")
    s1, s2 = g["slc1"].astype(np.complex64), g["slc2"].astype(np.complex64)
    r = sx.ati_dpca(s1, s2, mask_frac=0.05, complex_products=True, viewer_products=True)
    assert orc.rel_l2(r["slc1_mag"], g["slc1_mag"]) < 1e-6
    assert orc.rel_l2(r["dpca_mag"], g["dpca_mag"]) < 1e-5
    assert _masked_phase_err(r["ati_phase"], g["ati_phase"], g["mask"]) < TOL
    assert (r["mask"] != g["mask"]).sum() <= 2                       # pixels within fp32 rounding of the threshold
    both = r["mask"] & g["mask"]
    assert _masked_phase_err(r["ati_phase_masked"], g["ati_phase_masked"], both) < TOL
    assert np.all(r["ati_phase_masked"][~r["mask"]] == 0)
    assert abs(np.angle(r["sum_interf"]) - float(g["cal_phase"])) < 1e-6
    assert abs(sx.phase_balance(s1, s2) - float(g["cal_phase"])) < 1e-6
    ref = orc.ati_dpca(s1, s2)
    assert orc.rel_l2(r["ati_interf"], ref["ati_interf"]) < 1e-6
    assert orc.rel_l2(r["dpca_diff"], ref["dpca_diff"]) < 1e-6
    assert orc.rel_l2(r["Ch2 Magnitude"], ref["Ch2 Magnitude"]) < 1e-6
    for key in ("Ch1 Phase", "Ch2 Phase", "DPCA Phase"):
        assert _masked_phase_err(r[key], ref[key], g["mask"]) < TOL


def test_ati_cal_phase_and_views(sx):
    g = load_golden("ati_128x128.npz")
    s1, s2 = g["slc1"].astype(np.complex64), g["slc2"].astype(np.complex64)
    cal = float(g["cal_phase"])
    r = sx.ati_dpca(s1.T, s2.T, cal_phase=cal)                     # F-ordered views, as sar_focus_csa returns
    ref = orc.ati_dpca(s1.T, s2.T, cal_phase=cal)
    assert r["ati_phase"].shape == s1.T.shape
    assert _masked_phase_err(r["ati_phase"], ref["ati_phase"], ref["mask"]) < TOL
    assert orc.rel_l2(r["dpca_mag"], ref["dpca_mag"]) < 1e-5
    # after balancing, the mean interferogram phase is ~0
    assert abs(np.angle(np.sum(s1 * np.conj(s2 * np.exp(1j * cal))))) < 1e-6


def test_two_channel_end_to_end(sx):
    """focus_ati_dpca = the fused product stage (max slot + ATI epilogue of channel 2's last azimuth launch), against the
    oracle; the unmasked-phase form (separate ATI launch) gives the same planes bit for bit."""
    (r1, r2), k = orc.point_scene(512, 512, seed=31, clutter_db=-25.0, two_channel=True)
    args = orc.focus_args(k)
    res = sx.focus_ati_dpca(r1, r2, *args, pulse_shift=False)
    assert res["fused_products"] and "ati_phase" not in res
    o1 = orc.sar_focus_csa(r1, *args)[0]
    o2 = orc.sar_focus_csa(r2, *args)[0]
    ref = orc.ati_dpca(o1, o2)
    inside = ref["slc1_mag"] > 0.05 * ref["max_mag"] * (1 + 1e-4)          # away from the threshold: no borderline pixel
    outside = ref["slc1_mag"] < 0.05 * ref["max_mag"] * (1 - 1e-4)
    assert orc.rel_l2(np.abs(res["slc1"]), np.abs(o1)) < TOL
    assert orc.rel_l2(np.abs(res["slc2"]), np.abs(o2)) < TOL
    assert orc.rel_l2(res["slc1_mag"], ref["slc1_mag"]) < TOL
    assert _masked_phase_err(res["ati_phase_masked"], ref["ati_phase"], inside) < TOL
    assert (res["ati_phase_masked"][outside] == 0).all()
    assert abs(res["max_mag"] - ref["max_mag"]) < 1e-5 * ref["max_mag"]
    assert abs(res["sum_interf"] - np.sum(o1 * np.conj(o2))) < 1e-4 * abs(np.sum(o1 * np.conj(o2)))
    assert orc.rel_l2(res["dpca_mag"], ref["dpca_mag"]) < 5e-4      # difference of nearly equal images
    # physics: DPCA suppresses the stationary scene relative to channel 1
    assert np.median(res["dpca_mag"][ref["mask"]] / res["slc1_mag"][ref["mask"]]) < 0.5
    two = sx.focus_ati_dpca(r1, r2, *args, pulse_shift=False, unmasked_phase=True)
    assert not two["fused_products"]
    assert _masked_phase_err(two["ati_phase"], ref["ati_phase"], ref["mask"]) < TOL
    for key in ("slc1", "slc2", "slc1_mag", "dpca_mag", "ati_phase_masked"):
        np.testing.assert_array_equal(two[key], res[key])
    lean = sx.focus_ati_dpca(r1, r2, *args, pulse_shift=False, return_slc2=False)
    assert "slc2" not in lean
    for key in ("slc1", "slc1_mag", "dpca_mag", "ati_phase_masked"):
        np.testing.assert_array_equal(lean[key], res[key])


def test_host_call_results_come_from_a_page_locked_pool(sx):
    """sar_focus_csa(numpy) -> numpy, the call a maintainer makes (:410-411): results of 64 MiB and more sit on page-locked
    blocks of a per-context pool.  A result stays the caller's while it is alive (a second call gets another block), its block is
    reused once it is gone, `out=` overwrites an earlier result in place, and every variant returns the same image bit for bit."""
    raw, k = orc.point_scene(2048, 4096, seed=3, n_targets=3)            # 64 MiB image
    args = orc.focus_args(k)
    ctx = sx.Context(0)                                                  # its own context: the pool's request counts start at zero
    from sarx.engine import _PinnedBlock

    def root(x):
        while getattr(x, "base", None) is not None:
            x = x.base
        return x
    first, ra, ca = sx.sar_focus_csa(raw, *args, ctx=ctx)
    second, _, _ = sx.sar_focus_csa(raw, *args, ctx=ctx)
    third, _, _ = sx.sar_focus_csa(raw, *args, ctx=ctx)
    assert first.shape == (4096, 2048) and first.T.flags.c_contiguous
    # the first three requests of a size are ordinary arrays (a one-shot script never pays hipHostMalloc) ...
    assert not any(isinstance(root(x), _PinnedBlock) for x in (first, second, third))
    np.testing.assert_array_equal(first, second)
    np.testing.assert_array_equal(first, third)
    ref = first.copy()
    del first, second, third
    a, _, _ = sx.sar_focus_csa(raw, *args, ctx=ctx)                      # ... from the fourth on the size has page-locked blocks
    assert isinstance(root(a), _PinnedBlock)
    addr_a = a.T.ctypes.data
    b, _, _ = sx.sar_focus_csa(raw, *args, ctx=ctx)                      # a is still alive: b must not alias it
    assert b.T.ctypes.data != addr_a
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, ref)
    del a
    import gc
    gc.collect()
    c, _, _ = sx.sar_focus_csa(raw, *args, ctx=ctx)                      # a's block is back in the pool and is handed out again
    assert c.T.ctypes.data == addr_a
    np.testing.assert_array_equal(c, ref)
    b[...] = 0
    d, _, _ = sx.sar_focus_csa(raw, *args, ctx=ctx, out=b)               # an earlier result as the destination
    assert d.T.ctypes.data == b.T.ctypes.data
    np.testing.assert_array_equal(b, ref)
    plain = np.zeros((4096, 2048), np.complex64)                         # any array with the img.T layout works as out
    with pytest.raises(ValueError):
        sx.sar_focus_csa(raw, *args, ctx=ctx, out=plain)                 # C-contiguous [n_rg x n_az] is the wrong layout for the view
    e, _, _ = sx.sar_focus_csa(raw, *args, ctx=ctx, out=np.zeros((2048, 4096), np.complex64).T)
    np.testing.assert_array_equal(e, ref)
    assert isinstance(root(c), _PinnedBlock)
    small, ks = orc.point_scene(256, 256, seed=3, n_targets=3)
    s_img, _, _ = sx.sar_focus_csa(small, *orc.focus_args(ks), ctx=ctx)   # small results stay ordinary arrays
    assert not isinstance(root(s_img), _PinnedBlock)
    del b, c, d, e, s_img
    sx.clear_plan_cache()                                                # the facade's cached plans of this context
    ctx.close()


def test_two_channel_fallback_takes_the_threshold_from_channel_one(sx):
    """Sizes without the fused ATI epilogue (here n_rg = 16: not a multiple of 32) accept the max slot but not set_ati; the
    facade then runs the separate masked ATI launch, whose 5 % threshold must still be 0.05 * max|slc1| (:447) - channel 2's
    focus must not re-reduce the slot.  Channel 2 is three times stronger here, so a threshold taken from it would mask
    every pixel between 5 % and 15 % of max|slc1|."""
    (r1, r2), k = orc.point_scene(64, 16, seed=5, clutter_db=-6.0, two_channel=True)
    r2 = (3.0 * r2).astype(np.complex64)
    args = orc.focus_args(k)
    res = sx.focus_ati_dpca(r1, r2, *args, pulse_shift=False)
    assert not res["fused_products"]
    o1 = orc.sar_focus_csa(r1, *args)[0]
    o2 = orc.sar_focus_csa(r2, *args)[0]
    ref = orc.ati_dpca(o1, o2)
    inside = ref["slc1_mag"] > 0.05 * ref["max_mag"] * (1 + 1e-4)
    outside = ref["slc1_mag"] < 0.05 * ref["max_mag"] * (1 - 1e-4)
    between = inside & (ref["slc1_mag"] < 0.15 * ref["max_mag"])
    assert between.sum() > 10 and outside.sum() > 0
    assert abs(res["max_mag"] - ref["max_mag"]) < 1e-5 * ref["max_mag"]
    assert _masked_phase_err(res["ati_phase_masked"], ref["ati_phase"], inside) < TOL
    assert (res["ati_phase_masked"][between] != 0).all()
    assert (res["ati_phase_masked"][outside] == 0).all()
    assert orc.rel_l2(res["slc1_mag"], ref["slc1_mag"]) < TOL


def test_two_channel_facade_on_a_reused_workspace(sx, ctx):
    """focus_ati_dpca with device arrays in, device_output and a two_channel_workspace: nothing is allocated or fetched per
    call (fetch_stats=False: the call only enqueues), the planes equal the default call's bit for bit, and a second frame
    through the same workspace overwrites them."""
    from sarx.engine import DeviceArray
    (r1, r2), k = orc.point_scene(256, 1024, seed=11, clutter_db=-20.0, two_channel=True)
    args = orc.focus_args(k)
    ref = sx.focus_ati_dpca(r1, r2, *args, pulse_shift=False, ctx=ctx)
    ws = sx.two_channel_workspace(ctx, 256, 1024)
    d1, d2 = DeviceArray(ctx.to_device(r1), r1.shape), DeviceArray(ctx.to_device(r2), r2.shape)
    out = sx.focus_ati_dpca(d1, d2, *args, pulse_shift=False, ctx=ctx, device_output=True, workspace=ws, fetch_stats=False)
    assert out["max_mag"] is None and out["fused_products"] and out["slc1_mag"] is ws["slc1_mag"]
    mx, sm = ctx.ati_stats()
    assert mx == ref["max_mag"] and sm == ref["sum_interf"]
    for key in ("slc1_mag", "dpca_mag", "ati_phase_masked"):
        np.testing.assert_array_equal(out[key].download(np.float32, r1.shape).T, ref[key])
    np.testing.assert_array_equal(out["slc1"].download(np.complex64, r1.shape).T, ref["slc1"])
    out2 = sx.focus_ati_dpca(d2, d1, *args, pulse_shift=False, ctx=ctx, device_output=True, workspace=ws)     # channels swapped
    swapped = sx.focus_ati_dpca(r2, r1, *args, pulse_shift=False, ctx=ctx)
    np.testing.assert_array_equal(out2["dpca_mag"].download(np.float32, r1.shape).T, swapped["dpca_mag"])
    assert out2["max_mag"] == swapped["max_mag"]
    with pytest.raises(ValueError):
        sx.focus_ati_dpca(r1[:128], r2[:128], *args, pulse_shift=False, ctx=ctx, workspace=ws)
    for b in (d1, d2, *(v for kk, v in ws.items() if kk != "shape")):
        b.release()


def test_max_slot_and_ati_left_armed_together(sx, ctx):
    """A C caller may leave sarx_csa_plan_set_max_slot armed while the second channel's focus runs with
    sarx_csa_plan_set_ati: that focus reads the slot as its threshold and must neither clear nor re-reduce it
    (it used to clear it: threshold 0, every pixel passed the mask, sarx_ati_stats reported max 0)."""
    from sarx import _ffi
    (r1, r2), k = orc.point_scene(256, 512, seed=7, clutter_db=-20.0, two_channel=True)
    args = orc.focus_args(k)
    plan = _plan(sx, ctx, 256, 512, args, flags=_ffi.FUSE_RANGE)
    n = r1.size
    d1, d2 = ctx.to_device(r1), ctx.to_device(r2)
    s1, s2, d_max = ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc(_ffi.MAX_SLOT_BYTES)
    planes = [ctx.alloc(n * 4) for _ in range(6)]
    plan.set_max_slot(d_max)
    plan.focus_dev(d1, s1)
    plan.set_ati(s1, d_max, 0.05, 0.0, *planes[:3])
    plan.focus_dev(d2, s2)                                   # max slot still armed
    mx_armed, sum_armed = ctx.ati_stats()
    got = [b.download(np.float32, r1.shape) for b in planes[:3]]
    plan.set_ati(None)
    plan.focus_dev(d1, s1)                                   # the careful order: slot unset before the second focus
    plan.set_max_slot(None)
    plan.set_ati(s1, d_max, 0.05, 0.0, *planes[3:])
    plan.focus_dev(d2, s2)
    plan.set_ati(None)
    mx, sm = ctx.ati_stats()
    assert mx_armed == mx > 0 and sum_armed == sm
    for a, b in zip(got, planes[3:]):
        np.testing.assert_array_equal(a, b.download(np.float32, r1.shape))
    assert (got[0] == 0).any() and (got[0] != 0).any()       # the mask really masks
    for b in (d1, d2, s1, s2, d_max, *planes):
        b.release()


# ---- streaming helpers ---------------------------------------------------------------------
def test_corner_turn_multilook_noise(sx, ctx):
    x = _rand((192, 320), 5)
    d_in, d_out = ctx.to_device(x), ctx.alloc(x.nbytes)
    ctx.corner_turn(d_in, d_out, 192, 320)
    np.testing.assert_array_equal(d_out.download(np.complex64, (320, 192)), x.T)
    y = _rand((256, 1024), 6)
    d_y, d_m = ctx.to_device(y), ctx.alloc(64 * 256 * 4)
    ctx.multilook(d_y, d_m, 256, 1024, 4)
    ref = (np.abs(y.astype(np.complex128)) ** 2).reshape(64, 4, 256, 4).mean(axis=(1, 3))
    assert orc.rel_l2(d_m.download(np.float32, (64, 256)), ref) < 1e-6
    n = 1 << 20
    d_n = ctx.alloc(n * 8)
    ctx.fill_noise(d_n, n, 42)
    a = d_n.download(np.complex64, (n,))
    ctx.fill_noise(d_n, n, 42)
    np.testing.assert_array_equal(a, d_n.download(np.complex64, (n,)))
    assert abs(a.real.std() - 1) < 0.01 and abs(a.imag.std() - 1) < 0.01 and abs(a.mean()) < 0.01


@pytest.mark.parametrize("rows,cols", [(1, 1), (65, 3), (100, 77), (7199, 40), (64, 8192), (9000, 8200)])
def test_corner_turn_ragged_and_streaming(sx, ctx, rows, cols, monkeypatch):
    """sarx_corner_turn_dev moves bits only: ragged edge tiles (nothing read or written outside the image: the buffers are
    exactly image-sized and the neighbours are checked) and the nontemporal form large images take (9000 x 8200 = 563 MiB)."""
    x = _rand((rows, cols), rows + cols)
    guard = np.complex64(7 - 3j)
    d_in = ctx.to_device(x)
    d_out = ctx.to_device(np.full(rows * cols + 128, guard, np.complex64))
    ctx.corner_turn(d_in, d_out, rows, cols)
    out = d_out.download(np.complex64, (rows * cols + 128,))
    np.testing.assert_array_equal(out[: rows * cols].reshape(cols, rows), x.T)
    assert (out[rows * cols:] == guard).all()
    d_in.release(); d_out.release()


@pytest.mark.parametrize("n_az,n_rg,looks", [(512, 1024, 16), (128, 256, 4), (2048, 512, 32), (64, 64, 1)])
def test_fused_look_slot(sx, ctx, n_az, n_rg, looks):
    """sarx_csa_plan_set_look_slot: the multilooked intensity emitted by the focus itself equals the mean of |image|^2 over
    looks x looks blocks of the image it wrote, is reproduced bit for bit, leaves the image untouched, and switches off again."""
    from sarx import _ffi
    raw, k = orc.point_scene(n_az, n_rg, seed=looks + n_az, clutter_db=-10.0)
    args = orc.focus_args(k)
    plan = _plan(sx, ctx, n_az, n_rg, args, flags=_ffi.FUSE_RANGE)
    d_in, d_img, d_ref = ctx.to_device(raw), ctx.alloc(raw.nbytes), ctx.alloc(raw.nbytes)
    nslot = (n_az // looks) * (n_rg // looks)
    d_slot = ctx.alloc(nslot * 4 + 64)
    plan.focus_dev(d_in, d_ref)
    ctx.lib.sarx_memset(ctx.h, d_slot.ptr, 0xFF, nslot * 4 + 64)
    plan.set_look_slot(looks, d_slot.ptr)
    plan.focus_dev(d_in, d_img)
    img = d_img.download(np.complex64, raw.shape)
    np.testing.assert_array_equal(img, d_ref.download(np.complex64, raw.shape))
    slot = d_slot.download(np.float32, (n_az // looks, n_rg // looks))
    ref = (np.abs(img.astype(np.complex128)) ** 2).reshape(n_az // looks, looks, n_rg // looks, looks).mean(axis=(1, 3))
    assert orc.rel_l2(slot, ref) < 1e-6
    guard = d_slot.download(np.uint8, (nslot * 4 + 64,))[nslot * 4:]
    assert (guard == 0xFF).all()                                  # nothing written past the slot
    plan.focus_dev(d_in, d_img)
    np.testing.assert_array_equal(d_slot.download(np.float32, slot.shape), slot)
    plan.set_look_slot(looks, None)
    ctx.lib.sarx_memset(ctx.h, d_slot.ptr, 0, nslot * 4)
    plan.focus_dev(d_in, d_img)
    assert not d_slot.download(np.float32, slot.shape).any()      # switched off
    with pytest.raises(sx.SarxError):
        plan.set_look_slot(3, d_slot.ptr)
    plan.close()


@pytest.mark.parametrize("n_az,n_rg", [(512, 1024), (64, 128), (2048, 256)])
def test_fused_max_and_masked_ati(sx, ctx, n_az, n_rg):
    """sarx_csa_plan_set_max_slot + sarx_ati_dpca_masked_dev: max|image| emitted by the focus equals the maximum the ATI launch
    reduces from the finished image (same hypotf of the same floats), and the phase plane masked inside the ATI launch equals
    ATI + mask as two launches, bit for bit (sar_ati_dcpa_sim_csa.py:414-419, 447-449)."""
    from sarx import _ffi
    raw, k = orc.point_scene(n_az, n_rg, seed=n_az + 3, clutter_db=-10.0)
    args = orc.focus_args(k)
    plan = _plan(sx, ctx, n_az, n_rg, args, flags=_ffi.FUSE_RANGE)
    px = n_az * n_rg
    d_in, d_in2 = ctx.to_device(raw), ctx.to_device(np.roll(raw, 1, axis=0) * np.complex64(0.9 + 0.1j))
    s1, s2 = ctx.alloc(px * 8), ctx.alloc(px * 8)
    d_max = ctx.alloc(_ffi.MAX_SLOT_BYTES)
    ctx.lib.sarx_memset(ctx.h, d_max.ptr, 0xFF, _ffi.MAX_SLOT_BYTES)  # stale contents must not survive
    plan.set_max_slot(d_max)
    plan.focus_dev(d_in, s1)
    plan.set_max_slot(None)
    plan.focus_dev(d_in2, s2)                                        # must not touch d_max
    img = s1.download(np.complex64, (n_az, n_rg))
    shards = d_max.download(np.float32, (256, 32))
    assert not shards[:, 1:].any()
    got_max = shards[:, 0].max()
    outs = {kk: ctx.alloc(px * 4) for kk in ("ati_phase", "slc1_mag", "dpca_mag")}
    mx, _ = ctx.ati_dpca(s1, s2, px, 0.0, outs)
    assert np.float32(mx) == got_max and got_max > 0
    assert abs(float(got_max) - np.abs(img).max()) <= 2e-7 * float(got_max)
    two = ctx.alloc(px * 4)
    ctx.mask_phase_frac(outs["ati_phase"], outs["slc1_mag"], px, 0.05, two)
    ref = {kk: outs[kk].download(np.float32, (px,)) for kk in ("slc1_mag", "dpca_mag")}
    ref_masked = two.download(np.float32, (px,))
    assert 0 < np.count_nonzero(ref_masked) < px                     # the mask does something
    outs2 = {kk: ctx.alloc(px * 4) for kk in ("ati_phase", "slc1_mag", "dpca_mag")}
    ctx.ati_dpca_masked(s1, s2, px, 0.0, d_max, 0.05, outs2)
    np.testing.assert_array_equal(outs2["ati_phase"].download(np.float32, (px,)), ref_masked)
    for kk in ref:
        np.testing.assert_array_equal(outs2[kk].download(np.float32, (px,)), ref[kk])
    mx2, sm2 = ctx.ati_stats()
    assert mx2 == mx
    plan.close()


@pytest.mark.parametrize("n_az,n_rg,keep", [(1024, 256, False), (2048, 512, True), (64, 128, False), (128, 64, True), (256, 1024, False),
                                             (32, 32, True)])
def test_ati_products_fused_into_second_focus(sx, ctx, n_az, n_rg, keep):
    """sarx_csa_plan_set_ati: the second channel's last azimuth launch emits masked ATI phase, |slc1| and the DPCA magnitude -
    bit for bit what the ATI launch computes from the two finished images (sar_ati_dcpa_sim_csa.py:414-419, 447-449) - with or
    without writing slc2; the phase-balance sum agrees to rounding; sizes without the epilogue say so."""
    from sarx import _ffi
    raw, k = orc.point_scene(n_az, n_rg, seed=n_az + 11, clutter_db=-10.0)
    args = orc.focus_args(k)
    plan = _plan(sx, ctx, n_az, n_rg, args, flags=_ffi.FUSE_RANGE)
    px = n_az * n_rg
    d_in, d_in2 = ctx.to_device(raw), ctx.to_device(np.roll(raw, 1, axis=0) * np.complex64(0.9 + 0.1j))
    s1, s2, s2f = ctx.alloc(px * 8), ctx.alloc(px * 8), ctx.alloc(px * 8)
    d_max = ctx.alloc(_ffi.MAX_SLOT_BYTES)
    plan.set_max_slot(d_max)
    plan.focus_dev(d_in, s1)
    plan.set_max_slot(None)
    plan.focus_dev(d_in2, s2)
    ref = {kk: ctx.alloc(px * 4) for kk in ("ati_phase", "slc1_mag", "dpca_mag")}
    ctx.ati_dpca_masked(s1, s2, px, 0.3, d_max, 0.05, ref)
    mx, sm = ctx.ati_stats()
    got = {kk: ctx.alloc(px * 4) for kk in ("ati_phase", "slc1_mag", "dpca_mag")}
    ctx.lib.sarx_memset(ctx.h, s2f.ptr, 0, px * 8)
    plan.set_ati(s1, d_max, 0.05, 0.3, got["ati_phase"], got["slc1_mag"], got["dpca_mag"], keep_image=keep)
    plan.focus_dev(d_in2, s2f)
    with pytest.raises(sx.SarxError):
        plan.focus_dev(d_in2, s1)                                    # the first channel's image is not a valid scratch / output
    plan.set_ati(None)
    mx2, sm2 = ctx.ati_stats()
    for kk in ref:
        np.testing.assert_array_equal(got[kk].download(np.float32, (px,)), ref[kk].download(np.float32, (px,)))
    assert 0 < np.count_nonzero(ref["ati_phase"].download(np.float32, (px,))) < px
    assert mx2 == mx and abs(sm2 - sm) <= 1e-12 * abs(sm)
    if keep:
        np.testing.assert_array_equal(s2f.download(np.complex64, (px,)), s2.download(np.complex64, (px,)))
    plan.focus_dev(d_in2, s2f)                                       # switched off again: an ordinary focus
    np.testing.assert_array_equal(s2f.download(np.complex64, (px,)), s2.download(np.complex64, (px,)))
    plan.close()
    for shape, flags in (((256, 32), _ffi.FUSE_RANGE), ((64, 64), _ffi.FUSE_RANGE | _ffi.OUT_RG_MAJOR)):   # half-wave tiles; corner-turned output
        other = _plan(sx, ctx, *shape, orc.focus_args(orc.scaled_radar(*shape)), flags=flags)
        with pytest.raises(sx.SarxError):
            other.set_ati(s1, d_max, 0.05, 0.0, got["ati_phase"], got["slc1_mag"], got["dpca_mag"])
        other.close()


def test_errors_are_loud(sx, ctx):
    k = orc.scaled_radar(64, 64)
    with pytest.raises(sx.SarxError):
        sx.CsaPlan(ctx, 20000, 64, *orc.focus_args(k))             # non power of two n_az > 16384
    with pytest.raises(sx.SarxError):
        sx.CsaPlan(ctx, 1, 64, *orc.focus_args(k))
    with pytest.raises(sx.SarxError):
        sx.CsaPlan(ctx, 64, 40000, *orc.focus_args(k))
    with pytest.raises(ValueError):
        sx.sar_focus_csa(np.zeros(16, np.complex64), *orc.focus_args(k))
    bad = list(orc.focus_args(k))
    bad[3] = 0.0
    with pytest.raises(sx.SarxError):
        sx.CsaPlan(ctx, 64, 64, *bad)


@pytest.mark.parametrize("n_az,n_rg", [(64, 16384), (32, 8192), (16384, 16), (8192, 32)])
def test_longest_lines_each_pass(sx, ctx, n_az, n_rg):
    """The kernels the 16384^2 benchmark actually runs (32-point/thread range pass, fused
    wave-private range pass, 128x128 four-step azimuth pass), on a thin scene the oracle can hold."""
    from sarx import _ffi
    raw = _rand((n_az, n_rg), 99)
    k = orc.scaled_radar(n_az, n_rg)
    args = orc.focus_args(k)
    _, _, _, (s1, s2, s3, s4) = orc.sar_focus_csa(raw, *args, return_stages=True)
    plan = _plan(sx, ctx, n_az, n_rg, args)
    d_a, d_b = ctx.alloc(raw.nbytes), ctx.alloc(raw.nbytes)

    def run(pid, src):
        d_a.upload(src.astype(np.complex64))
        plan.run_pass(pid, d_a, d_b)
        return d_b.download(np.complex64, (n_az, n_rg))

    assert orc.rel_l2(run(_ffi.PASS_AZ_FFT_PHI1, raw), s1) < 5e-6
    assert orc.rel_l2(run(_ffi.PASS_RG_FFT_PHI2, s1), s2) < 5e-6
    assert orc.rel_l2(run(_ffi.PASS_RG_IFFT_PHI3, s2), s3) < 5e-6
    assert orc.rel_l2(run(_ffi.PASS_RG_FUSED_23, s1), s3) < 5e-6
    assert orc.rel_l2(run(_ffi.PASS_AZ_IFFT, s3), s4) < 5e-6
    if n_rg == 16384:      # the permuted-spectrum pair of the unfused focus, whole image (fewer lines than persistent workgroups)
        unperm = lambda p: np.ascontiguousarray(p.reshape(n_az, 16, 16, 64).transpose(0, 1, 3, 2)).reshape(n_az, -1)
        perm = lambda x: np.ascontiguousarray(x.reshape(n_az, 16, 64, 16).transpose(0, 1, 3, 2)).reshape(n_az, -1)
        assert orc.rel_l2(unperm(run(_ffi.PASS_RG_FFT_PHI2_PERM, s1)), s2) < 5e-6
        assert orc.rel_l2(run(_ffi.PASS_RG_IFFT_PHI3_PERM, perm(s2)), s3) < 5e-6
    else:
        with pytest.raises(sx.SarxError):
            plan.run_pass(_ffi.PASS_RG_FFT_PHI2_PERM, d_a, d_b)
    for fuse in (True, False):
        img = sx.sar_focus_csa(raw, *args, fuse_range=fuse)[0]
        assert orc.rel_l2(img, s4.T) < 1e-5
    plan.close()


def test_lanes_and_range_cu_share_leave_results_unchanged(sx, ctx):
    """sarx_select_lane / sarx_set_range_cus: a focus enqueued on lane 1, with the persistent 16384-sample range launch sized for
    64 of the CUs, while another frame runs on lane 0, equals the focus on lane 0 with the whole chip bit for bit; sarx_sync waits
    for every lane; a lane out of range is refused."""
    from sarx import _ffi
    n_az, n_rg = 256, 16384
    raw, k = orc.point_scene(n_az, n_rg, seed=21, n_targets=3)
    args = orc.focus_args(k)
    p0, p1 = _plan(sx, ctx, n_az, n_rg, args, flags=_ffi.FUSE_RANGE), sx.CsaPlan(ctx, n_az, n_rg, *args, flags=_ffi.FUSE_RANGE)
    d_in, a0, a1, b1 = ctx.to_device(raw), ctx.alloc(raw.nbytes), ctx.alloc(raw.nbytes), ctx.alloc(raw.nbytes)
    p0.focus_dev(d_in, a0)
    ctx.sync()
    ref = a0.download(np.complex64, raw.shape)
    ctx.set_range_cus(64)
    for _ in range(3):                                   # two frames in flight, alternating lanes
        ctx.select_lane(0)
        p0.focus_dev(d_in, a1)
        ctx.select_lane(1)
        p1.focus_dev(d_in, b1)
    ctx.select_lane(0)
    ctx.set_range_cus(0)
    ctx.lanes_join()
    ctx.sync()
    np.testing.assert_array_equal(a1.download(np.complex64, raw.shape), ref)
    np.testing.assert_array_equal(b1.download(np.complex64, raw.shape), ref)
    with pytest.raises(sx.SarxError):
        ctx.select_lane(4)
    p0.close()
    p1.close()
    for b in (d_in, a0, a1, b1):
        b.release()


def test_lane_probe_and_choice(sx):
    """sarx_probe_lanes: two lanes either run side by side (ratio ~1) or share a hardware queue and take turns (~2);
    Context.concurrent_lanes picks lanes of the first kind for frames in flight, lane 0 first, and remembers its choice."""
    ctx = sx.Context(0)
    ratios = {(a, b): ctx.probe_lanes(a, b) for a in range(4) for b in range(a + 1, 4)}
    # structure only: the ratios are host wall-clock quotients of 300 us launches, and one scheduling hiccup on a busy box must not
    # fail the suite for a reason that has nothing to do with correctness
    assert all(math.isfinite(r) and r > 0 for r in ratios.values()), ratios
    two = ctx.concurrent_lanes(2)
    assert two[0] == 0 and len(two) == 2 and two[1] in (1, 2, 3)
    assert ctx.concurrent_lanes(2) == two and ctx.concurrent_lanes(1) == [0]         # the choice is cached
    assert len(set(ctx.concurrent_lanes(3))) == 3
    with pytest.raises(sx.SarxError):
        ctx.probe_lanes(1, 1)
    ctx.close()


def test_focus_lanes_helper(sx, ctx):
    """sarx.FocusLanes: a frame loop with two frames in flight returns every frame's image bit-identical to CsaPlan.focus_dev,
    alternates lanes, and leaves lane 0 selected with the range launch's CU share reset after finish()."""
    n_az, n_rg = 4096, 4096
    raws, k = [], None
    for seed in (1, 2, 3):
        raw, k = orc.point_scene(n_az, n_rg, seed=seed, n_targets=3)
        raws.append(raw)
    args = orc.focus_args(k)
    plan = sx.CsaPlan(ctx, n_az, n_rg, *args, flags=sx._ffi.FUSE_RANGE)
    d_in = [ctx.to_device(r) for r in raws]
    d_ref, d_out = ctx.alloc(raws[0].nbytes), [ctx.alloc(raws[0].nbytes) for _ in raws]
    fl = sx.FocusLanes(ctx, n_az, n_rg, *args)
    assert fl.lanes == 2 and sx.FocusLanes(ctx, 256, 256, *args).lanes == 1
    used = [fl.focus_dev(d_in[i], d_out[i]) for i in range(3)]
    fl.finish()
    assert used == [0, 1, 0]
    for i in range(3):
        plan.focus_dev(d_in[i], d_ref)
        np.testing.assert_array_equal(d_out[i].download(np.complex64, raws[i].shape), d_ref.download(np.complex64, raws[i].shape))
    fl.close()
    plan.close()
    for b in (*d_in, d_ref, *d_out):
        b.release()


def test_lanes_with_different_plan_kinds_in_flight(sx, ctx):
    """Four lanes, a different kind of plan on each - power-of-two fused, power-of-two unfused with the RG-major corner turn, a
    chirp-z (any-size) plan, and a two-channel product stage - enqueued round-robin without any host synchronisation: every
    result equals its one-lane result bit for bit (per-plan scratch and per-lane reduction scratch never collide)."""
    from sarx import _ffi
    jobs = []
    for n_az, n_rg, flags, seed in [(512, 1024, _ffi.FUSE_RANGE, 1), (256, 512, _ffi.OUT_RG_MAJOR, 2), (96, 80, _ffi.FUSE_RANGE, 3)]:
        raw, k = orc.point_scene(n_az, n_rg, seed=seed, n_targets=3)
        plan = sx.CsaPlan(ctx, n_az, n_rg, *orc.focus_args(k), flags=flags)
        jobs.append({"plan": plan, "in": ctx.to_device(raw), "out": ctx.alloc(raw.nbytes), "shape": raw.shape})
    (r1, r2), k = orc.point_scene(256, 1024, seed=4, clutter_db=-20.0, two_channel=True)
    p2 = sx.CsaPlan(ctx, 256, 1024, *orc.focus_args(k), flags=_ffi.FUSE_RANGE)
    tc = {"d1": ctx.to_device(r1), "d2": ctx.to_device(r2), "s1": ctx.alloc(r1.nbytes), "s2": ctx.alloc(r1.nbytes),
          "d_max": ctx.alloc(_ffi.MAX_SLOT_BYTES), "planes": [ctx.alloc(r1.size * 4) for _ in range(3)]}

    def two_channel():
        p2.set_max_slot(tc["d_max"])
        p2.focus_dev(tc["d1"], tc["s1"])
        p2.set_max_slot(None)
        p2.set_ati(tc["s1"], tc["d_max"], 0.05, 0.0, *tc["planes"])
        p2.focus_dev(tc["d2"], tc["s2"])
        p2.set_ati(None)

    def fetch():
        outs = [j["out"].download(np.complex64, (j["shape"][0] * j["shape"][1],)) for j in jobs]
        return outs + [b.download(np.float32, r1.shape) for b in tc["planes"]], ctx.ati_stats()
    for j in jobs:                                            # one lane, one after the other
        j["plan"].focus_dev(j["in"], j["out"])
    two_channel()
    ctx.sync()
    ref, ref_stats = fetch()
    for j in jobs:
        sx._ffi.check(ctx.lib.sarx_memset(ctx.h, j["out"].ptr, 0, j["shape"][0] * j["shape"][1] * 8), ctx.h)
    for rep in range(3):                                      # four lanes, no host synchronisation in between
        for lane, j in enumerate(jobs):
            ctx.select_lane(lane)
            j["plan"].focus_dev(j["in"], j["out"])
        ctx.select_lane(3)
        two_channel()
    got, stats = fetch()                                      # downloads wait for every lane; ati_stats reads lane 3's scratch
    ctx.select_lane(0)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    assert stats == ref_stats
    for j in jobs:
        j["plan"].close(); j["in"].release(); j["out"].release()
    p2.close()
    for b in (tc["d1"], tc["d2"], tc["s1"], tc["s2"], tc["d_max"], *tc["planes"]):
        b.release()


def test_rccl_allgather_single_rank(sx, ctx):
    """The RCCL path end to end on one GPU: communicator of one rank, gather = copy, on the comm stream."""
    from sarx.batch import RcclStackComm
    x = np.arange(1 << 16, dtype=np.float32)
    d_s, d_r = ctx.to_device(x), ctx.alloc(x.nbytes)
    comm = RcclStackComm(ctx, 1, 0)
    comm.all_gather_dev(d_s, d_r, x.nbytes)
    comm.finish()
    np.testing.assert_array_equal(d_r.download(np.float32, x.shape), x)
    ctx.lib.sarx_comm_destroy(ctx.h)


def test_global_max_reduction_and_allreduce_single_rank(sx, ctx):
    """The two halves of the stack's global normalisation (sar_batch_sim.py:337-338): sarx_max_abs_f32_dev folds max|x| of an
    fp32 buffer into a device float (negative values, a tail that is not a multiple of four, accumulation over several
    buffers), sarx_allreduce_max_dev is ncclAllReduce(max) on the comm stream - with a one-rank communicator the identity."""
    from sarx import _ffi
    from sarx.batch import RcclStackComm
    rng = np.random.default_rng(9)
    x = rng.standard_normal(1_000_003).astype(np.float32)
    x[777_001] = -9.5
    y = rng.standard_normal(4099).astype(np.float32) * 0.1
    d_x, d_y, d_m = ctx.to_device(x), ctx.to_device(y), ctx.alloc(4)
    _ffi.check(ctx.lib.sarx_memset(ctx.h, d_m.ptr, 0, 4), ctx.h)
    ctx.max_abs(d_y, y.size, d_m)
    assert d_m.download(np.float32, (1,))[0] == np.abs(y).max()
    ctx.max_abs(d_x, x.size, d_m)
    ctx.max_abs(d_y, y.size, d_m)                       # a smaller buffer afterwards does not lower it
    assert d_m.download(np.float32, (1,))[0] == np.float32(9.5)
    # any 4-byte alignment of the buffer (a stack slot of odd pixel count starts 4, 8 or 12 bytes past a 16-byte boundary), extreme
    # value in the scalar head, in the body and in the tail
    class _At:                                           # a device address inside d_x
        def __init__(self, ptr): self.ptr = ptr
    for off in (1, 2, 3):
        for pos in (0, 5000, 9996):
            z = x[off:off + 9997].copy()
            z[pos] = 50.0 + off + pos
            _ffi.check(ctx.lib.sarx_memcpy_h2d(ctx.h, d_x.ptr + 4 * off, z.ctypes.data, z.nbytes), ctx.h)
            _ffi.check(ctx.lib.sarx_memset(ctx.h, d_m.ptr, 0, 4), ctx.h)
            ctx.max_abs(_At(d_x.ptr + 4 * off), z.size, d_m)
            assert d_m.download(np.float32, (1,))[0] == np.abs(z).max(), (off, pos)
    # a NaN in the frame comes out as a NaN, like np.max(np.abs(frame)) (sar_batch_sim.py:337)
    z = x[:4099].copy()
    z[1234] = np.nan
    _ffi.check(ctx.lib.sarx_memcpy_h2d(ctx.h, d_x.ptr, z.ctypes.data, z.nbytes), ctx.h)
    _ffi.check(ctx.lib.sarx_memset(ctx.h, d_m.ptr, 0, 4), ctx.h)
    ctx.max_abs(d_x, z.size, d_m)
    assert np.isnan(d_m.download(np.float32, (1,))[0]) and np.isnan(np.max(np.abs(z)))
    _ffi.check(ctx.lib.sarx_memset(ctx.h, d_m.ptr, 0, 4), ctx.h)
    ctx.max_abs(d_y, y.size, d_m)
    _ffi.check(ctx.lib.sarx_memcpy_h2d(ctx.h, d_x.ptr, x.ctypes.data, x.nbytes), ctx.h)
    ctx.max_abs(d_x, x.size, d_m)
    with pytest.raises(sx.SarxError):
        ctx.allreduce_max(d_m, 1)                       # no communicator yet
    comm = RcclStackComm(ctx, 1, 0)
    ctx.allreduce_max(d_m, 1)
    comm.finish()
    assert d_m.download(np.float32, (1,))[0] == np.float32(9.5)
    ctx.lib.sarx_comm_destroy(ctx.h)
    for b in (d_x, d_y, d_m):
        b.release()


def test_host_pipeline_bit_identical_and_bounded(sx, ctx):
    """sarx_csa_focus_host_begin / _end, sar_focus_csa_async and focus_stream (the frame loop of sar_batch_sim.py:303-331 on host arrays,
    frame i+1 uploading while frame i focuses and downloads): every frame bit-identical to the synchronous sar_focus_csa call, in order,
    with pageable and page-locked results, `out=`, three frames' worth of tickets refused, and the plan usable synchronously afterwards."""
    n_az, n_rg = 2048, 4096                                  # 64 MiB per frame: the staged copy threads and the page-locked pool are in play
    raws, k = [], None
    for seed in (11, 12, 13, 14, 15):
        raw, k = orc.point_scene(n_az, n_rg, seed=seed, n_targets=3) if seed == 11 else (None, k)
        if raw is None:
            r = np.random.default_rng(seed)
            raw = (r.standard_normal((n_az, n_rg), dtype=np.float32) + 1j * r.standard_normal((n_az, n_rg), dtype=np.float32)).astype(np.complex64)
        raws.append(raw)
    args = orc.focus_args(k)
    ref = [sx.sar_focus_csa(r, *args, ctx=ctx)[0].copy() for r in raws]
    assert orc.rel_l2(np.abs(ref[0]), np.abs(orc.sar_focus_csa(raws[0], *args)[0])) < TOL
    assert ctx.reserve_pinned((n_az, n_rg), np.complex64, 3) >= 1
    outs = list(sx.focus_stream(iter(raws), *args, ctx=ctx))
    assert len(outs) == len(raws)
    for i, (img, ra, ca) in enumerate(outs):
        np.testing.assert_array_equal(img, ref[i])
        assert img.shape == (n_rg, n_az) and ra.shape == (n_rg,) and ca.shape == (n_az,)
    # futures by hand, two pending; a third is refused by the library (two frames per plan), and the pending ones still complete
    f0 = sx.sar_focus_csa_async(raws[0], *args, ctx=ctx)
    f1 = sx.sar_focus_csa_async(raws[1], *args, ctx=ctx)
    with pytest.raises(sx.SarxError):
        sx.sar_focus_csa_async(raws[2], *args, ctx=ctx)
    np.testing.assert_array_equal(f1.result()[0], ref[1])    # any order
    np.testing.assert_array_equal(f0.result()[0], ref[0])
    assert f0.done() and f1.done()
    np.testing.assert_array_equal(f0.result()[0], ref[0])    # result() twice: same arrays, no second wait
    # a pageable `out` (ordinary NumPy memory): downloaded by result(), same bytes
    mine = np.empty((n_rg, n_az), dtype=np.complex64, order="F")
    f2 = sx.sar_focus_csa_async(raws[2], *args, ctx=ctx, out=mine)
    got = f2.result()[0]
    assert np.shares_memory(got, mine)
    np.testing.assert_array_equal(got, ref[2])
    # futures dropped without result(): the pipeline and the download slots are given back (no "two frames in flight" afterwards)
    for _ in range(3):
        sx.sar_focus_csa_async(raws[4], *args, ctx=ctx)
    import gc
    gc.collect()
    fa, fb = sx.sar_focus_csa_async(raws[0], *args, ctx=ctx), sx.sar_focus_csa_async(raws[1], *args, ctx=ctx)
    np.testing.assert_array_equal(fa.result()[0], ref[0])
    np.testing.assert_array_equal(fb.result()[0], ref[1])
    # the synchronous call on the same cached plan afterwards
    np.testing.assert_array_equal(sx.sar_focus_csa(raws[3], *args, ctx=ctx)[0], ref[3])
    # C ABI argument checking
    plan = sx.CsaPlan(ctx, 256, 256, *args)
    t = C.c_int(7)
    assert ctx.lib.sarx_csa_focus_host_begin(plan.h, None, None, C.byref(t)) != 0 and t.value in (-1, 7)
    assert ctx.lib.sarx_csa_focus_host_end(plan.h, 0) != 0 and ctx.lib.sarx_csa_focus_host_end(plan.h, 5) != 0
    slot = C.c_int(0)
    pageable = np.empty(1024, np.uint8)
    d = ctx.alloc(1024)
    assert ctx.lib.sarx_memcpy_d2h_begin(ctx.h, pageable.ctypes.data, d.ptr, 1024, C.byref(slot)) != 0      # needs page-locked memory
    assert "page-locked" in ctx.last_error()
    assert ctx.lib.sarx_memcpy_d2h_end(ctx.h, 3) != 0
    d.release()
    plan.close()


def test_two_channel_host_inputs_overlap_transfers_same_results(sx, ctx):
    """focus_ati_dpca on HOST arrays (sar_ati_dcpa_sim_csa.py:402-419,447-449 as the script calls it): channel 2 uploads while channel 1
    focuses, slc1 downloads meanwhile, every plane's download is in flight at once - the results equal the device-array path's."""
    from sarx.engine import DeviceArray
    n = 2048
    (r1, r2), k = orc.point_scene(n, n, seed=77, two_channel=True, n_targets=4)
    args = orc.focus_args(k)
    host = sx.focus_ati_dpca(r1, r2, *args, ctx=ctx, pulse_shift=False)
    d1, d2 = DeviceArray(ctx.to_device(r1), (n, n)), DeviceArray(ctx.to_device(r2), (n, n))
    dev = sx.focus_ati_dpca(d1, d2, *args, ctx=ctx, pulse_shift=False)
    for key in ("slc1", "slc2", "slc1_mag", "dpca_mag", "ati_phase_masked"):
        np.testing.assert_array_equal(host[key], dev[key])
    assert host["max_mag"] == dev["max_mag"] and host["sum_interf"] == dev["sum_interf"]
    o1 = orc.sar_focus_csa(r1, *args)[0]
    assert orc.rel_l2(host["slc1"], o1) < TOL
    d1.release(); d2.release()


def test_host_pipeline_edges(sx, ctx):
    """focus_stream / sar_focus_csa_async at the edges: an empty iterable, a single frame, the reference's complex128 arrays, a size
    that is not a power of two (chirp-z route, reference fixture 96 x 80), frames of changing shape inside one stream - every result
    equal to the synchronous call's and, for the fixture, to the reference's own output."""
    g = load_golden("csa_96x80.npz")
    args = tuple(float(v) for v in g["args"])
    raw = g["phist"].astype(np.complex128)                             # the dtype the reference's arrays have
    assert raw.shape == (96, 80)
    assert list(sx.focus_stream(iter(()), *args, ctx=ctx)) == []
    one = list(sx.focus_stream([raw], *args, ctx=ctx))
    assert len(one) == 1
    sync = sx.sar_focus_csa(raw, *args, ctx=ctx)
    np.testing.assert_array_equal(one[0][0], sync[0])
    np.testing.assert_array_equal(one[0][1], sync[1])
    assert orc.rel_l2(one[0][0], g["img_T"]) < TOL                      # against the reference's own image
    # frames of changing shape: each shape has its own cached plan and its own two-deep pipeline
    raw_b, k = orc.point_scene(128, 64, seed=5, n_targets=3)
    args_b = orc.focus_args(k)
    ref_b = sx.sar_focus_csa(raw_b, *args_b, ctx=ctx)[0]
    fa = sx.sar_focus_csa_async(raw, *args, ctx=ctx)
    fb = sx.sar_focus_csa_async(raw_b, *args_b, ctx=ctx)
    fc = sx.sar_focus_csa_async(raw.astype(np.complex64), *args, ctx=ctx)
    np.testing.assert_array_equal(fb.result()[0], ref_b)
    np.testing.assert_array_equal(fa.result()[0], sync[0])
    assert orc.rel_l2(fc.result()[0], sync[0]) < 1e-6                  # complex64 input: the same rounding happens on the host instead
    with pytest.raises(ValueError):
        sx.sar_focus_csa_async(raw[0], *args, ctx=ctx)                 # not 2-D
