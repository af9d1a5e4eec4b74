"""Size-independent properties at BASELINE.json's full sizes (the oracle cannot hold these scenes whole; sampled rows
and columns of them are checked against the oracle in test_gpu_benchsize.py): the CSA chain is unitary (unnormalised
FFT / 1/N IFFT pairs, unit-modulus phases), linear and deterministic; identical channels give zero DPCA and zero ATI
phase.  Data never leaves the GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _energy(ctx, buf, n, scratch):
    """sum |x|^2 through the ATI kernel's fp64 reduction of x * conj(x)."""
    _, s = ctx.ati_dpca(buf, buf, n, 0.0, scratch)
    return s.real


@pytest.mark.parametrize("fuse", [True, False])
def test_16384_unitary_linear_deterministic(fuse):
    import sarx
    from sarx import _ffi, radar
    ctx = sarx.default_context()
    n = 16384
    px = n * n
    plan = sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE if fuse else 0)
    x, y, z = ctx.alloc(px * 8), ctx.alloc(px * 8), ctx.alloc(px * 8)
    scratch = {k: ctx.alloc(px * 4) for k in ("ati_phase", "slc1_mag", "dpca_mag")}
    ctx.fill_noise(x, px, 123)
    e_in = _energy(ctx, x, px, scratch)
    assert abs(e_in / (2.0 * px) - 1.0) < 1e-3               # unit-variance complex noise
    plan.focus_dev(x, y)
    e_out = _energy(ctx, y, px, scratch)
    assert abs(e_out / e_in - 1.0) < 2e-5                    # Parseval through all four passes
    rows = y.download(np.complex64, (8, n))
    assert np.isfinite(rows).all()
    plan.focus_dev(x, z)                                     # determinism, bit for bit
    np.testing.assert_array_equal(rows, z.download(np.complex64, (8, n)))
    # two runs of the same input agree everywhere (DPCA difference of the two images)
    mx, _ = ctx.ati_dpca(y, z, px, 0.0, scratch)
    diff = scratch["dpca_mag"].download(np.float32, (4, n))
    assert mx > 0 and float(diff.max()) == 0.0
    for b in (x, y, z, *scratch.values()):
        b.release()
    plan.close()


def test_16384_linearity():
    """focus(x + y) = focus(x) + focus(y) at the benchmark size.  Sums are formed on the device by the DPCA-difference
    output with a calibration phase of pi (slc1 - slc2 * exp(i pi) = slc1 + slc2); the residual's energy by the ATI
    kernel's fp64 reduction."""
    import sarx
    from sarx import _ffi, radar
    ctx = sarx.default_context()
    n = 16384
    px = n * n
    plan = sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE)
    x, y, z, fx, fy, fz = (ctx.alloc(px * 8) for _ in range(6))
    planes = {k: ctx.alloc(px * 4) for k in ("ati_phase", "slc1_mag", "dpca_mag")}
    ctx.fill_noise(x, px, 11)
    ctx.fill_noise(y, px, 12)

    def add(a, b, out):                       # out = a + b
        ctx.ati_dpca(a, b, px, np.pi, dict(planes, dpca_diff=out), want_stats=False)

    def sub(a, b, out):                       # out = a - b
        ctx.ati_dpca(a, b, px, 0.0, dict(planes, dpca_diff=out), want_stats=False)

    add(x, y, z)
    probe = z.download(np.complex64, (2, n)) - (x.download(np.complex64, (2, n)) + y.download(np.complex64, (2, n)))
    assert np.abs(probe).max() == 0.0         # the device-side sum is the fp32 sum
    plan.focus_dev(x, fx)
    plan.focus_dev(y, fy)
    plan.focus_dev(z, fz)
    add(fx, fy, x)                            # x := focus(x) + focus(y)   (inputs no longer needed)
    sub(fz, x, y)                             # y := focus(x + y) - (focus(x) + focus(y))
    e_res = _energy(ctx, y, px, planes)
    e_img = _energy(ctx, fz, px, planes)
    assert e_img > 0 and np.sqrt(e_res / e_img) < 3e-6, np.sqrt(e_res / e_img)
    for b in (x, y, z, fx, fy, fz, *planes.values()):
        b.release()
    plan.close()


def test_8192_two_channel_identities():
    import sarx
    from sarx import _ffi, radar
    ctx = sarx.default_context()
    n = 8192
    px = n * n
    plan = sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE)
    raw, s1, s2 = ctx.alloc(px * 8), ctx.alloc(px * 8), ctx.alloc(px * 8)
    outs = {k: ctx.alloc(px * 4) for k in ("ati_phase", "slc1_mag", "dpca_mag")}
    ctx.fill_noise(raw, px, 7)
    plan.focus_dev(raw, s1)
    plan.focus_dev(raw, s2)
    mx, sm = ctx.ati_dpca(s1, s2, px, 0.0, outs)
    assert mx > 0 and abs(sm.imag) < 1e-6 * sm.real          # sum |s|^2 is real
    assert float(outs["dpca_mag"].download(np.float32, (16, n)).max()) == 0.0
    # a*conj(a) through an fma leaves the rounding residue of a.x*a.y in the imaginary part: ~1e-8 rad
    assert float(np.abs(outs["ati_phase"].download(np.float32, (16, n))).max()) < 1e-6
    # a calibration phase of pi/3 on channel 2 shows up as exactly -pi/3 of ATI phase everywhere
    ctx.ati_dpca(s1, s2, px, np.pi / 3, outs, want_stats=False)
    ph = outs["ati_phase"].download(np.float32, (16, n))
    mag = outs["slc1_mag"].download(np.float32, (16, n))
    sel = mag > 1e-3 * mx
    assert np.allclose(ph[sel], -np.pi / 3, atol=2e-3)
    for b in (raw, s1, s2, *outs.values()):
        b.release()
    plan.close()
