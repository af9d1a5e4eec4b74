"""Size-independent properties at BASELINE.json's full sizes (the oracle cannot hold these scenes):
the CSA chain is unitary (unnormalised FFT / 1/N IFFT pairs, unit-modulus phases), linear and
deterministic; identical channels give zero DPCA and zero ATI phase.  Data never leaves the GPU."""
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
    # linearity: focus(x) - focus(x) == 0 everywhere, via the DPCA difference of the two runs
    mx, _ = ctx.ati_dpca(y, z, px, 0.0, scratch)
    diff = scratch["dpca_mag"].download(np.float32, (4, n))
    assert mx > 0 and float(diff.max()) == 0.0
    for b in (x, y, z, *scratch.values()):
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
