"""Ocean noise model on the GPU (sar_satellite_sim.py:331-344, sar_batch_sim.py:66-82).  The reference draws
from unseeded global generators, so there is no sample-level fixture: parity is on the moments the model
defines (thermal variance, clutter power, K-distribution intensity moments, uniform phase) - "parity unpinned"
at the sample level, pinned at the distribution level."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _noise(n, ref_power, snr_db, scr_db, k_nu, seed):
    import sarx
    ctx = sarx.default_context()
    d = ctx.to_device(np.zeros(n, dtype=np.complex64))
    sarx.add_noise_dev(d, n, ref_power, snr_db, scr_db, k_nu, seed)
    x = d.download(np.complex64, (n,))
    d.release()
    return x


def test_thermal_only_moments():
    x = _noise(1 << 20, 4.0, 6.0, None, 1.0, 1)
    p_n = 4.0 / 10 ** 0.6
    assert abs(np.mean(np.abs(x) ** 2) / p_n - 1) < 0.01
    assert abs(np.var(x.real) / (p_n / 2) - 1) < 0.01 and abs(np.var(x.imag) / (p_n / 2) - 1) < 0.01
    assert abs(np.mean(x)) < 0.01 * np.sqrt(p_n)
    assert abs(np.mean(np.abs(x) ** 4) / (2 * p_n ** 2) - 1) < 0.03          # complex Gaussian: E|x|^4 = 2 P^2


@pytest.mark.parametrize("k_nu", [0.5, 1.0, 4.0])
def test_k_clutter_moments(k_nu):
    """intensity I = Pc * G * E: E[I] = Pc, E[I^2] = 2 Pc^2 (1 + 1/nu); thermal made negligible (SNR 80 dB)."""
    pc = 2.0 / 10 ** 1.0
    x = _noise(1 << 21, 2.0, 80.0, 10.0, k_nu, 7)
    i = np.abs(x.astype(np.complex128)) ** 2
    assert abs(np.mean(i) / pc - 1) < 0.02
    assert abs(np.mean(i ** 2) / (2 * pc ** 2 * (1 + 1 / k_nu)) - 1) < 0.08
    ph = np.angle(x)
    assert abs(np.mean(np.exp(1j * ph))) < 0.005


def test_k_clutter_shape_one_is_the_exponential_texture():
    """K_NU = 1.0 in every script of the reference: Gamma(1, 1) = Exp(1), drawn as -log(u) beside the speckle from one hash.
    Third intensity moment E[I^3] = Pc^3 E[G^3] E[E^3] = 36 Pc^3, texture and speckle independent of the phase, and the
    real and imaginary parts uncorrelated."""
    pc = 2.0 / 10 ** 1.0
    x = _noise(1 << 22, 2.0, 80.0, 10.0, 1.0, 11).astype(np.complex128)
    i = np.abs(x) ** 2
    assert abs(np.mean(i ** 3) / (36 * pc ** 3) - 1) < 0.12
    assert abs(np.mean(np.log(i / pc)) - 2 * (-0.5772156649)) < 0.01          # E[log G] + E[log E] = -2 gamma_Euler
    ph = np.angle(x)
    assert abs(np.corrcoef(i, np.cos(ph))[0, 1]) < 0.003 and abs(np.corrcoef(i, np.sin(2 * ph))[0, 1]) < 0.003
    assert abs(np.mean(x.real * x.imag)) < 0.005 * pc


def test_reproducible_and_seeded():
    a = _noise(4096, 1.0, 10.0, 10.0, 1.0, 3)
    assert np.array_equal(a, _noise(4096, 1.0, 10.0, 10.0, 1.0, 3))
    assert not np.array_equal(a, _noise(4096, 1.0, 10.0, 10.0, 1.0, 4))


def test_add_ocean_noise_and_power_stats():
    import sarx
    rng = np.random.default_rng(0)
    raw = (rng.standard_normal((64, 512)) + 1j * rng.standard_normal((64, 512))).astype(np.complex64) * 3
    ctx = sarx.default_context()
    d = ctx.to_device(raw)
    mx, mean = sarx.power_stats(d, raw.size)
    d.release()
    p = np.abs(raw.astype(np.complex128)) ** 2
    assert abs(mx / p.max() - 1) < 1e-6 and abs(mean / p.mean() - 1) < 1e-6
    out = sarx.add_ocean_noise(raw, 3.0, seed=5)
    assert out.shape == raw.shape and out.dtype == np.complex64
    added = np.mean(np.abs(out - raw) ** 2)
    want = p.mean() / 10 ** 0.3 + p.mean() / 10 ** 1.0
    assert abs(added / want - 1) < 0.05
    snr, gain = sarx.calculate_snr_db(509e3, 50000.0, 0.031, 500e6, 1.2)
    assert np.isfinite(snr) and abs(gain - 10 * np.log10(4 * np.pi * 3.5 * 0.5 * 0.6 / 0.031 ** 2)) < 1e-9


@pytest.mark.parametrize("ref,scr_db", [("max", 12.0), ("mean", 12.0), ("max", None)])
def test_relative_noise_on_the_device_equals_the_two_call_form(ref, scr_db):
    """sarx.add_noise_rel_dev (levels taken on the device from the buffer's own max / mean power, no host round trip) adds exactly the
    samples of power_stats + add_noise_dev with the levels computed on the host (sar_batch_sim.py:313-314, sar_satellite_sim.py:333-343):
    same partial sums in the same order, same arithmetic."""
    import sarx
    ctx = sarx.default_context()
    rng = np.random.default_rng(5)
    n = 1_000_003
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * rng.uniform(0.1, 3.0, n)).astype(np.complex64)
    d_a, d_b = ctx.to_device(x), ctx.to_device(x)
    mx, mean = sarx.power_stats(d_a, n)
    assert abs(mx - float((np.abs(x.astype(np.complex128)) ** 2).max())) < 1e-5 * mx
    sarx.add_noise_dev(d_a, n, mx if ref == "max" else mean, 9.0, scr_db, 1.5, seed=77)
    sarx.add_noise_rel_dev(d_b, n, 9.0, scr_db, 1.5, seed=77, ref=ref)
    a, b = d_a.download(np.complex64, (n,)), d_b.download(np.complex64, (n,))
    np.testing.assert_array_equal(a, b)
    assert np.abs(a - x).max() > 0
    with pytest.raises(ValueError):
        sarx.add_noise_rel_dev(d_b, n, 9.0, scr_db, 1.5, ref="median")
    d_a.release(); d_b.release()
