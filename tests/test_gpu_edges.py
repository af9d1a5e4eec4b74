"""Smallest and most lopsided inputs the three focusers accept (2 x 2 images, one pulse, one pixel): the chirp-z paths,
the single-step column transforms and the grid/chunk logic at their limits, against the oracles (<= 1e-4)."""
import numpy as np
import pytest

from oracle import csa_oracle as orc
from oracle import rda_oracle as rda
from oracle import tdbp_oracle as tb

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_az,n_rg", [(2, 2), (2, 3), (3, 2), (16, 2), (2, 16), (1000, 2), (2, 1000), (17, 17)])
def test_csa_tiny(n_az, n_rg):
    import sarx
    r = np.random.default_rng(n_az * 7 + n_rg)
    raw = (r.standard_normal((n_az, n_rg)) + 1j * r.standard_normal((n_az, n_rg))).astype(np.complex64)
    args = orc.focus_args(orc.scaled_radar(max(n_az, 16), max(n_rg, 16)))
    img, rax, cax = sarx.sar_focus_csa(raw, *args)
    ref, orax, ocax = orc.sar_focus_csa(raw, *args)
    assert img.shape == (n_rg, n_az) and orc.rel_l2(img, ref) < 1e-4
    np.testing.assert_allclose(rax, orax, rtol=1e-14)
    np.testing.assert_allclose(cax, ocax, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("n_r,n_p", [(2, 2), (3, 5), (16, 2), (2, 16), (40, 3)])
def test_rda_tiny(n_r, n_p):
    import sarx
    phist, args = rda.rda_scene(max(n_r, 8), max(n_p, 8), seed=1)
    phist = np.ascontiguousarray(phist[:n_r, :n_p])
    a, b = sarx.sar_focus_rda(phist, *args), rda.sar_focus_rda(phist, *args)
    assert a[0].shape == (n_p, n_r)
    for i in (0, 3, 4, 5):
        assert orc.rel_l2(a[i], b[i]) < 1e-4


@pytest.mark.parametrize("n_p,nx,ny", [(1, 1, 1), (2, 3, 1), (5, 1, 7), (33, 17, 16)])
def test_tdbp_tiny(n_p, nx, ny):
    import sarx
    k = tb.scaled_constants()
    sc = tb.tdbp_scene(n_pulses=n_p, seed=2, k=k)
    raw = sc["raw"].astype(np.complex64)
    args = (sc["pos"], sc["vel"], sc["t_start"], sc["num_samples"], sc["v_tgt"], sc["t_vec"], sc["swath"], nx, ny)
    a = sarx.tdbp_gpu(raw, *args, consts=k)
    assert a.shape == (ny, nx) and orc.rel_l2(a, tb.tdbp(raw, *args, k)) < 1e-4
