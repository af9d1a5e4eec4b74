"""Sizes that are not powers of two: chirp-z path against the reference-generated
fixture and the oracle, including the reference's native extents 7199 (azimuth) and 13200 (range)."""
import numpy as np
import pytest

from conftest import load_golden
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
                                       (48, 13200),      # native range extent: chirp-z over a 32768-point line
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
