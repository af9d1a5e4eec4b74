"""Range-Doppler focuser on the GPU against the reference's own outputs
(tests/golden/rda_*.npz, written by oracle/make_golden.py from sar_satellite_sim.py:356-448) and against
oracle/rda_oracle.py at sizes the fixtures do not cover.  Tolerance: relative L2 <= 1e-4 for complex64
(the bar BASELINE.json states for the hot path); axes are float64 and compared to 1e-9 relative."""
import os

import numpy as np
import pytest

from oracle import rda_oracle as rda
from oracle.csa_oracle import rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


def _check(out, ref):
    mag, r_ax, c_ax, pc, rd, rc, fd = out
    rmag, rr, rcx, rpc, rrd, rrc, rfd = ref
    assert mag.shape == rmag.shape and pc.shape == rpc.shape
    assert rel_l2(pc, rpc) < TOL, "range compression"
    assert rel_l2(rd, rrd) < TOL, "range-Doppler map"
    assert rel_l2(rc, rrc) < TOL, "RCMC"
    assert rel_l2(mag, rmag) < TOL, "image"
    np.testing.assert_allclose(r_ax, rr, rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(c_ax, rcx, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(fd, rfd, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("name", ["rda_200x96", "rda_257x101", "rda_128x64"])
def test_rda_golden(name):
    import sarx
    g = np.load(os.path.join(GOLD, name + ".npz"))
    out = sarx.sar_focus_rda(g["phist"], *[float(v) for v in g["args"]])
    ref = (g["image_mag_T"], g["range_axis_centered"], g["cross_range_m"], g["phist_compressed"], g["range_doppler"],
           g["range_doppler_rcmc"], g["doppler_freq"])
    _check(out, ref)


def test_rda_the_other_two_copies_golden():
    """variant="moving": sar_satellite_moving_sim.py:208-285's three outputs; variant="vehicle": sar_vehicle_sim.py:182-273's
    eight, range_doppler_filtered (:268) among them - against fixtures each script's own copy wrote.  The vehicle fixture
    has 256 pulses: the power-of-two direct route (one convolution launch, two-step pulse-axis transforms with the window,
    the fftshifts and the magnitude in their ends: six launches)."""
    import sarx
    g = np.load(os.path.join(GOLD, "rda_moving_144x80.npz"))
    out = sarx.sar_focus_rda(g["phist"], *[float(v) for v in g["args"]], variant="moving")
    assert len(out) == 3
    assert rel_l2(out[0], g["image_mag_T"]) < TOL
    np.testing.assert_allclose(out[1], g["range_axis_centered"], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(out[2], g["cross_range_m"], rtol=1e-9, atol=1e-9)
    g = np.load(os.path.join(GOLD, "rda_vehicle_96x256.npz"))
    out = sarx.sar_focus_rda(g["phist"], *[float(v) for v in g["args"]], variant="vehicle")
    assert len(out) == 8
    _check(out[:6] + (out[7],), (g["image_mag_T"], g["range_axis_centered"], g["cross_range_m"], g["phist_compressed"],
                                  g["range_doppler"], g["range_doppler_rcmc"], g["doppler_freq"]))
    assert rel_l2(out[6], g["range_doppler_filtered"]) < TOL, "azimuth compression"
    with pytest.raises(ValueError):
        sarx.sar_focus_rda(g["phist"], *[float(v) for v in g["args"]], variant="airship")


@pytest.mark.parametrize("n_r,n_p", [(1024, 512), (333, 256), (96, 4096)])
def test_rda_power_of_two_direct_route_equals_padded_route(n_r, n_p, monkeypatch):
    """SARX_RDA_DIRECT=0 keeps the route of rounds 2-4 (padded work arrays, fourteen launches) for A/B: every output of the
    six-launch route against it, the azimuth-compressed map included; n_r = 333 has pad columns in the work arrays."""
    import sarx
    phist, args = rda.rda_scene(n_r, n_p, seed=5 * n_r + n_p)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SARX_RDA_DIRECT", mode)
        outs[mode] = sarx.sar_focus_rda(phist, *args, variant="vehicle")
    for i, what in ((0, "image"), (3, "range compression"), (4, "range-Doppler map"), (5, "RCMC"), (6, "azimuth compression")):
        assert np.isfinite(outs["1"][i]).all()
        assert rel_l2(outs["1"][i], outs["0"][i]) < 2e-5, what
    ref = rda.sar_focus_rda(phist, *args, variant="vehicle")
    assert rel_l2(outs["1"][0], ref[0]) < TOL and rel_l2(outs["1"][6], ref[6]) < TOL


def test_rda_airborne_script_size():
    """sar_vehicle_sim.py's own size and radar (:21-41,86-87,166-168,283-287): 2048 range samples x 32768 pulses, 10 GHz,
    300 MHz over 1 us (361 taps), 360 MHz sampling, 2 kHz PRF, 150 m/s.  Point targets + noise, device-resident; the whole
    image and every map against the oracle's complex128 run on the downloaded pulses."""
    import sarx
    n_r, n_p = 2048, 32768
    c = 299792458.0
    args = (c / 10e9, 1.0e-6, 300e6 / 1.0e-6, 360e6, 2000.0, 150.0, 20000.0 / np.cos(np.radians(45.0)))
    r = np.random.default_rng(77)
    raw = (0.05 * (r.standard_normal((n_p, n_r), dtype=np.float32) + 1j * r.standard_normal((n_p, n_r), dtype=np.float32))).astype(np.complex64)
    # three point scatterers: chirp echoes with a quadratic range history over the aperture (what run_custom_physics writes, :83-128)
    t_slow = (np.arange(n_p) - n_p / 2) / args[4]
    fast = (np.arange(n_r) - n_r / 2) / args[3]
    for (dx, dy, amp) in ((0.0, 0.0, 1.0), (40.0, -60.0, 0.7), (-25.0, 90.0, 0.5)):
        rng_t = np.sqrt((args[6] + dx) ** 2 + (args[5] * t_slow - dy) ** 2)
        tau = 2 * (rng_t - args[6]) / c
        for i0 in range(0, n_p, 4096):
            tl = fast[None, :] - tau[i0:i0 + 4096, None]
            m = np.abs(tl) <= args[1] / 2
            ph = -4 * np.pi * rng_t[i0:i0 + 4096, None] / args[0] + np.pi * args[2] * tl ** 2
            raw[i0:i0 + 4096] += (amp * np.exp(1j * ph) * m).astype(np.complex64)
    ctx = sarx.default_context()
    d = sarx.DeviceArray(ctx.to_device(raw), raw.shape)
    out = sarx.sar_focus_rda(d.T, *args, variant="vehicle")
    d.release()
    ref = rda.sar_focus_rda(raw.T, *args, variant="vehicle")
    for i, what in ((0, "image"), (3, "range compression"), (4, "range-Doppler map"), (5, "RCMC"), (6, "azimuth compression")):
        assert rel_l2(out[i], ref[i]) < TOL, what
    assert out[0].max() > 20 * out[0].mean()          # the scatterers focus
    np.testing.assert_allclose(out[7], ref[7], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("n_r,n_p", [(1024, 512), (1500, 700), (4096, 2048)])
def test_rda_oracle(n_r, n_p):
    import sarx
    phist, args = rda.rda_scene(n_r, n_p, seed=n_r + n_p)
    out = sarx.sar_focus_rda(phist, *args)
    _check(out, rda.sar_focus_rda(phist, *args))


def test_rda_native_range_extent():
    """13200 range samples with the reference's 12001-tap matched filter: the convolution runs on 32768-point split lines
    whose zero padding is neither stored nor read."""
    import sarx
    from oracle import csa_oracle as orc
    k = orc.reference_radar_constants()
    r = np.random.default_rng(12)
    phist = (r.standard_normal((13200, 24)) + 1j * r.standard_normal((13200, 24))).astype(np.complex64)
    args = (k["Lambda"], k["T_p"], k["Kr"], k["FS"], k["PRF"], k["V_eff"], k["R0"])
    _check(sarx.sar_focus_rda(phist, *args), rda.sar_focus_rda(phist, *args))


def test_rda_7200_pulses_prime_factor_route():
    """7200 pulses (the satellite scripts' count, sar_satellite_sim.py:83-85): the pulse-axis transforms run as 32 x 225
    prime-factor launches with the Hamming window and both fftshifts folded into their row addresses."""
    import sarx
    phist, args = rda.rda_scene(160, 7200, seed=72)
    _check(sarx.sar_focus_rda(phist, *args), rda.sar_focus_rda(phist, *args))


def test_rda_native_size_direct_route_equals_padded_route(monkeypatch):
    """The satellite script's own size, 13200 samples x 7200 pulses, device-resident: the direct route (one 19683-point
    circular convolution per pulse + 32 x 225 prime-factor transforms: six launches) against the route through 32768-point
    split lines and a 16384-row chirp-z (thirteen launches), which the oracle tests above pin at smaller sizes."""
    import sarx
    n_r, n_p = 13200, 7200
    from oracle import csa_oracle as orc
    k = orc.reference_radar_constants()
    args = (k["Lambda"], k["T_p"], k["Kr"], k["FS"], k["PRF"], k["V_eff"], k["R0"])
    ctx = sarx.default_context()
    buf = ctx.alloc(n_p * n_r * 8)
    ctx.fill_noise(buf, n_p * n_r, 4242)
    d = sarx.DeviceArray(buf, (n_p, n_r))
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SARX_RDA_DIRECT", mode)
        o = sarx.sar_focus_rda(d.T, *args, device_output=True)
        outs[mode] = [o[0].download(np.float32, (n_p, n_r))] + [o[i].download(np.complex64, (n_p, n_r)) for i in (3, 4, 5)]
        for i in (0, 3, 4, 5):
            o[i].release()
    for a, b, what in zip(outs["1"], outs["0"], ("image", "range compression", "range-Doppler map", "RCMC")):
        assert np.isfinite(a).all() and np.abs(a).max() > 0
        assert rel_l2(a, b) < 2e-5, what
    d.release()


def test_rda_view_input_and_no_intermediates():
    """raw.T (a view) as the scripts pass it; intermediates=False returns None for the three maps."""
    import sarx
    phist, args = rda.rda_scene(256, 128, seed=9)
    raw = np.ascontiguousarray(phist.T)
    out = sarx.sar_focus_rda(raw.T, *args, intermediates=False)
    assert out[3] is None and out[4] is None and out[5] is None
    assert rel_l2(out[0], rda.sar_focus_rda(phist, *args)[0]) < TOL


def test_rda_device_input_equals_host_input():
    """sar_focus_rda(d.T, ...) on the DeviceArray of an echo generator = the same call on its NumPy copy."""
    import sarx
    phist, args = rda.rda_scene(300, 128, seed=4)
    raw = np.ascontiguousarray(phist.T)                     # [pulses x ranges], the generators' layout
    ctx = sarx.default_context()
    d = sarx.DeviceArray(ctx.to_device(raw), raw.shape)
    assert d.T.shape == (300, 128) and np.array_equal(d.T.numpy(), raw.T)
    a = sarx.sar_focus_rda(d.T, *args)
    b = sarx.sar_focus_rda(raw.T, *args)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    with pytest.raises(ValueError):
        sarx.sar_focus_rda(d, *args)                        # not the transposed view
    d.release()


@pytest.mark.parametrize("n_r,n_p", [(512, 300), (300, 1000), (256, 256)])
def test_rda_device_outputs_stay_on_the_gpu(n_r, n_p):
    """device_output=True: magnitude and maps come back as device buffers (the C call only enqueues); their content is
    what the host path returns.  (512, 300) and (300, 1000) take the three-launch chirp-z along pulses - shifts, window
    and magnitude folded into its ends - (256, 256) the direct power-of-two route."""
    import sarx
    phist, args = rda.rda_scene(n_r, n_p, seed=n_r * 7 + n_p)
    raw = np.ascontiguousarray(phist.T)
    ctx = sarx.default_context()
    d = sarx.DeviceArray(ctx.to_device(raw), raw.shape)
    host = sarx.sar_focus_rda(raw.T, *args)
    dev = sarx.sar_focus_rda(d.T, *args, device_output=True)
    np.testing.assert_array_equal(dev[0].download(np.float32, (n_p, n_r)), host[0])
    for i in (3, 4, 5):
        np.testing.assert_array_equal(dev[i].download(np.complex64, (n_p, n_r)).T, host[i])
    for i in (1, 2, 6):
        np.testing.assert_array_equal(dev[i], host[i])
    # without the intermediates the RCMC map is not even stored; the image is the same
    lean = sarx.sar_focus_rda(d.T, *args, intermediates=False, device_output=True)
    assert lean[3] is None and lean[5] is None
    np.testing.assert_array_equal(lean[0].download(np.float32, (n_p, n_r)), host[0])
    _check(host, rda.sar_focus_rda(phist, *args))
    for b in (dev[0], dev[3], dev[4], dev[5], lean[0]):
        b.release()
    d.release()
    with pytest.raises(ValueError):
        sarx.sar_focus_rda(raw.T, *args, device_output=True)


def test_rda_errors():
    import sarx
    phist, args = rda.rda_scene(64, 32, seed=1)
    with pytest.raises(ValueError):
        sarx.sar_focus_rda(phist[0], *args)
    bad = list(args); bad[4] = 0.0
    with pytest.raises(sarx.SarxError):
        sarx.sar_focus_rda(phist, *bad)
