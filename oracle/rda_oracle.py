"""CPU oracle for the Range-Doppler focuser.  TEST INFRASTRUCTURE ONLY.

NumPy complex128 restatement of ``sar_focus_rda`` (sar_satellite_sim.py:356-448; the same function
is pasted into sar_satellite_moving_sim.py:208 and sar_vehicle_sim.py:182).  Pinned by
``tests/golden/rda_*.npz``, which ``oracle/make_golden.py`` produces by running the reference's own
function (AST-extracted) on seeded inputs; ``tests/test_oracle_golden.py`` holds this file to them.
"""
from __future__ import annotations

import numpy as np

C_LIGHT = 299792458      # sar_satellite_sim.py:359 (an int there)


def hamming(m):
    """scipy.signal.windows.hamming(m), symmetric."""
    if m == 1:
        return np.ones(1)
    n = np.arange(m)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * n / (m - 1))


def matched_filter(pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz):
    """Hamming-weighted, unit-norm conjugate chirp (:377-385)."""
    step = 1 / sample_rate_hz
    n_mf = int(np.floor(pulse_width_sec / step)) + 1
    t = np.linspace(-pulse_width_sec / 2, pulse_width_sec / 2, n_mf)
    mf = np.conj(np.exp(1j * np.pi * chirp_rate_hzpsec * t ** 2)) * hamming(n_mf)
    return mf / np.linalg.norm(mf)


def conv_same(x, h):
    """scipy.signal.convolve(x, h, mode='same') along axis 0 of x [N x K]: the N samples of the full
    convolution starting at (len(h)-1)//2."""
    n, l = x.shape[0], h.size
    m = 1
    while m < n + l - 1:
        m <<= 1
    full = np.fft.ifft(np.fft.fft(x, m, axis=0) * np.fft.fft(h, m)[:, None], axis=0)
    s = (l - 1) // 2
    return full[s:s + n]


def sar_focus_rda(phist, center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz, prf_hz,
                  platform_speed_mps, range_grp_m, variant="satellite"):
    """phist: [num_ranges x num_pulses].  Returns the reference's 7-tuple (:447-448); variant="moving": the 3-tuple of
    sar_satellite_moving_sim.py:208-285; variant="vehicle": the 8-tuple of sar_vehicle_sim.py:182-273 (range_doppler_filtered,
    :268, in front of doppler_freq)."""
    phist = np.asarray(phist, dtype=np.complex128)
    c = C_LIGHT
    n_r, n_p = phist.shape
    slow = (np.arange(n_p) - (n_p / 2 if n_p % 2 == 0 else (n_p - 1) / 2)) / prf_hz                    # :363-366
    t_grp = 2 * range_grp_m / c
    fast = (np.arange(n_r) - (n_r / 2 if n_r % 2 == 0 else (n_r - 1) / 2)) / sample_rate_hz + t_grp    # :370-373
    # 1 range compression (:375-392)
    pc = conv_same(phist, matched_filter(pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz))
    # 2 azimuth window, range-Doppler transform (:396-399)
    rd = np.fft.fftshift(np.fft.fft(np.fft.fftshift(pc * hamming(n_p)[None, :], axes=1), axis=1), axes=1)
    if n_p % 2 == 0:
        fd = np.arange(-n_p / 2, n_p / 2) * (prf_hz / n_p)                                             # :402-405
    else:
        fd = np.arange(-(n_p - 1) / 2, (n_p - 1) / 2 + 1) * (prf_hz / n_p)
    r_axis = fast * c / 2                                                                                # :407
    # 3 RCMC by linear interpolation (:411-427): profile sampled at r*(1-alpha_k), read back at r
    lam = c / (c / center_wavelength_m)
    vr = platform_speed_mps
    d_r = (r_axis[:, None] * (fd[None, :] ** 2) * lam ** 2) / (8 * vr ** 2)
    rcmc = np.zeros_like(rd)
    for k in range(n_p):
        xs = r_axis - d_r[:, k]
        if n_r > 1:
            rcmc[:, k] = (np.interp(r_axis, xs, rd[:, k].real, left=0, right=0)
                          + 1j * np.interp(r_axis, xs, rd[:, k].imag, left=0, right=0))
        else:
            rcmc[:, k] = rd[:, k]
    # 4 azimuth compression (:431-435)
    ka = (2 * vr ** 2) / (lam * r_axis)
    filt = rcmc * np.exp(-1j * np.pi * ((1.0 / ka)[:, None] * fd[None, :] ** 2))
    # 5 image (:438-446)
    img = np.fft.ifftshift(np.fft.ifft(np.fft.ifftshift(filt, axes=1), axis=1), axes=1)
    if variant == "moving":
        return (np.abs(img).T, r_axis - np.mean(r_axis), vr * slow)
    if variant == "vehicle":
        return (np.abs(img).T, r_axis - np.mean(r_axis), vr * slow, pc, rd, rcmc, filt, fd)
    return (np.abs(img).T, r_axis - np.mean(r_axis), vr * slow, pc, rd, rcmc, fd)


def rda_scene(n_r, n_p, seed=0):
    """Seeded [n_r x n_p] complex64 input plus RDA arguments scaled so the pulse fits the window."""
    from . import csa_oracle as orc
    k = orc.scaled_radar(n_p, n_r, chirp_fill=0.3)
    rng = np.random.default_rng(seed)
    raw, _ = orc.point_scene(n_p, n_r, seed=seed, n_targets=4, clutter_db=-25.0)
    phist = np.ascontiguousarray(raw.T) + 0.01 * (rng.standard_normal((n_r, n_p)) + 1j * rng.standard_normal((n_r, n_p)))
    args = (k["Lambda"], k["T_p"], k["Kr"], k["FS"], k["PRF"], k["V_eff"], k["R0"])
    return phist.astype(np.complex64), args
