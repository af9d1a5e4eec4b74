"""CPU oracle for the CSA focus + ATI/DPCA hot path.  TEST INFRASTRUCTURE ONLY.

This module is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The shipped path (``nis-sar-amtigmti-video_amd/sarx``) must not.

It is a NumPy complex128 restatement of the reference algorithm, written to
follow the reference's own structure (fftshift bookkeeping kept, phases built
in shifted order) so that it is an *independent* check of the product, whose
host code uses the shift-free natural-bin-order formulation.

Parity pin: ``tests/golden/*.npz`` were produced by running the reference's own
``sar_focus_csa`` (AST-extracted by ``oracle/make_golden.py``) on seeded
scenes; ``tests/test_oracle_golden.py`` checks this file against them to
1e-12 relative L2.

Reference citations are file:line under /root/reference (not present on the
GPU box; nothing here reads it at run time).
"""
from __future__ import annotations

import numpy as np

C_LIGHT = 299792458.0  # sar_ati_dcpa_sim_csa.py:18,211


# ---------------------------------------------------------------------------
# A12  scene / radar constants  (sar_ati_dcpa_sim_csa.py:18-38,42,68,407)
# ---------------------------------------------------------------------------
def reference_radar_constants() -> dict:
    """Scalars exactly as the reference script derives them."""
    C = C_LIGHT
    Re = 6371000.0
    h = 350000.0
    R_sat = Re + h
    GM = 3.986004418e14
    V_sat = np.sqrt(GM / R_sat)
    FC = 9.65e9
    BW = 500e6
    lam = C / FC
    PRF = 6000.0
    T_p = 20e-6
    FS = 600e6
    look = np.radians(45.0)
    inc = np.arcsin((R_sat / Re) * np.sin(look))
    gamma = inc - look
    R0 = np.sqrt(Re**2 + R_sat**2 - 2 * Re * R_sat * np.cos(gamma))
    return dict(
        C=C, Re=Re, h=h, R_sat=R_sat, GM=GM, V_sat=float(V_sat), FC=FC, BW=BW,
        Lambda=lam, PRF=PRF, T_p=T_p, FS=FS, gamma_rad=float(gamma), R0=float(R0),
        V_eff=float(V_sat * np.sqrt(Re / R_sat)),   # :68
        Kr=BW / T_p,                                 # :407
        d_rx=float(2 * V_sat / PRF),                 # :42
    )


# ---------------------------------------------------------------------------
# A1-A7  sar_focus_csa  (sar_ati_dcpa_sim_csa.py:202-396)
# ---------------------------------------------------------------------------
def csa_axes(n_az, n_rg, sample_rate_hz, prf_hz, t_start_fast):
    """tau_j, fr_k, fa_i in *natural* FFT order (:217-225)."""
    dt = 1.0 / sample_rate_hz
    tau = t_start_fast + np.arange(n_rg) * dt
    fr = np.fft.fftfreq(n_rg, dt)
    fa = np.fft.fftfreq(n_az, 1.0 / prf_hz)
    return tau, fr, fa


def migration_factors(fa, lam, vr, r_ref):
    """D(fa), Cs(fa), tau_ref(fa)  (:244-249,262).  Negative arguments are SET to 1e-9."""
    arg = 1.0 - (lam * fa / (2.0 * vr)) ** 2
    arg = np.where(arg < 0, 1e-9, arg)
    D = np.sqrt(arg)
    Cs = 1.0 / D - 1.0
    tau_ref = 2.0 * r_ref / (C_LIGHT * D)
    return D, Cs, tau_ref


def sar_focus_csa(phist, center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec,
                  sample_rate_hz, prf_hz, platform_speed_mps, range_ref_m, t_start_fast,
                  return_stages=False):
    """Chirp Scaling focus, same signature/returns as the reference (:202, :396).

    ``pulse_width_sec`` is accepted and unused, as in the reference.
    With ``return_stages`` also returns the four post-phase intermediates in
    *natural* (unshifted) bin order, for per-pass kernel checks.
    """
    fft = np.fft
    phist = np.asarray(phist, dtype=np.complex128)
    n_az, n_rg = phist.shape
    lam, Kr, Vr, R_ref = center_wavelength_m, chirp_rate_hzpsec, platform_speed_mps, range_ref_m
    c = C_LIGHT
    tau, fr, fa = csa_axes(n_az, n_rg, sample_rate_hz, prf_hz, t_start_fast)

    # pass 1: azimuth FFT to range-Doppler, centre zero Doppler (:233-235)
    S = fft.fftshift(fft.fft(phist, axis=0), axes=0)
    fa_s = fft.fftshift(fa)
    D, Cs, tau_ref = migration_factors(fa_s, lam, Vr, R_ref)
    Dc, Csc, trc = D[:, None], Cs[:, None], tau_ref[:, None]
    tr = tau[None, :]
    # Phi_1 chirp scaling (:272-274)
    S *= np.exp(-1j * np.pi * Kr * Csc * (tr - trc) ** 2)
    st1 = fft.ifftshift(S, axes=0).copy() if return_stages else None

    # pass 2: range FFT, centre zero range frequency (:278-281)
    S = fft.fftshift(fft.fft(S, axis=1), axes=1)
    fr_s = fft.fftshift(fr)[None, :]
    # Phi_2 = range compression + bulk RCMC (:318-326)
    S *= np.exp(1j * (np.pi * fr_s**2 / (Kr * (1.0 + Csc)) + 4.0 * np.pi * R_ref * Csc * fr_s / c))
    st2 = fft.ifftshift(fft.ifftshift(S, axes=1), axes=0).copy() if return_stages else None

    # pass 3: range IFFT (:331)
    S = fft.ifft(fft.ifftshift(S, axes=1), axis=1)
    R_vec = c * tau / 2.0                                           # :346
    # Phi_3 = azimuth compression + residual (:359,375-382)
    S *= np.exp(1j * (4.0 * np.pi * R_vec[None, :] * Dc / lam
                      - np.pi * Kr * Csc * (1.0 + Csc) * (tr - 2.0 * R_ref / c) ** 2))
    st3 = fft.ifftshift(S, axes=0).copy() if return_stages else None

    # pass 4: azimuth IFFT (:385)
    img = fft.ifft(fft.ifftshift(S, axes=0), axis=0)

    t_slow = np.arange(n_az) / prf_hz                               # :392-394
    t_slow = t_slow - np.mean(t_slow)
    cross_range_axis = t_slow * Vr
    out = (img.T, R_vec, cross_range_axis)
    if return_stages:
        return out + ((st1, st2, st3, img),)
    return out


def sar_focus_csa_lean(phist, center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec,
                       sample_rate_hz, prf_hz, platform_speed_mps, range_ref_m, t_start_fast,
                       block=512, workers=1):
    """Same arithmetic, shift-free and row-blocked so 8192^2 / 16384^2 fit in RAM.

    Used only as the timed ``cpu_baseline`` ("port") and for large-size spot
    checks; equality with :func:`sar_focus_csa` is asserted in tests.
    ``workers``>1 threads the azimuth FFTs (scipy.fft) and runs the row blocks in a thread pool
    (stated in the bench line as cores).
    """
    import scipy.fft as sfft
    S = np.array(phist, dtype=np.complex128)       # private copy, worked on in place
    n_az, n_rg = S.shape
    lam, Kr, Vr, R_ref = center_wavelength_m, chirp_rate_hzpsec, platform_speed_mps, range_ref_m
    c = C_LIGHT
    tau, fr, fa = csa_axes(n_az, n_rg, sample_rate_hz, prf_hz, t_start_fast)
    D, Cs, tau_ref = migration_factors(fa, lam, Vr, R_ref)
    R_vec = c * tau / 2.0
    S = sfft.fft(S, axis=0, overwrite_x=True, workers=workers)

    def rows(i0):
        sl = slice(i0, min(i0 + block, n_az))
        Csc, Dc, trc = Cs[sl, None], D[sl, None], tau_ref[sl, None]
        blk = S[sl]
        blk *= np.exp(-1j * np.pi * Kr * Csc * (tau[None, :] - trc) ** 2)
        blk = sfft.fft(blk, axis=1, overwrite_x=True, workers=1 if workers > 1 else workers)
        blk *= np.exp(1j * (np.pi * fr[None, :] ** 2 / (Kr * (1.0 + Csc))
                            + 4.0 * np.pi * R_ref * Csc * fr[None, :] / c))
        blk = sfft.ifft(blk, axis=1, overwrite_x=True, workers=1 if workers > 1 else workers)
        blk *= np.exp(1j * (4.0 * np.pi * R_vec[None, :] * Dc / lam
                            - np.pi * Kr * Csc * (1.0 + Csc) * (tau[None, :] - 2.0 * R_ref / c) ** 2))
        S[sl] = blk

    if workers > 1:                                 # row blocks in parallel (NumPy releases the GIL in ufuncs and FFTs)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(rows, range(0, n_az, block)))
    else:
        for i0 in range(0, n_az, block):
            rows(i0)
    S = sfft.ifft(S, axis=0, overwrite_x=True, workers=workers)
    t_slow = np.arange(n_az) / prf_hz
    t_slow = t_slow - np.mean(t_slow)
    return S.T, R_vec, t_slow * Vr


# ---------------------------------------------------------------------------
# Sampled rows / columns of the chain, for scenes the full-array functions above cannot hold
# (16384^2, 8192^2).  Range passes act on each azimuth bin's row alone (:278-382) and azimuth
# passes on each range column alone (:233-274, :385), so a handful of rows / columns of a
# device-resident image is checked exactly.  Same shift bookkeeping as sar_focus_csa.
# ---------------------------------------------------------------------------
def range_chain_rows(s1_rows, bins, n_az, center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec,
                     sample_rate_hz, prf_hz, platform_speed_mps, range_ref_m, t_start_fast):
    """Passes 2 and 3 for the rows of the pass-1 output that belong to azimuth bins ``bins``
    (natural fftfreq indices into an n_az-bin axis).  Returns (st2_rows, st3_rows), natural order."""
    fft = np.fft
    S = np.asarray(s1_rows, dtype=np.complex128)
    n_rg = S.shape[1]
    lam, Kr, Vr, R_ref = center_wavelength_m, chirp_rate_hzpsec, platform_speed_mps, range_ref_m
    c = C_LIGHT
    tau, fr, fa = csa_axes(n_az, n_rg, sample_rate_hz, prf_hz, t_start_fast)
    D, Cs, _ = migration_factors(fa[np.asarray(bins)], lam, Vr, R_ref)
    Dc, Csc = D[:, None], Cs[:, None]
    tr = tau[None, :]
    S = fft.fftshift(fft.fft(S, axis=1), axes=1)                                   # :278-281
    fr_s = fft.fftshift(fr)[None, :]
    S = S * np.exp(1j * (np.pi * fr_s**2 / (Kr * (1.0 + Csc)) + 4.0 * np.pi * R_ref * Csc * fr_s / c))   # :318-326
    st2 = fft.ifftshift(S, axes=1)
    S = fft.ifft(fft.ifftshift(S, axes=1), axis=1)                                 # :331
    R_vec = c * tau / 2.0                                                          # :346
    S = S * np.exp(1j * (4.0 * np.pi * R_vec[None, :] * Dc / lam
                         - np.pi * Kr * Csc * (1.0 + Csc) * (tr - 2.0 * R_ref / c) ** 2))   # :359,375-382
    return st2, S


def azimuth_fft_cols(raw_cols, cols, n_rg, center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec,
                     sample_rate_hz, prf_hz, platform_speed_mps, range_ref_m, t_start_fast):
    """Pass 1 (:233-274) for the range columns ``cols`` of an [n_az x n_rg] echo: raw_cols is
    [n_az x len(cols)].  Returns the pass-1 output columns in natural azimuth-bin order."""
    fft = np.fft
    S = np.asarray(raw_cols, dtype=np.complex128)
    n_az = S.shape[0]
    lam, Kr, Vr, R_ref = center_wavelength_m, chirp_rate_hzpsec, platform_speed_mps, range_ref_m
    tau, _, fa = csa_axes(n_az, n_rg, sample_rate_hz, prf_hz, t_start_fast)
    S = fft.fftshift(fft.fft(S, axis=0), axes=0)                                   # :233-235
    D, Cs, tau_ref = migration_factors(fft.fftshift(fa), lam, Vr, R_ref)
    tr = tau[np.asarray(cols)][None, :]
    S = S * np.exp(-1j * np.pi * Kr * Cs[:, None] * (tr - tau_ref[:, None]) ** 2)  # :272-274
    return fft.ifftshift(S, axes=0)


def azimuth_ifft_cols(s3_cols):
    """Pass 4 (:385) for columns of the pass-3 output given in natural azimuth-bin order."""
    fft = np.fft
    S = fft.fftshift(np.asarray(s3_cols, dtype=np.complex128), axes=0)             # the array the reference holds
    return fft.ifft(fft.ifftshift(S, axes=0), axis=0)


# ---------------------------------------------------------------------------
# A8  DPCA co-registration (sar_ati_dcpa_sim_csa.py:402-403)
# ---------------------------------------------------------------------------
def dpca_pulse_shift(raw_rx1, raw_rx2):
    return raw_rx1[1:, :], raw_rx2[:-1, :]


# ---------------------------------------------------------------------------
# A9-A11  ATI / DPCA products
# ---------------------------------------------------------------------------
def phase_balance(slc1, slc2):
    """cal_phase = angle(mean(slc1*conj(slc2)))  (sar_ati_dcpa_viewer_csa.py:249-250)."""
    return float(np.angle(np.mean(slc1 * np.conj(slc2))))


def ati_dpca(slc1, slc2, mask_frac=0.05, cal_phase=0.0):
    """Products of sar_ati_dcpa_sim_csa.py:414-419,447-449 and the viewer's
    seven-product table (sar_ati_dcpa_viewer_csa.py:42-52)."""
    slc1 = np.asarray(slc1)
    s2 = np.asarray(slc2) * np.exp(1j * cal_phase)          # viewer :43
    ati_interf = slc1 * np.conj(s2)                        # :414
    ati_phase = np.angle(ati_interf)                       # :415
    slc1_mag = np.abs(slc1)                                # :416
    dpca_diff = slc1 - s2                                  # :418
    dpca_mag = np.abs(dpca_diff)                           # :419
    max_mag = float(np.max(slc1_mag))
    mask = slc1_mag > max_mag * mask_frac                  # :447
    ati_phase_masked = np.copy(ati_phase)
    ati_phase_masked[~mask] = 0                            # :448-449
    return {
        "ati_interf": ati_interf, "ati_phase": ati_phase, "slc1_mag": slc1_mag,
        "dpca_diff": dpca_diff, "dpca_mag": dpca_mag, "mask": mask,
        "ati_phase_masked": ati_phase_masked, "max_mag": max_mag,
        "sum_interf": complex(np.sum(slc1 * np.conj(np.asarray(slc2)))),
        # viewer-only products
        "Ch2 Magnitude": np.abs(s2), "Ch1 Phase": np.angle(slc1), "Ch2 Phase": np.angle(s2),
        "DPCA Phase": np.angle(dpca_diff),
    }


# ---------------------------------------------------------------------------
# A13  echo convention (scene generators restated; seeded by the caller)
# ---------------------------------------------------------------------------
def orbit_track(t_vec, consts=None):
    """Great-circle Tx positions/velocities (sar_ati_dcpa_sim_csa.py:50-66)."""
    k = consts or reference_radar_constants()
    R_sat, V_sat, Re, g = k["R_sat"], k["V_sat"], k["Re"], k["gamma_rad"]
    omega = V_sat / R_sat
    S0 = np.array([-R_sat * np.sin(g), 0.0, R_sat * np.cos(g)])
    V_unit = np.array([0.0, 1.0, 0.0])
    wt = omega * np.asarray(t_vec)[:, None]
    pos = S0[None, :] * np.cos(wt) + (R_sat * V_unit)[None, :] * np.sin(wt) + np.array([0, 0, -Re])[None, :]
    vel = (V_sat * V_unit)[None, :] * np.cos(wt) - (S0 * omega)[None, :] * np.sin(wt)
    return pos, vel


def echo_monostatic(targets, pos_sat, n_rg, fs, t_start_fast, fc, kr, t_p, linspace_grid=True, t_vec=None, vel_target=None):
    """run_physics_engine's signal model (sar_satellite_sim.py:254-302); with ``t_vec`` and ``vel_target`` the targets
    move as p0 + v t: run_moving_physics (sar_satellite_moving_sim.py:111-159); run_custom_physics
    (sar_vehicle_sim.py:83-128) is the same model on a 2048-sample grid.

    tau=2d/C, phase_base=-4*pi*FC*d/C, chirp pi*k*(t-tau-Tp/2)^2 gated to
    |t-tau-Tp/2|<=Tp/2.  ``linspace_grid`` reproduces the reference's
    linspace(0, n/fs, n) fast-time grid (step n/((n-1)fs), :254).
    """
    C = C_LIGHT
    fast = np.linspace(0, n_rg / fs, n_rg) if linspace_grid else np.arange(n_rg) / fs
    t_abs = t_start_fast + fast
    p = np.array([t["position"] for t in targets], dtype=np.float64)
    amp = np.sqrt(np.array([t["rcs"] for t in targets], dtype=np.float64))
    raw = np.zeros((len(pos_sat), n_rg), dtype=np.complex128)
    for i in range(len(pos_sat)):
        pi_ = p if vel_target is None else p + np.asarray(vel_target, dtype=np.float64) * t_vec[i]
        dist = np.sqrt(np.sum((pi_ - pos_sat[i]) ** 2, axis=1))
        tau = 2 * dist / C
        phase_base = -4.0 * np.pi * fc * dist / C
        t_local = t_abs[None, :] - tau[:, None]
        mask = np.abs(t_local - t_p / 2) <= t_p / 2
        chirp = np.pi * kr * (t_local - t_p / 2) ** 2
        raw[i] = np.sum(amp[:, None] * np.exp(1j * (phase_base[:, None] + chirp)) * mask, axis=0)
    return raw


def echo_bistatic(targets, t_vec, pos_tx, vel_tx, rx_offset_dist, vel_target, n_rg, fs,
                  t_start_fast, fc, kr, t_p, linspace_grid=True):
    """run_bistatic_physics_gpu's signal model (sar_ati_dcpa_sim_csa.py:113-178).

    Rx at p_tx + v_hat*offset; targets at p0+v*t; tau=(d_tx+d_rx)/C;
    phase -2*pi*FC*tau; same gated chirp as the monostatic model.
    """
    C = C_LIGHT
    fast = np.linspace(0, n_rg / fs, n_rg) if linspace_grid else np.arange(n_rg) / fs
    t_abs = t_start_fast + fast
    p0 = np.array([t["position"] for t in targets], dtype=np.float64)
    amp = np.sqrt(np.array([t["rcs"] for t in targets], dtype=np.float64))
    v_t = np.asarray(vel_target, dtype=np.float64)
    raw = np.zeros((len(t_vec), n_rg), dtype=np.complex128)
    for i in range(len(t_vec)):
        p_tx = pos_tx[i]
        v_dir = vel_tx[i] / np.linalg.norm(vel_tx[i])
        p_rx = p_tx + v_dir * rx_offset_dist
        p_now = p0 + v_t[None, :] * t_vec[i]
        d_tx = np.linalg.norm(p_now - p_tx, axis=1)
        d_rx = np.linalg.norm(p_now - p_rx, axis=1)
        tau = (d_tx + d_rx) / C
        phase_base = -2.0 * np.pi * fc * tau
        t_local = t_abs[None, :] - tau[:, None]
        mask = np.abs(t_local - t_p / 2) <= t_p / 2
        chirp = np.pi * kr * (t_local - t_p / 2) ** 2
        raw[i] = np.sum(amp[:, None] * np.exp(1j * (phase_base[:, None] + chirp)) * mask, axis=0)
    return raw


def destroyer_targets(center_pos=(0.0, 0.0, 0.0)):
    """The 35-scatterer destroyer of vehicle_targets.py:102-141 (data restated)."""
    cx, cy, cz = center_pos
    length, width = 154.0, 20.0
    out = []
    for x in np.linspace(-length / 2, length / 2, 5):
        for y in np.linspace(-width / 2, width / 2, 3):
            out.append({"position": [cx + x, cy + y, cz + 1], "rcs": 1000.0})
            out.append({"position": [cx + x, cy + y, cz + 6], "rcs": 1000.0})
    out.append({"position": [cx + length * 0.2, cy, cz + 15], "rcs": 5000.0})
    out.append({"position": [cx + length * 0.1, cy, cz + 25], "rcs": 3000.0})
    out.append({"position": [cx - length * 0.1, cy, cz + 12], "rcs": 3000.0})
    out.append({"position": [cx + length / 2.0 + 10.0, cy, cz + 6], "rcs": 1000.0})
    out.append({"position": [cx - length / 2.0 - 5.0, cy, cz + 6], "rcs": 1000.0})
    return out


# ---------------------------------------------------------------------------
# seeded synthetic scenes shared by golden generation, tests and bench
# ---------------------------------------------------------------------------
def scaled_radar(n_az, n_rg, consts=None, chirp_fill=0.45):
    """Radar parameters for an n_az x n_rg window.

    Keeps the reference's carrier, sample rate, PRF, geometry and bandwidth,
    but shortens the pulse so the whole chirp fits the n_rg-sample window
    (the reference's 20 us pulse needs 12000 samples).  Kr = BW/T_p follows.
    """
    k = dict(consts or reference_radar_constants())
    window = n_rg / k["FS"]
    T_p = min(k["T_p"], chirp_fill * window)
    k["T_p"] = T_p
    k["Kr"] = k["BW"] / T_p
    k["t_start_fast"] = 2 * k["R0"] / k["C"] - window / 2
    k["n_az"], k["n_rg"] = n_az, n_rg
    return k


def point_scene(n_az, n_rg, seed=0, n_targets=5, clutter_db=None, consts=None, two_channel=False,
                mover_speed=15.0, frame_time=0.0):
    """Seeded point-target scene.

    Returns (raw [n_az x n_rg] complex64, params dict) or, for ``two_channel``,
    ((raw1, raw2) already DPCA pulse-shifted to n_az rows each, params).
    """
    k = scaled_radar(n_az, n_rg, consts)
    rng = np.random.default_rng(seed)
    n_pulses = n_az + (1 if two_channel else 0)
    T_int = n_pulses / k["PRF"]
    t_vec = np.linspace(-T_int / 2, T_int / 2, n_pulses)
    pos_tx, vel_tx = orbit_track(t_vec, k)
    # scene extent that stays inside the window after range migration
    rg_half = 0.25 * (n_rg / k["FS"] - k["T_p"]) * k["C"] / 2 / np.sin(np.radians(45.0))
    az_half = 0.25 * n_az / k["PRF"] * k["V_eff"]
    targets = []
    for _ in range(n_targets):
        targets.append({"position": [rng.uniform(-rg_half, rg_half), rng.uniform(-az_half, az_half), 0.0],
                        "rcs": float(rng.uniform(1.0, 1000.0))})
    common = dict(n_rg=n_rg, fs=k["FS"], t_start_fast=k["t_start_fast"], fc=k["FC"], kr=k["Kr"], t_p=k["T_p"])

    def add_clutter(raw, sub):
        if clutter_db is None:
            return raw
        r2 = np.random.default_rng([seed, sub])
        p = np.mean(np.abs(raw) ** 2) * 10 ** (clutter_db / 10)
        return raw + np.sqrt(p / 2) * (r2.standard_normal(raw.shape) + 1j * r2.standard_normal(raw.shape))

    if not two_channel:
        raw = echo_monostatic(targets, pos_tx, linspace_grid=True, **common)
        return add_clutter(raw, 0).astype(np.complex64), k
    stationary = np.zeros(3)
    mover = [{"position": [0.3 * rg_half, -0.2 * az_half, 0.0], "rcs": 2000.0}]
    v_m = np.array([mover_speed, 0.0, 0.0])
    mover[0]["position"][0] += mover_speed * frame_time
    chans = []
    for off in (-k["d_rx"] / 2, k["d_rx"] / 2):
        r = echo_bistatic(targets, t_vec, pos_tx, vel_tx, off, stationary, **common)
        r += echo_bistatic(mover, t_vec, pos_tx, vel_tx, off, v_m, **common)
        chans.append(r)
    r1, r2 = dpca_pulse_shift(chans[0], chans[1])
    r1, r2 = add_clutter(r1, 1), add_clutter(r2, 2)
    return (r1.astype(np.complex64), r2.astype(np.complex64)), k


def focus_args(k):
    """Positional argument tuple after ``phist`` for sar_focus_csa (:410)."""
    return (k["Lambda"], k["T_p"], k["Kr"], k["FS"], k["PRF"], k["V_eff"], k["R0"], k["t_start_fast"])


def rel_l2(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
