"""Radar/scene scalars of the reference's two-channel CSA script
(sar_ati_dcpa_sim_csa.py:18-38,42,68,112,407), as host-side constants."""
from __future__ import annotations

import math

C = 299792458.0


def reference_constants():
    Re, h, GM = 6371000.0, 350000.0, 3.986004418e14
    R_sat = Re + h
    V_sat = math.sqrt(GM / R_sat)                       # :23
    FC, BW, PRF, T_p, FS = 9.65e9, 500e6, 6000.0, 20e-6, 600e6
    look = math.radians(45.0)
    inc = math.asin((R_sat / Re) * math.sin(look))       # :35
    gamma = inc - look
    R0 = math.sqrt(Re**2 + R_sat**2 - 2 * Re * R_sat * math.cos(gamma))   # :38
    return dict(C=C, Re=Re, h=h, R_sat=R_sat, GM=GM, V_sat=V_sat, FC=FC, BW=BW, Lambda=C / FC, PRF=PRF,
                T_p=T_p, FS=FS, gamma_rad=gamma, R0=R0, V_eff=V_sat * math.sqrt(Re / R_sat),   # :68
                Kr=BW / T_p, d_rx=2 * V_sat / PRF)                                              # :407, :42


def focus_args(n_rg=None, k=None):
    """Positional arguments of sar_focus_csa after phist (:410) for the reference radar;
    t_start_fast centres an n_rg-sample window on R0 (:112 uses the 22 us window)."""
    k = k or reference_constants()
    if n_rg is None:
        t0 = 2 * k["R0"] / C - k["T_p"] / 2 - 1e-6       # :112
    else:
        t0 = 2 * k["R0"] / C - (n_rg / k["FS"]) / 2
    return (k["Lambda"], k["T_p"], k["Kr"], k["FS"], k["PRF"], k["V_eff"], k["R0"], t0)


def scaled_constants(n_az, n_rg, chirp_fill=0.45, k=None):
    """The reference radar with the pulse shortened so that the whole chirp fits an n_rg-sample window (the reference's 20 us
    pulse needs 12000 of its 13200 samples): carrier, bandwidth, sample rate, PRF and geometry unchanged, Kr = BW / T_p follows.
    What the power-of-two BASELINE configurations use when a scene has real point targets in it."""
    k = dict(k or reference_constants())
    window = n_rg / k["FS"]
    k["T_p"] = min(k["T_p"], chirp_fill * window)
    k["Kr"] = k["BW"] / k["T_p"]
    k["t_start_fast"] = 2 * k["R0"] / C - window / 2
    k["n_az"], k["n_rg"] = n_az, n_rg
    return k


def c3_scene(frame=0, frame_dt=0.1):
    """SURVEY.md 8(d) C3 / C5 content: a 5 x 5 grid of stationary scatterers, one radial mover at 15 m/s (velocity_ship,
    sar_ati_dcpa_sim_csa.py:184) and one slow mover, the movers advanced by frame * frame_dt seconds (VideoSAR cadence,
    cf. sar_batch_sim.py:244-252).  Returns [(targets, velocity), ...] as run_bistatic_physics_gpu takes them."""
    grid = [{"position": [float(x), float(y), 0.0], "rcs": 100.0 + 10.0 * i} for i, (x, y) in
            enumerate((gx, gy) for gx in (-60.0, -30.0, 0.0, 30.0, 60.0) for gy in (-1500.0, -750.0, 0.0, 750.0, 1500.0))]
    groups = [(grid, [0.0, 0.0, 0.0])]
    for p0, rcs, vel in (([20.0, -400.0, 0.0], 2000.0, [15.0, 0.0, 0.0]), ([-35.0, 700.0, 0.0], 1500.0, [2.0, 0.0, 0.0])):
        t = frame * frame_dt
        groups.append(([{"position": [p0[0] + vel[0] * t, p0[1] + vel[1] * t, p0[2] + vel[2] * t], "rcs": rcs}], vel))
    return groups


def orbit_track(t_vec, k=None):
    """Great-circle transmitter positions and velocities over the slow-time vector
    (sar_ati_dcpa_sim_csa.py:50-66), as [n x 3] arrays."""
    import numpy as np
    k = k or reference_constants()
    R_sat, V_sat, Re, g = k["R_sat"], k["V_sat"], k["Re"], k["gamma_rad"]
    omega = V_sat / R_sat
    S0 = np.array([-R_sat * np.sin(g), 0.0, R_sat * np.cos(g)])
    V_unit = np.array([0.0, 1.0, 0.0])
    wt = omega * np.asarray(t_vec, dtype=np.float64)[:, None]
    pos = S0[None, :] * np.cos(wt) + (R_sat * V_unit)[None, :] * np.sin(wt) + np.array([0.0, 0.0, -Re])[None, :]
    vel = (V_sat * V_unit)[None, :] * np.cos(wt) - (S0 * omega)[None, :] * np.sin(wt)
    return pos, vel
