#!/usr/bin/env python3
"""The reference's moving-ship script (sar_satellite_moving_sim.py: a stationary destroyer and four headings at 15 m/s under
the 350 km orbit) with its hot sections on the MI355X through sarx: echo synthesis of the moving scatterers (:322), ocean
noise (:328), Range-Doppler focus (:331).  Orbit, target, scenarios and the output files follow the reference; the only
changes are the imports and a seed for the noise.

    python examples/sar_satellite_moving_gpu.py [--pulses 7200] [--outdir .] [--scenarios stationary,moving_45deg]

Writes one .npz per scenario with the keys of sar_satellite_moving_sim.py:337-353 (what sar_satellite_moving_viewer.py opens).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx import radar  # noqa: E402
from sarx.targets import generate_destroyer  # noqa: E402


def rotate_points(points, angle_deg):                                 # :22-28
    th = np.radians(angle_deg)
    c, s = np.cos(th), np.sin(th)
    return points @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]).T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pulses", type=int, default=None, help="default ceil(1.2 s * PRF) = 7200 (:69-71)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--scenarios", default="", help="comma-separated names; default: all five (:293-299)")
    a = ap.parse_args()

    k = radar.reference_constants()                                   # :31-58, the constants of the CSA script
    PRF, T_p, BW, Lambda, R0, V_eff = k["PRF"], k["T_p"], k["BW"], k["Lambda"], k["R0"], k["V_eff"]
    T_int = 1.2
    num_pulses = a.pulses or int(np.ceil(T_int * PRF))                # :69-71
    if num_pulses % 2 != 0:
        num_pulses += 1
    T_int = num_pulses / PRF if a.pulses else T_int
    t_vec = np.linspace(-T_int / 2, T_int / 2, num_pulses)            # :76
    pos_sat, _ = radar.orbit_track(t_vec, k)                          # :79-94
    base_targets = generate_destroyer(center_pos=(0, 0, 0))           # :102-107 (no initial rotation in this script)
    chirp_rate, ship_speed = BW / T_p, 15.0                           # :288-291
    scenarios = [("stationary", 0, 0.0, "sar_satellite_moving_scen_stationary.npz"),
                 ("moving_0deg", 0, ship_speed, "sar_satellite_moving_scen_0deg.npz"),
                 ("moving_45deg", 45, ship_speed, "sar_satellite_moving_scen_45deg.npz"),
                 ("moving_90deg", 90, ship_speed, "sar_satellite_moving_scen_90deg.npz"),
                 ("moving_135deg", 135, ship_speed, "sar_satellite_moving_scen_135deg.npz")]   # :293-299
    wanted = [s for s in a.scenarios.split(",") if s]
    inc_deg = np.degrees(np.arcsin((k["R_sat"] / k["Re"]) * np.sin(np.radians(45.0))))
    for name, angle, speed, fname in scenarios:
        if wanted and name not in wanted:
            continue
        pos_rot = rotate_points(np.array([t["position"] for t in base_targets]), angle)       # :305-312
        current = [dict(t, position=pos_rot[j]) for j, t in enumerate(base_targets)]
        th = np.radians(angle)
        velocity = [speed * np.cos(th), speed * np.sin(th), 0.0]                               # :315-319
        t0 = time.time()
        d_raw, t_start, fs_val = sarx.run_moving_physics(current, t_vec, pos_sat, velocity, device=True)   # :322
        snr_db, gain_db = sarx.calculate_snr_db(R0, 50000.0, Lambda, BW, T_int)                # :325-326
        sarx.add_ocean_noise(d_raw, snr_db, seed=a.seed)                                       # :328, in place
        img, r_ax, cr_ax = sarx.sar_focus_rda(d_raw.T, Lambda, T_p, chirp_rate, fs_val, PRF, V_eff, R0, variant="moving")   # :331
        d_raw.release()
        print(f"{name}: heading {angle} deg, {speed} m/s; gain {gain_db:.1f} dB, SNR {snr_db:.1f} dB; echo + noise + focus "
              f"{time.time() - t0:.2f} s ({img.shape[0]} x {img.shape[1]})")
        out = os.path.join(a.outdir, fname)
        np.savez(out, final_image=img, range_axis=r_ax, cross_range=cr_ax, orbit_alt=k["h"], orbit_vel=k["V_sat"], look_ang=45.0,
                 inc_ang=inc_deg, r0=R0, v_eff=V_eff, prf=PRF, scen_name=name, ship_speed=speed, ship_heading=angle,
                 ship_vel=velocity)                                                            # :337-353
        print("wrote", out)


if __name__ == "__main__":
    main()
