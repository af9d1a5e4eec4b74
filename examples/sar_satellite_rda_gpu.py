#!/usr/bin/env python3
"""The reference's single-channel satellite script (sar_satellite_sim.py) with its hot sections on the MI355X
through sarx: echo synthesis (:346), ocean noise (:351), Range-Doppler focus (:453-462).  Geometry, target and
the output file follow the reference; the only changes are the imports and a seed for the noise.

    python examples/sar_satellite_rda_gpu.py [--pulses 7200] [--out sar_satellite_data.npz]

Writes the .npz the reference's viewer opens unchanged (keys of sar_satellite_sim.py:483-500).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pulses", type=int, default=None, help="default ceil(1.2 s * PRF) = 7200 (:82-85)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="sar_satellite_data.npz")
    a = ap.parse_args()

    k = radar.reference_constants()                                   # :22-58 (same constants as the CSA script)
    PRF, T_p, BW, FC, Lambda, R0, V_eff = k["PRF"], k["T_p"], k["BW"], k["FC"], k["Lambda"], k["R0"], k["V_eff"]
    T_int = 1.2
    num_pulses = a.pulses or int(np.ceil(T_int * PRF))                # :82-85
    if num_pulses % 2 != 0:
        num_pulses += 1
    T_int = num_pulses / PRF if a.pulses else T_int
    t_vec = np.linspace(-T_int / 2, T_int / 2, num_pulses)            # :90
    pos_sat, _ = sarx.orbit_arc(t_vec, sarx.batch_constants())        # :131-172 (same orbit as sar_batch_sim.py)

    base = generate_destroyer(center_pos=(0, 0, 0))                   # :191
    th = np.radians(90.0)                                             # :194-195: rotate 90 degrees about z
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    sim_targets = [dict(t, position=np.asarray(t["position"]) @ rot.T) for t in base]

    t0 = time.time()
    d_raw, t_start, fs_val = sarx.run_physics_engine(sim_targets, pos_sat, t_vec, device=True)   # :346, stays on the GPU
    snr_db, gain_db = sarx.calculate_snr_db(R0, 50000.0, Lambda, BW, T_int)                   # :349-350
    sarx.add_ocean_noise(d_raw, snr_db, seed=a.seed)                                          # :351, in place
    t_echo = time.time() - t0

    t0 = time.time()
    final_img, r_axis, cross_rng, rc_time_T, rd_map_T, rd_rcmc_T, dop_axis = sarx.sar_focus_rda(
        d_raw.T, Lambda, T_p, BW / T_p, fs_val, PRF, V_eff, R0)                               # :453-462
    raw_data = d_raw.numpy()                                                                  # the file keeps raw_phist (:484)
    d_raw.release()
    t_proc = time.time() - t0
    print(f"radar equation: gain {gain_db:.1f} dB, SNR {snr_db:.1f} dB; echo + noise {t_echo:.2f} s, "
          f"RDA focus {t_proc:.2f} s ({raw_data.shape[0]} x {raw_data.shape[1]}, the echoes are focused where they were synthesised; all seven outputs downloaded)")
    np.savez(a.out, raw_phist=raw_data, range_comp=rc_time_T.T, rd_map=rd_map_T.T, rd_rcmc=rd_rcmc_T.T,
             final_image=final_img, range_axis=r_axis, cross_range=t_vec * V_eff, doppler_axis=dop_axis,
             orbit_alt=k["h"], orbit_vel=k["V_sat"], look_ang=45.0,
             inc_ang=np.degrees(np.arcsin((k["R_sat"] / k["Re"]) * np.sin(np.radians(45.0)))), bw=BW, r0=R0, fc=FC, v_eff=V_eff)                                                # :483-500
    print("wrote", a.out)


if __name__ == "__main__":
    main()
