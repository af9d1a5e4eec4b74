#!/usr/bin/env python3
"""The reference's airborne script (sar_vehicle_sim.py: 20 km, 150 m/s, 10 GHz, 300 MHz, 32768 pulses x 2048 samples) with
its hot sections on the MI355X through sarx: echo synthesis (:171), ocean noise (:177), Range-Doppler focus (:276-287).
Geometry, target and the output file follow the reference; the only changes are the imports and a seed for the noise.

    python examples/sar_vehicle_rda_gpu.py [--pulses 32768] [--out sar_simulation_data.npz]

Writes the .npz the reference's interactive viewer opens unchanged (keys of sar_vehicle_sim.py:291-307, rd_az_comp among them).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx.targets import generate_destroyer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pulses", type=int, default=32768, help="the script's full synthetic aperture (:41)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="sar_simulation_data.npz")
    a = ap.parse_args()

    C = 299792458.0                                                   # :21-27
    Re = 6378137.0
    Rs = Re + 20000.0
    V_plat, FC = 150.0, 10e9
    Lambda = C / FC
    theta_look_deg = 45.0                                             # :33-37
    h = Rs - Re
    R0 = h / np.cos(np.radians(theta_look_deg))
    tuned_prp = 500e-6                                                # :39-43
    num_pulses = a.pulses
    T_int = num_pulses * tuned_prp
    t_vec = np.linspace(-T_int / 2, T_int / 2, num_pulses)            # :57
    pos = np.zeros((num_pulses, 3))                                   # :60-71: straight track
    pos[:, 0] = -R0 * np.sin(np.radians(theta_look_deg))
    pos[:, 1] = V_plat * t_vec
    pos[:, 2] = R0 * np.cos(np.radians(theta_look_deg))
    sim_targets = generate_destroyer(center_pos=(0, 0, 0))            # :75
    f_c, bw, t_p = 10e9, 300e6, 1.0e-6                                # :166-168

    t0 = time.time()
    d_raw = sarx.run_custom_physics(sim_targets, t_vec, pos, tuned_prp, t_p, f_c, bw, R0=R0, C=C, device=True)   # :171
    snr_db, gain_db = sarx.calculate_snr_db(R0, 50000.0, Lambda, bw, T_int, p_tx=2000.0, ant_l=1.5, ant_w=0.3, t_sys=290.0,
                                            nf_db=4.0, loss_db=3.0)                                              # :130-135,174-175
    sarx.add_ocean_noise(d_raw, snr_db, seed=a.seed)                                                             # :177, in place
    sarx.default_context().sync()
    t_echo = time.time() - t0

    t0 = time.time()
    prf_val, fs_val = 1.0 / tuned_prp, 360e6                          # :277-281
    (sar_image, range_axis, cross_range, phist_comp, rd_map, rd_rcmc, rd_az_comp, doppler_axis) = sarx.sar_focus_rda(
        d_raw.T, C / f_c, t_p, bw / t_p, fs_val, prf_val, V_plat, R0, variant="vehicle")                         # :283-287
    raw_data = d_raw.numpy()
    d_raw.release()
    t_proc = time.time() - t0
    print(f"radar equation: gain {gain_db:.1f} dB, SNR {snr_db:.1f} dB; echo + noise {t_echo:.2f} s, RDA focus {t_proc:.2f} s "
          f"({raw_data.shape[0]} x {raw_data.shape[1]}; all eight outputs downloaded)")
    np.savez(a.out, raw_phist=raw_data.T, range_comp=phist_comp, rd_map=rd_map, rd_rcmc=rd_rcmc, rd_az_comp=rd_az_comp,
             final_image=sar_image, range_axis=range_axis, cross_range=cross_range, doppler_axis=doppler_axis,
             platform_alt=h, platform_vel=V_plat, look_ang=theta_look_deg, inc_ang=theta_look_deg, r0=R0, prf=prf_val)   # :291-307
    print("wrote", a.out)


if __name__ == "__main__":
    main()
