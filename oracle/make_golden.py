#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own functions.

Runs only in the build container (needs /root/reference).  Nothing from the
reference is copied: each function is pulled out of its script with ``ast``
(the scripts run whole simulations at import, so they cannot be imported),
executed on seeded inputs, and only the numeric inputs/outputs are saved.

    python oracle/make_golden.py            # rewrites tests/golden/

Fixtures (inputs are stored as complex64 so GPU and reference see identical
values; reference outputs are complex128):
  csa_<naz>x<nrg>.npz      sar_focus_csa   (sar_ati_dcpa_sim_csa.py:202-396)
  csa_refconst_*.npz       same, literal reference radar constants (partial chirp)
  csa_digest_1024.npz      1024x1024: peak, sampled rows/cols, norms only
  ati_128x128.npz          two-channel scene -> slc1, slc2 + the literal
                           expressions of :414-419,447-449 and viewer :249-250
  rda_<nr>x<np>.npz        sar_focus_rda             (sar_satellite_sim.py:356-448), all 7 outputs
  rda_moving_<nr>x<np>.npz the copy in sar_satellite_moving_sim.py:208-285, its 3 outputs
  rda_vehicle_<nr>x<np>.npz the copy in sar_vehicle_sim.py:182-273, its 8 outputs (range_doppler_filtered among them)
  destroyer.npz            generate_destroyer        (vehicle_targets.py:102-141)
  echo_mono.npz            run_physics_engine        (sar_satellite_sim.py:211-305)
  echo_bistatic.npz        run_bistatic_physics_gpu  (sar_ati_dcpa_sim_csa.py:106-181)
  echo_moving.npz          run_moving_physics        (sar_satellite_moving_sim.py:111-159)
  echo_vehicle.npz         run_custom_physics        (sar_vehicle_sim.py:83-128)
  spot_<a|b>.npz           run_physics_spotlight     (sar_batch_sim.py:85-169) + calculate_raw_snr_db (:54-64)
  tdbp_<a|b>.npz           tdbp_gpu                  (sar_batch_sim.py:171-238), moving-target and static focus
"""
import ast
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("SARX_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import csa_oracle as orc  # noqa: E402


def extract(script, name, env):
    """Compile one top-level ``def`` of a reference script into ``env``."""
    path = os.path.join(REF, script)
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), env)
            return env[name]
    raise KeyError(f"{name} not found in {script}")


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_focus = extract("sar_ati_dcpa_sim_csa.py", "sar_focus_csa", {"np": np})

    def save_csa(tag, raw64, k):
        args = orc.focus_args(k)
        img_t, rax, cax = ref_focus(raw64.astype(np.complex128), *args)
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), phist=raw64, args=np.array(args, dtype=np.float64),
                 img_T=np.ascontiguousarray(img_t), range_axis=rax, cross_range_axis=cax)
        peak = np.abs(img_t).max() / np.abs(img_t).mean()
        print(f"{tag}: out {img_t.shape} peak/mean {peak:.1f}")

    # focused point-target scenes, pulse shortened to fit the window
    for (naz, nrg, seed, cl) in [(64, 64, 1, None), (96, 80, 2, None), (128, 128, 3, -20.0),
                                 (256, 256, 4, -20.0), (128, 512, 5, None), (512, 128, 6, None)]:
        raw, k = orc.point_scene(naz, nrg, seed=seed, clutter_db=cl)
        save_csa(f"csa_{naz}x{nrg}", raw, k)

    # literal reference constants (20 us pulse: only part of the chirp is in the window)
    k = orc.reference_radar_constants()
    k["t_start_fast"] = 2 * k["R0"] / k["C"] - k["T_p"] / 2 - 1e-6      # sar_ati_dcpa_sim_csa.py:112
    rng = np.random.default_rng(7)
    raw = (rng.standard_normal((128, 256)) + 1j * rng.standard_normal((128, 256))).astype(np.complex64)
    save_csa("csa_refconst_128x256", raw, k)

    # 1024^2 digest
    raw, k = orc.point_scene(1024, 1024, seed=11, clutter_db=-20.0)
    img_t, rax, cax = ref_focus(raw.astype(np.complex128), *orc.focus_args(k))
    mag = np.abs(img_t)
    pk = np.unravel_index(np.argmax(mag), mag.shape)
    rows = np.array([0, 1, 127, 300, 511, 512, 777, 1023])
    np.savez_compressed(os.path.join(OUT, "csa_digest_1024.npz"), seed=11, clutter_db=-20.0,
             args=np.array(orc.focus_args(k)), peak_index=np.array(pk), peak_value=img_t[pk],
             rows=rows, row_values=img_t[rows, :], col_values=np.ascontiguousarray(img_t[:, rows]),
             l2=np.linalg.norm(img_t), sum=img_t.sum(), range_axis=rax, cross_range_axis=cax)
    print("csa_digest_1024: peak", pk, abs(img_t[pk]))

    # two-channel scene -> ATI/DPCA literal expressions
    (r1, r2), k = orc.point_scene(128, 128, seed=21, clutter_db=-25.0, two_channel=True)
    slc1, rax, cax = ref_focus(r1.astype(np.complex128), *orc.focus_args(k))
    slc2, _, _ = ref_focus(r2.astype(np.complex128), *orc.focus_args(k))
    ati_interf = slc1 * np.conj(slc2)                       # :414
    ati_phase = np.angle(ati_interf)                        # :415
    slc1_mag = np.abs(slc1)                                 # :416
    dpca_mag = np.abs(slc1 - slc2)                          # :418-419
    mag_mask = np.abs(slc1) > (np.max(np.abs(slc1)) * 0.05)  # :447
    ati_phase_masked = np.copy(ati_phase)
    ati_phase_masked[~mag_mask] = 0                         # :448-449
    cal_phase = np.angle(np.mean(slc1 * np.conj(slc2)))     # viewer :249-250
    np.savez_compressed(os.path.join(OUT, "ati_128x128.npz"), raw1=r1, raw2=r2, args=np.array(orc.focus_args(k)),
             slc1=np.ascontiguousarray(slc1), slc2=np.ascontiguousarray(slc2), ati_phase=ati_phase,
             slc1_mag=slc1_mag, dpca_mag=dpca_mag, mask=mag_mask, ati_phase_masked=ati_phase_masked,
             cal_phase=cal_phase)
    print("ati_128x128: cal_phase", cal_phase, "mask px", int(mag_mask.sum()))

    # Range-Doppler focuser (f3)
    from scipy.interpolate import interp1d
    from scipy.signal import convolve
    from scipy.signal.windows import hamming
    from oracle import rda_oracle as rda
    ref_rda = extract("sar_satellite_sim.py", "sar_focus_rda", {"np": np, "convolve": convolve, "hamming": hamming,
                                                                 "interp1d": interp1d})
    for (nr, npul, seed) in [(200, 96, 31), (257, 101, 32), (128, 64, 33)]:
        phist, args = rda.rda_scene(nr, npul, seed=seed)
        o = quiet(ref_rda, phist.astype(np.complex128), *args)
        np.savez_compressed(os.path.join(OUT, f"rda_{nr}x{npul}.npz"), phist=phist, args=np.array(args, dtype=np.float64),
                            image_mag_T=o[0], range_axis_centered=o[1], cross_range_m=o[2], phist_compressed=o[3],
                            range_doppler=o[4], range_doppler_rcmc=o[5], doppler_freq=o[6])
        print(f"rda_{nr}x{npul}: peak/mean {o[0].max() / o[0].mean():.1f}")

    # target model: vehicle_targets.py imports cleanly (pure functions)
    sys.path.insert(0, REF)
    import vehicle_targets as vt
    tg = vt.generate_destroyer(center_pos=(3.0, -2.0, 1.0))
    np.savez_compressed(os.path.join(OUT, "destroyer.npz"), center=np.array([3.0, -2.0, 1.0]),
                        pos=np.array([x["position"] for x in tg], dtype=np.float64),
                        rcs=np.array([x["rcs"] for x in tg], dtype=np.float64))

    # echo generators, tiny, with scaled module constants injected as globals
    kk = orc.reference_radar_constants()
    fs_small = 6e6                       # int(22e-6*FS) -> 132 samples
    tgts = [{"position": np.array([30.0, -40.0, 0.0]), "rcs": 10.0},
            {"position": np.array([-80.0, 25.0, 5.0]), "rcs": 250.0}]
    t_vec = np.linspace(-0.01, 0.01, 24)
    pos_tx, vel_tx = orc.orbit_track(t_vec, kk)
    env = {"np": np, "BW": 5e6, "T_p": kk["T_p"], "R0": kk["R0"], "C": kk["C"], "FC": kk["FC"]}
    mono = extract("sar_satellite_sim.py", "run_physics_engine", env)
    # the reference hard-codes fs = 600e6 inside (:245): window is 13200 samples
    raw_m, t0_m, fs_m = quiet(mono, tgts, pos_tx[:3], t_vec[:3])
    np.savez_compressed(os.path.join(OUT, "echo_mono.npz"), raw=raw_m.astype(np.complex128), t_start_fast=t0_m, fs=fs_m,
             pos_sat=pos_tx[:3], tgt_pos=np.array([t["position"] for t in tgts]),
             tgt_rcs=np.array([t["rcs"] for t in tgts]), BW=5e6, T_p=kk["T_p"], FC=kk["FC"])
    print("echo_mono:", raw_m.shape, np.abs(raw_m).max())
    try:
        import torch
        env = {"np": np, "torch": torch, "device": torch.device("cpu"), "FS": fs_small, "BW": 5e6,
               "T_p": kk["T_p"], "R0": kk["R0"], "C": kk["C"], "FC": kk["FC"]}
        bist = extract("sar_ati_dcpa_sim_csa.py", "run_bistatic_physics_gpu", env)
        raw_b, t0_b = quiet(bist, tgts, t_vec, pos_tx, vel_tx, -kk["d_rx"] / 2, np.array([15.0, 0.0, 0.0]))
        np.savez_compressed(os.path.join(OUT, "echo_bistatic.npz"), raw=raw_b, t_start_fast=t0_b, fs=fs_small,
                 t_vec=t_vec, pos_tx=pos_tx, vel_tx=vel_tx, rx_offset=-kk["d_rx"] / 2,
                 vel_target=np.array([15.0, 0.0, 0.0]), tgt_pos=np.array([t["position"] for t in tgts]),
                 tgt_rcs=np.array([t["rcs"] for t in tgts]), BW=5e6, T_p=kk["T_p"], FC=kk["FC"])
        print("echo_bistatic:", raw_b.shape, np.abs(raw_b).max())
    except ImportError:
        print("torch missing: echo_bistatic.npz not regenerated")
    make_mono_variants()
    make_tdbp()


def make_mono_variants():
    """echo_moving.npz / echo_vehicle.npz: run_moving_physics (sar_satellite_moving_sim.py:111-159) and
    run_custom_physics (sar_vehicle_sim.py:83-128); both hard-code their sample grids."""
    kk = orc.reference_radar_constants()
    tgts = [{"position": np.array([30.0, -40.0, 0.0]), "rcs": 10.0},
            {"position": np.array([-80.0, 25.0, 5.0]), "rcs": 250.0},
            {"position": np.array([5.0, 60.0, 12.0]), "rcs": 40.0}]
    t_vec = np.linspace(-0.2, 0.2, 4)
    pos_tx, _ = orc.orbit_track(t_vec, kk)
    env = {"np": np, "BW": 5e6, "T_p": kk["T_p"], "R0": kk["R0"], "C": kk["C"], "FC": kk["FC"]}
    mov = extract("sar_satellite_moving_sim.py", "run_moving_physics", env)
    vel = [12.0, -7.0, 0.0]
    raw, t0, fs = quiet(mov, tgts, t_vec, pos_tx, vel)
    np.savez_compressed(os.path.join(OUT, "echo_moving.npz"), raw=raw, t_start_fast=t0, fs=fs, t_vec=t_vec, pos_sat=pos_tx,
                        vel_target=np.array(vel), tgt_pos=np.array([t["position"] for t in tgts]),
                        tgt_rcs=np.array([t["rcs"] for t in tgts]), BW=5e6, T_p=kk["T_p"], FC=kk["FC"], R0=kk["R0"])
    env = {"np": np, "R0": 9000.0, "C": kk["C"]}
    cus = extract("sar_vehicle_sim.py", "run_custom_physics", env)
    pos = np.stack([np.linspace(-40, 40, 6), np.full(6, -6000.0), np.full(6, 6708.2)], axis=1)
    raw = quiet(cus, tgts, np.arange(6) / 1000.0, pos, 1e-3, 4e-6, 10e9, 150e6)
    np.savez_compressed(os.path.join(OUT, "echo_vehicle.npz"), raw=raw, pos=pos, t_p=4e-6, fc=10e9, bw=150e6, R0=9000.0,
                        tgt_pos=np.array([t["position"] for t in tgts]), tgt_rcs=np.array([t["rcs"] for t in tgts]))
    print("echo_moving:", np.abs(np.load(os.path.join(OUT, "echo_moving.npz"))["raw"]).max(), "echo_vehicle:", np.abs(raw).max())


def make_tdbp():
    """spot_*.npz / tdbp_*.npz: run_physics_spotlight (:85-169) and tdbp_gpu (:171-238) of sar_batch_sim.py,
    torch on the CPU, module constants injected (two scaled radars)."""
    import torch
    from oracle import tdbp_oracle as tb
    cases = [("a", tb.scaled_constants(), dict(n_pulses=96, seed=3, speed=15.0, heading_deg=30.0, swath=400.0), 48, 40),
             ("b", tb.scaled_constants(fs=90e6, t_p=1.5e-6, bw=75e6, prf=4000.0),
              dict(n_pulses=70, seed=5, speed=120.0, heading_deg=135.0, swath=300.0), 33, 37)]
    for tag, k, kw, nx, ny in cases:
        env = {"np": np, "torch": torch, "device": torch.device("cpu")}
        env.update({n: k[n] for n in ("C", "FC", "FS", "T_P", "K_RATE", "R0", "Lambda", "P_TX", "ANT_WIDTH", "T_SYS",
                                      "NF_DB", "LOSS_DB", "K_BOLTZ")})
        spot = extract("sar_batch_sim.py", "run_physics_spotlight", env)
        tdbp = extract("sar_batch_sim.py", "tdbp_gpu", env)
        snr = extract("sar_batch_sim.py", "calculate_raw_snr_db", env)
        sc = tb.tdbp_scene(k=k, **kw)
        raw_t, t_start, n, v_tgt = spot(sc["targets"], sc["t_vec"], sc["pos"], sc["vel"], heading_deg=sc["heading_deg"],
                                        speed=sc["speed"], l_ant=sc["l_ant"])
        raw = raw_t.numpy()
        consts = np.array([k[n] for n in ("C", "FC", "FS", "T_P", "K_RATE", "R0", "Lambda", "PRF", "BW")])
        np.savez_compressed(os.path.join(OUT, f"spot_{tag}.npz"), consts=consts, raw=raw, t_start=t_start, num_samples=n,
                            v_tgt=v_tgt, t_vec=sc["t_vec"], pos=sc["pos"], vel=sc["vel"], l_ant=sc["l_ant"],
                            heading_deg=sc["heading_deg"], speed=sc["speed"],
                            tgt_pos=np.array([t["position"] for t in sc["targets"]]),
                            tgt_rcs=np.array([t["rcs"] for t in sc["targets"]]),
                            snr_db=snr(k["R0"], 5000.0, k["Lambda"], k["BW"], sc["l_ant"]))
        # the input the focuser sees is complex64 (what the GPU path takes); both focus velocities of the script
        raw64 = raw.astype(np.complex64)
        out = {}
        for name, vf in (("mbp", v_tgt), ("stdbp", np.zeros(3))):
            out[name] = tdbp(torch.tensor(raw64.astype(np.complex128)), sc["pos"], sc["vel"], t_start, n, vel_focus=vf,
                             t_pulses=sc["t_vec"], scene_size=sc["swath"], nx=nx, ny=ny)
        np.savez_compressed(os.path.join(OUT, f"tdbp_{tag}.npz"), consts=consts, raw=raw64, t_start=t_start, num_samples=n,
                            v_tgt=v_tgt, t_vec=sc["t_vec"], pos=sc["pos"], vel=sc["vel"], swath=sc["swath"], nx=nx, ny=ny,
                            img_mbp=out["mbp"], img_stdbp=out["stdbp"])
        print(f"tdbp_{tag}: raw {raw.shape}, image {out['mbp'].shape}, peak {np.abs(out['mbp']).max():.4g} / "
              f"{np.abs(out['stdbp']).max():.4g}")


def make_rda_variants():
    """The two other pasted copies of sar_focus_rda, each run as it stands in its own script.  256 pulses: the size class
    of the airborne script's 32768 (power of two, two-step pulse-axis transforms)."""
    from scipy.interpolate import interp1d
    from scipy.signal import convolve
    from scipy.signal.windows import hamming
    from oracle import rda_oracle as rda
    env = {"np": np, "convolve": convolve, "hamming": hamming, "interp1d": interp1d}
    ref_mov = extract("sar_satellite_moving_sim.py", "sar_focus_rda", dict(env))
    ref_veh = extract("sar_vehicle_sim.py", "sar_focus_rda", dict(env))
    nr, npul = 144, 80
    phist, args = rda.rda_scene(nr, npul, seed=41)
    o = quiet(ref_mov, phist.astype(np.complex128), *args)
    assert len(o) == 3
    np.savez_compressed(os.path.join(OUT, f"rda_moving_{nr}x{npul}.npz"), phist=phist, args=np.array(args, dtype=np.float64),
                        image_mag_T=o[0], range_axis_centered=o[1], cross_range_m=o[2])
    print(f"rda_moving_{nr}x{npul}: peak/mean {o[0].max() / o[0].mean():.1f}")
    nr, npul = 96, 256
    phist, args = rda.rda_scene(nr, npul, seed=42)
    o = quiet(ref_veh, phist.astype(np.complex128), *args)
    assert len(o) == 8
    np.savez_compressed(os.path.join(OUT, f"rda_vehicle_{nr}x{npul}.npz"), phist=phist, args=np.array(args, dtype=np.float64),
                        image_mag_T=o[0], range_axis_centered=o[1], cross_range_m=o[2], phist_compressed=o[3],
                        range_doppler=o[4], range_doppler_rcmc=o[5], range_doppler_filtered=o[6], doppler_freq=o[7])
    print(f"rda_vehicle_{nr}x{npul}: peak/mean {o[0].max() / o[0].mean():.1f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rda_variants":      # only these two files (the others are unchanged)
        make_rda_variants()
    else:
        main()
        make_rda_variants()
