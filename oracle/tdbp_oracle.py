"""CPU oracle for the VideoSAR time-domain back-projection path.  TEST INFRASTRUCTURE ONLY.

NumPy restatement of ``run_physics_spotlight`` (sar_batch_sim.py:85-169), ``tdbp_gpu`` (:171-238),
``calculate_raw_snr_db`` (:54-64) and the orbit arc of ``main`` (:258-268).  The reference's functions read
module globals (C, FC, FS, T_P, K_RATE, R0); here they travel in a dict ``k`` (``batch_constants()`` is
the literal block :12-50) so the fixtures can use scaled-down radars.

Pinned by ``tests/golden/tdbp_*.npz`` / ``spot_*.npz``, produced by ``oracle/make_golden.py`` running the
reference's own functions (AST-extracted, torch on CPU) on seeded inputs.

One detail decides parity: the reference interpolates the range-compressed pulse with
``torch.nn.functional.grid_sample`` on FLOAT32 copies of data and coordinates (:226).  The sample position
is therefore quantised (~1e-3 sample at 22004 samples; ignoring that costs 1e-5..1e-4 relative L2 already at
1122 samples); ``_grid_sample_1d`` reproduces the float32 arithmetic of torch's CPU kernel: unnormalise as
``fma(x + 1, W / 2, -0.5)`` (a single rounding - checked bit-for-bit against torch 2.10 CPU at W = 22004,
where the two-rounding form is off by up to 2e-3), bilinear weights ``1 - w`` / ``w``, zeros outside.
"""
from __future__ import annotations

import numpy as np


def batch_constants():
    """sar_batch_sim.py:12-50."""
    k = {}
    k["C"] = 299792458.0
    k["Re"] = 6371000.0
    k["h"] = 350000.0
    k["R_sat"] = k["Re"] + k["h"]
    k["GM"] = 3.986004418e14
    k["V_sat"] = np.sqrt(k["GM"] / k["R_sat"])
    k["FC"] = 9.65e9
    k["BW"] = 500e6
    k["Lambda"] = k["C"] / k["FC"]
    k["T_P"] = 20e-6
    k["K_RATE"] = k["BW"] / k["T_P"]
    k["FS"] = 600e6
    k["PRF"] = 5000.0
    look = np.radians(45.0)
    inc = np.arcsin((k["R_sat"] / k["Re"]) * np.sin(look))
    gamma = inc - look
    k["S0_from_C"] = np.array([0, -k["R_sat"] * np.sin(gamma), k["R_sat"] * np.cos(gamma)])
    k["V_unit"] = np.array([1.0, 0.0, 0.0])
    k["C_offset"] = np.array([0, 0, -k["Re"]])
    k["R0"] = np.linalg.norm(k["S0_from_C"] + k["C_offset"])
    k["P_TX"], k["ANT_WIDTH"], k["T_SYS"], k["NF_DB"], k["LOSS_DB"] = 1000.0, 0.5, 290.0, 5.0, 3.0
    k["K_BOLTZ"] = 1.380649e-23
    return k


def scaled_constants(fs=60e6, t_p=2e-6, bw=50e6, prf=5000.0):
    """Same geometry, smaller waveform: keeps fixtures small (window = 2000/C + T_P + 10 us)."""
    k = batch_constants()
    k["FS"], k["T_P"], k["BW"], k["PRF"] = fs, t_p, bw, prf
    k["K_RATE"] = bw / t_p
    return k


def orbit_arc(t_vec, k):
    """Circular-orbit positions/velocities, sar_batch_sim.py:262-268."""
    omega = k["V_sat"] / k["R_sat"]
    wt = omega * np.asarray(t_vec, dtype=np.float64)[:, None]
    pos = k["S0_from_C"][None, :] * np.cos(wt) + (k["R_sat"] * k["V_unit"])[None, :] * np.sin(wt) + k["C_offset"][None, :]
    vel = (k["V_sat"] * k["V_unit"])[None, :] * np.cos(wt) - (k["S0_from_C"] * omega)[None, :] * np.sin(wt)
    return pos, vel


def calculate_raw_snr_db(r_slant, rcs, wavelength, bandwidth, ant_l, k=None, **kw):
    """sar_batch_sim.py:54-64."""
    k = k or batch_constants()
    p_tx, ant_w = kw.get("p_tx", k["P_TX"]), kw.get("ant_w", k["ANT_WIDTH"])
    t_sys, nf_db, loss_db = kw.get("t_sys", k["T_SYS"]), kw.get("nf_db", k["NF_DB"]), kw.get("loss_db", k["LOSS_DB"])
    gain = 4 * np.pi * (ant_l * ant_w * 0.6) / (wavelength ** 2)
    num = p_tx * (gain ** 2) * (wavelength ** 2) * rcs
    den = ((4 * np.pi) ** 3) * (r_slant ** 4) * k["K_BOLTZ"] * t_sys * bandwidth * (10 ** (loss_db / 10)) * (10 ** (nf_db / 10))
    return 10 * np.log10(num / den)


def spotlight_window(k):
    """num_samples, t_start, t_fast_abs (sar_batch_sim.py:86-91)."""
    win_len = (2000.0 / k["C"]) + k["T_P"] + 10e-6
    n = int(np.ceil(win_len * k["FS"]))
    if n % 2 != 0:
        n += 1
    t_start = 2 * k["R0"] / k["C"] - win_len / 2
    return n, t_start, t_start + np.arange(n) / k["FS"]


def run_physics_spotlight(base_targets, t_vec, pos_sat, vel_sat, heading_deg, speed, l_ant, k=None, chunk=64):
    """sar_batch_sim.py:85-169 -> (raw [n_pulses x num_samples] complex128, t_start, num_samples, v_tgt)."""
    k = k or batch_constants()
    C, FC, T_P, K_RATE, Lambda = k["C"], k["FC"], k["T_P"], k["K_RATE"], k["Lambda"]
    n, t_start, tf = spotlight_window(k)
    phi = np.radians(heading_deg)
    v_tgt = np.array([speed * np.cos(phi), speed * np.sin(phi), 0])
    c, s = np.cos(phi), np.sin(phi)
    rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    p0 = np.array([rot @ np.asarray(t["position"], dtype=np.float64) for t in base_targets])
    rcs = np.array([t["rcs"] for t in base_targets], dtype=np.float64)
    t_vec = np.asarray(t_vec, dtype=np.float64)
    pos_sat = np.asarray(pos_sat, dtype=np.float64)
    vel_sat = np.asarray(vel_sat, dtype=np.float64)
    raw = np.zeros((t_vec.size, n), dtype=np.complex128)
    for i in range(0, t_vec.size, chunk):
        j = min(i + chunk, t_vec.size)
        t = t_vec[i:j, None, None]
        ps = pos_sat[i:j, None, :]
        vs = vel_sat[i:j, None, :]
        p_tgt = p0[None, :, :] + v_tgt[None, None, :] * t
        d_tx_v = p_tgt - ps
        d_tx = np.linalg.norm(d_tx_v, axis=2)
        tau_a = 2 * d_tx / C
        p_rx = ps + vs * tau_a[:, :, None]
        d_rx = np.linalg.norm(p_tgt - p_rx, axis=2)
        tau = (d_tx + d_rx) / C
        b = -ps                                              # p_center = origin (:105)
        look = b / np.linalg.norm(b, axis=2, keepdims=True)
        tgt = d_tx_v / d_tx[:, :, None]
        ang = np.arccos(np.clip(np.sum(look * tgt, axis=2), -1, 1))
        x = np.pi * l_ant * np.sin(ang) / Lambda
        gain = np.ones_like(x)
        m = np.abs(x) > 1e-6
        gain[m] = (np.sin(x[m]) / x[m]) ** 2
        t_loc = tf[None, None, :] - tau[:, :, None]
        mask = np.abs(t_loc) <= (T_P / 2)
        ph = np.pi * K_RATE * (t_loc ** 2) - 2 * np.pi * FC * tau[:, :, None]
        raw[i:j] = np.sum((rcs[None, :] * gain)[:, :, None] * np.exp(1j * ph) * mask, axis=1)
    return raw, t_start, n, v_tgt


def range_compress(raw, num_samples, k):
    """Circular correlation with the fftshifted reference chirp (sar_batch_sim.py:180-185)."""
    n_ref = int(k["T_P"] * k["FS"])
    t_ref = np.linspace(-k["T_P"] / 2, k["T_P"] / 2, n_ref)
    ref = np.exp(1j * np.pi * k["K_RATE"] * t_ref ** 2)
    ref_f = np.fft.fft(np.fft.fftshift(ref), n=num_samples)
    return np.fft.ifft(np.fft.fft(np.asarray(raw), n=num_samples, axis=1) * np.conj(ref_f)[None, :], axis=1)


def _grid_sample_1d(sig32, idx_norm, fused=True):
    """F.grid_sample(bilinear, zeros, align_corners=False) for H = 1, y = 0, as torch's CPU kernel computes it.

    sig32: [P x W] float32 (one plane); idx_norm: [P x B] float64 normalised coordinate, cast to float32 first.
    """
    p, w_ = sig32.shape
    xn = idx_norm.astype(np.float32)
    if fused:
        x = ((xn + np.float32(1)).astype(np.float64) * (w_ / 2) - 0.5).astype(np.float32)
    else:
        x = (xn + np.float32(1)) * np.float32(w_ / 2) - np.float32(0.5)
    x0 = np.floor(x)
    wgt = x - x0
    e = np.float32(1) - wgt
    i0 = x0.astype(np.int64)
    i1 = i0 + 1
    rows = np.arange(p)[:, None]
    v0 = np.where((i0 >= 0) & (i0 < w_), sig32[rows, np.clip(i0, 0, w_ - 1)], np.float32(0))
    v1 = np.where((i1 >= 0) & (i1 < w_), sig32[rows, np.clip(i1, 0, w_ - 1)], np.float32(0))
    return (v0 * e + v1 * wgt).astype(np.float32)


def tdbp(raw, pos_plat, vel_plat, t_start, num_samples, vel_focus, t_pulses, scene_size, nx=512, ny=512, k=None,
         batch=2048, fused=True, rc_data=None):
    """sar_batch_sim.py:171-238 -> complex128 [ny x nx]."""
    k = k or batch_constants()
    C, FC, FS, K_RATE = k["C"], k["FC"], k["FS"], k["K_RATE"]
    x_axis = np.linspace(-scene_size / 2, scene_size / 2, nx)
    y_axis = np.linspace(-scene_size / 2, scene_size / 2, ny)
    pos = np.asarray(pos_plat, dtype=np.float64)
    vel = np.asarray(vel_plat, dtype=np.float64)
    rc = range_compress(raw, num_samples, k) if rc_data is None else rc_data
    re32 = np.ascontiguousarray(rc.real.astype(np.float32))
    im32 = np.ascontiguousarray(rc.imag.astype(np.float32))
    gx, gy = np.meshgrid(x_axis, y_axis, indexing="xy")
    grid = np.stack((gx.ravel(), gy.ravel(), np.zeros(gx.size)), axis=1)
    n_pix = grid.shape[0]
    out = np.zeros(n_pix, dtype=np.complex128)
    v_f = np.asarray(vel_focus, dtype=np.float64).reshape(1, 1, 3)
    t_p = np.asarray(t_pulses, dtype=np.float64).reshape(-1, 1, 1)
    dt = t_p - np.mean(t_p)
    for b0 in range(0, n_pix, batch):
        b1 = min(b0 + batch, n_pix)
        g = grid[None, b0:b1, :] + v_f * dt                                   # :205
        d_tx_v = g - pos[:, None, :]
        d_tx = np.linalg.norm(d_tx_v, axis=2)
        r_unit = d_tx_v / d_tx[:, :, None]
        v_rel = vel[:, None, :] - v_f
        v_rad = np.sum(v_rel * r_unit, axis=2)
        t_shift = (-FC * (2 * v_rad / C)) / K_RATE                            # :213
        tau_a = 2 * d_tx / C
        p_rx = pos[:, None, :] + vel[:, None, :] * tau_a[:, :, None]
        g_rx = g + v_f * tau_a[:, :, None]
        d_rx = np.linalg.norm(g_rx - p_rx, axis=2)
        tau = (d_tx + d_rx) / C
        idx_f = (tau - t_start + t_shift) * FS
        idx_norm = 2 * (idx_f / num_samples) - 1
        s_re = _grid_sample_1d(re32, idx_norm, fused).astype(np.float64)
        s_im = _grid_sample_1d(im32, idx_norm, fused).astype(np.float64)
        out[b0:b1] = np.sum((s_re + 1j * s_im) * np.exp(1j * (2 * np.pi * FC * tau)), axis=0)
    return out.reshape(ny, nx)


def tdbp_scene(n_pulses=96, seed=0, k=None, speed=15.0, heading_deg=30.0, swath=400.0, n_targets=6):
    """Seeded small spotlight CPI: targets, orbit arc, raw echo (oracle), and the tdbp arguments."""
    k = k or scaled_constants()
    rng = np.random.default_rng(seed)
    tg = [{"position": np.array([rng.uniform(-0.3, 0.3) * swath, rng.uniform(-0.3, 0.3) * swath, rng.uniform(0, 10.0)]),
           "rcs": float(rng.uniform(1.0, 50.0))} for _ in range(n_targets)]
    t_vec = (np.arange(n_pulses) - n_pulses / 2) / k["PRF"]
    pos, vel = orbit_arc(t_vec, k)
    l_ant = k["Lambda"] * k["R0"] / swath
    raw, t_start, n, v_tgt = run_physics_spotlight(tg, t_vec, pos, vel, heading_deg, speed, l_ant, k)
    return dict(targets=tg, t_vec=t_vec, pos=pos, vel=vel, l_ant=l_ant, raw=raw, t_start=t_start, num_samples=n,
                v_tgt=v_tgt, swath=swath, k=k, heading_deg=heading_deg, speed=speed)


CONST_NAMES = ("C", "FC", "FS", "T_P", "K_RATE", "R0", "Lambda", "PRF", "BW")


def constants_from_fixture(vec):
    """The ``consts`` vector of tests/golden/spot_*.npz / tdbp_*.npz back into the dict the functions take."""
    k = batch_constants()
    k.update({n: float(v) for n, v in zip(CONST_NAMES, vec)})
    return k
