"""Raw-echo generators with the reference's names and argument order.

The reference keeps its radar constants as module globals; here they are keyword
arguments that default to the same values (sarx.radar.reference_constants()).
Geometry per pulse and target (ranges, delays, carrier phase) is NumPy fp64 on the
host exactly as the reference computes it; the n_pulses x n_targets x n_samples
sample loop - the reference's real wall-clock sink - runs in one HIP kernel.
Returns complex64 (the reference returns complex128; the focuser computes in complex64).
"""
from __future__ import annotations

import numpy as np

from . import radar
from .engine import default_context

_PULSE_CHUNK = 4096          # bounds the [pulses x targets] fp64 table sent per launch


def _synth(ctx, tau, pb_rev, amp, t_fast_abs, kr, t_p):
    n_pulses, n_tgt = tau.shape
    n_samp = t_fast_abs.size
    d_amp = ctx.to_device(np.ascontiguousarray(amp, dtype=np.float32))
    d_tf = ctx.to_device(np.ascontiguousarray(t_fast_abs, dtype=np.float64))
    out = np.empty((n_pulses, n_samp), dtype=np.complex64)
    step = max(1, min(_PULSE_CHUNK, (256 << 20) // max(16 * n_tgt, 1)))
    for i0 in range(0, n_pulses, step):
        i1 = min(i0 + step, n_pulses)
        tp = np.empty((i1 - i0, n_tgt, 2), dtype=np.float64)
        tp[..., 0] = tau[i0:i1]
        tp[..., 1] = pb_rev[i0:i1]
        d_tp = ctx.to_device(tp)
        d_raw = ctx.alloc((i1 - i0) * n_samp * 8)
        ctx.echo_synth(d_tp, d_amp, d_tf, i1 - i0, n_tgt, n_samp, kr, t_p, d_raw)
        out[i0:i1] = d_raw.download(np.complex64, (i1 - i0, n_samp))
        d_tp.release()
        d_raw.release()
    d_amp.release()
    d_tf.release()
    return out


def run_physics_engine(targets, pos_sat, t_vec, *, BW=None, T_p=None, R0=None, C=None, FC=None, fs=600e6,
                       window_sec=22e-6, ctx=None):
    """Monostatic echo; drop-in for sar_satellite_sim.py:211-305.
    returns (raw [len(t_vec) x N_samples] complex64, t_start_fast, fs)"""
    k = radar.reference_constants()
    BW, T_p, R0 = (BW or k["BW"]), (T_p or k["T_p"]), (R0 or k["R0"])
    C, FC = (C or k["C"]), (FC or k["FC"])
    ctx = ctx or default_context()
    num_samples = int(window_sec * fs)                              # :247
    t_start_fast = (2 * R0 / C) - (T_p / 2) - 1e-6                  # :251
    t_fast_abs = t_start_fast + np.linspace(0, num_samples / fs, num_samples)   # :254-255
    k_rate = BW / T_p
    t_pos = np.array([t["position"] for t in targets], dtype=np.float64)
    amp = np.sqrt(np.array([t["rcs"] for t in targets], dtype=np.float64))
    pos_sat = np.asarray(pos_sat, dtype=np.float64)[: len(t_vec)]
    diff = t_pos[None, :, :] - pos_sat[:, None, :]
    dist = np.sqrt(np.sum(diff ** 2, axis=2))                       # :268-269
    tau = 2 * dist / C                                              # :271
    pb_rev = -2.0 * FC * dist / C                                   # :272 (-4 pi FC d / C) / (2 pi)
    return _synth(ctx, tau, pb_rev, amp, t_fast_abs, k_rate, T_p), t_start_fast, fs


def run_bistatic_physics_gpu(targets, t_vec, pos_tx_np, vel_tx_np, rx_offset_dist, vel_target_np, *, FS=None,
                             BW=None, T_p=None, R0=None, C=None, FC=None, window_sec=22e-6, ctx=None):
    """Bistatic (one Tx, offset Rx) echo of moving point targets; drop-in for
    sar_ati_dcpa_sim_csa.py:106-181.  returns (raw complex64, t_start_fast)"""
    k = radar.reference_constants()
    FS, BW, T_p, R0 = (FS or k["FS"]), (BW or k["BW"]), (T_p or k["T_p"]), (R0 or k["R0"])
    C, FC = (C or k["C"]), (FC or k["FC"])
    ctx = ctx or default_context()
    num_samples = int(window_sec * FS)                              # :111
    t_start_fast = (2 * R0 / C) - (T_p / 2) - 1e-6                  # :112
    t_fast_abs = t_start_fast + np.linspace(0, num_samples / FS, num_samples)   # :113-114
    k_rate = BW / T_p
    t_vec = np.asarray(t_vec, dtype=np.float64)
    p0 = np.array([t["position"] for t in targets], dtype=np.float64)
    amp = np.sqrt(np.array([t["rcs"] for t in targets], dtype=np.float64))
    p_tx = np.asarray(pos_tx_np, dtype=np.float64)
    v_tx = np.asarray(vel_tx_np, dtype=np.float64)
    v_dir = v_tx / np.linalg.norm(v_tx, axis=1, keepdims=True)      # :145
    p_rx = p_tx + v_dir * rx_offset_dist                             # :148
    p_now = p0[None, :, :] + np.asarray(vel_target_np, dtype=np.float64)[None, None, :] * t_vec[:, None, None]   # :151
    d_tx = np.linalg.norm(p_now - p_tx[:, None, :], axis=2)         # :156
    d_rx = np.linalg.norm(p_now - p_rx[:, None, :], axis=2)         # :157
    tau = (d_tx + d_rx) / C                                         # :159
    pb_rev = -FC * tau                                              # :160 (-2 pi FC tau) / (2 pi)
    return _synth(ctx, tau, pb_rev, amp, t_fast_abs, k_rate, T_p), t_start_fast
