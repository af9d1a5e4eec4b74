"""Raw-echo generators with the reference's names and argument order.

The reference keeps its radar constants as module globals; here they are keyword
arguments that default to the same values (sarx.radar.reference_constants()).
Geometry per pulse and target (ranges, delays, carrier phase: fp64, the reference's
formulas) and the n_pulses x n_targets x n_samples sample loop - the reference's
real wall-clock sink - both run in HIP kernels; the host only uploads targets and
the per-pulse platform state.
Returns complex64 (the reference returns complex128; the focuser computes in complex64).
"""
from __future__ import annotations

import numpy as np

from . import radar
from ._ffi import check
from .engine import DeviceArray, default_context

_PULSE_CHUNK = 4096          # bounds the [pulses x targets] fp64 table held on the device per launch


def synth_device(ctx, model, tgt_pos, tgt_vel, t_pulse, tx_pos, aux, amp_or_rcs, t_fast_abs, kr, t_p, c_light, fc,
                 l_ant=0.0, wavelength=0.0, out=None, accumulate=False, sync=True):
    """Geometry kernel + sample kernel, pulse chunk by pulse chunk, entirely on the device.
    model 0/1: amp_or_rcs = sqrt(rcs) per target; model 2 (spotlight): rcs per target (the gain is per pulse).
    Returns the DeviceBuffer holding raw [n_pulses x n_samples] complex64 (``out`` if given; ``accumulate`` adds to it).
    ``sync=False`` only enqueues (a frame loop with frames in flight); the tables live in per-lane scratch either way."""
    lib = ctx.lib
    n_pulses, n_tgt, n_samp = tx_pos.shape[0], tgt_pos.shape[0], t_fast_abs.size
    # the small tables live in per-context scratch buffers: a frame loop (sar_batch_sim.py:303-331) calls this once per frame, and
    # ten hipMalloc / hipFree pairs per frame cost more than the tables' upload
    up = lambda tag, a, dt: ctx.scratch_upload("echo." + tag, np.ascontiguousarray(a, dtype=dt))
    d_tp = up("tp", tgt_pos, np.float64)
    d_tv = up("tv", tgt_vel, np.float64) if tgt_vel is not None else None
    d_t = up("t", t_pulse, np.float64) if t_pulse is not None else None
    d_tx = up("tx", tx_pos, np.float64)
    d_aux = up("aux", aux, np.float64) if aux is not None else None
    d_tf = up("tf", t_fast_abs, np.float64)
    if model == 2:
        d_rcs = up("rcs", amp_or_rcs, np.float64)
        d_amp = None
    else:
        d_rcs = None
        d_amp = up("amp", amp_or_rcs, np.float32)
    d_raw = out if out is not None else ctx.alloc(n_pulses * n_samp * 8)
    if d_raw.nbytes < n_pulses * n_samp * 8:
        raise ValueError("out buffer too small")
    step = max(1, min(_PULSE_CHUNK, n_pulses, (512 << 20) // max(20 * n_tgt, 1)))
    d_tab = ctx.scratch("echo.tab", step * n_tgt * 16)
    d_apt = ctx.scratch("echo.apt", step * n_tgt * 4) if model == 2 else None
    ptr = lambda b, off=0: (b.ptr + off) if b is not None else None
    for i0 in range(0, n_pulses, step):
        n = min(step, n_pulses - i0)
        check(lib.sarx_echo_geometry_dev(ctx.h, model, n, n_tgt, d_tp.ptr, ptr(d_tv), ptr(d_t, i0 * 8), d_tx.ptr + i0 * 24,
                                         ptr(d_aux, i0 * 24), ptr(d_rcs), float(c_light), float(fc), float(l_ant),
                                         float(wavelength), d_tab.ptr, ptr(d_apt)), ctx.h)
        dst = d_raw.ptr + i0 * n_samp * 8
        if model == 2:
            check(lib.sarx_echo_spotlight_dev(ctx.h, d_tab.ptr, d_apt.ptr, d_tf.ptr, n, n_tgt, n_samp, float(kr), float(t_p), dst), ctx.h)
        else:
            check(lib.sarx_echo_synth_dev(ctx.h, d_tab.ptr, d_amp.ptr, d_tf.ptr, n, n_tgt, n_samp, float(kr), float(t_p), dst,
                                          1 if accumulate else 0), ctx.h)
    if sync:
        ctx.sync()
    return d_raw


def _download(d_raw, shape):
    out = d_raw.download(np.complex64, shape)
    d_raw.release()
    return out


def run_physics_engine(targets, pos_sat, t_vec, *, BW=None, T_p=None, R0=None, C=None, FC=None, fs=600e6,
                       window_sec=22e-6, ctx=None, device=False):
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
    # ranges, delays tau = 2 d / C and carrier phase -4 pi FC d / C (:268-272) per pulse and target: geometry kernel
    d_raw = synth_device(ctx, 0, t_pos, None, None, pos_sat, None, amp, t_fast_abs, k_rate, T_p, C, FC)
    if device:                                                      # stays on the GPU for add_ocean_noise / sar_focus_rda
        return DeviceArray(d_raw, (pos_sat.shape[0], num_samples)), t_start_fast, fs
    return _download(d_raw, (pos_sat.shape[0], num_samples)), t_start_fast, fs


def run_moving_physics(targets, t_vec, pos_sat, vel_target, *, BW=None, T_p=None, R0=None, C=None, FC=None, ctx=None,
                       device=False):
    """Monostatic echo of targets moving at ``vel_target``; drop-in for sar_satellite_moving_sim.py:111-159.
    returns (raw [len(t_vec) x 13200] complex64, t_start_fast, fs); ``device=True``: raw stays on the GPU as a DeviceArray"""
    k = radar.reference_constants()
    BW, T_p, R0 = (BW or k["BW"]), (T_p or k["T_p"]), (R0 or k["R0"])
    C, FC = (C or k["C"]), (FC or k["FC"])
    ctx = ctx or default_context()
    fs = 600e6                                                      # :114
    num_samples = int(22e-6 * fs)
    t_start_fast = (2 * R0 / C) - (T_p / 2) - 1e-6                  # :116
    t_fast_abs = t_start_fast + np.linspace(0, num_samples / fs, num_samples)   # :117-118
    t_vec = np.asarray(t_vec, dtype=np.float64)
    t_pos = np.array([t["position"] for t in targets], dtype=np.float64)
    amp = np.sqrt(np.array([t["rcs"] for t in targets], dtype=np.float64))
    pos_sat = np.asarray(pos_sat, dtype=np.float64)[: t_vec.size]
    # P(t) = P0 + V t (:137), tau = 2 d / C, phase -4 pi FC d / C (:143-144): geometry kernel
    d_raw = synth_device(ctx, 0, t_pos, np.asarray(vel_target, dtype=np.float64), t_vec, pos_sat, None, amp, t_fast_abs, BW / T_p,
                         T_p, C, FC)
    if device:
        return DeviceArray(d_raw, (t_vec.size, num_samples)), t_start_fast, fs
    return _download(d_raw, (t_vec.size, num_samples)), t_start_fast, fs


def run_custom_physics(targets, t_vec, pos, tuned_prp, t_p, fc, bw, *, R0=None, C=None, ctx=None, device=False):
    """The vehicle script's monostatic echo (2048 samples at 360 MHz); drop-in for sar_vehicle_sim.py:83-128
    (``tuned_prp`` is accepted and unused, as in the reference; R0 and C are its module globals).  returns raw complex64;
    ``device=True``: raw stays on the GPU as a DeviceArray"""
    k = radar.reference_constants()
    R0, C = (R0 or k["R0"]), (C or k["C"])
    ctx = ctx or default_context()
    fs, num_samples = 360e6, 2048                                   # :85-86
    t_start_fast = (2 * R0 / C) - (num_samples / fs) / 2            # :89
    t_fast_abs = t_start_fast + np.linspace(0, num_samples / fs, num_samples)   # :88,100
    t_pos = np.array([t["position"] for t in targets], dtype=np.float64)
    amp = np.sqrt(np.array([t["rcs"] for t in targets], dtype=np.float64))
    pos = np.asarray(pos, dtype=np.float64)[: len(t_vec)]
    d_raw = synth_device(ctx, 0, t_pos, None, None, pos, None, amp, t_fast_abs, bw / t_p, t_p, C, fc)
    if device:
        return DeviceArray(d_raw, (pos.shape[0], num_samples))
    return _download(d_raw, (pos.shape[0], num_samples))


def run_bistatic_physics_gpu(targets, t_vec, pos_tx_np, vel_tx_np, rx_offset_dist, vel_target_np, *, FS=None,
                             BW=None, T_p=None, R0=None, C=None, FC=None, window_sec=22e-6, ctx=None, device=False,
                             add_to=None):
    """Bistatic (one Tx, offset Rx) echo of moving point targets; drop-in for
    sar_ati_dcpa_sim_csa.py:106-181.  returns (raw complex64, t_start_fast).
    ``device=True`` returns a DeviceArray instead of a NumPy array; ``add_to`` (a DeviceArray of the same shape)
    receives this call's echoes on top of what it holds - the reference's ``raw += clutter`` (:193-196) - and is
    returned."""
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
    # target motion p0 + v t (:151), d_tx, d_rx (:156-157), tau = (d_tx + d_rx) / C, phase -2 pi FC tau (:159-160): geometry kernel
    if add_to is not None:
        if add_to.shape != (t_vec.size, num_samples) or add_to.offset:
            raise ValueError("add_to must be a whole DeviceArray of this call's shape")
        synth_device(ctx, 1, p0, np.asarray(vel_target_np, dtype=np.float64), t_vec, p_tx, p_rx, amp, t_fast_abs, k_rate, T_p,
                     C, FC, out=add_to.buf, accumulate=True)
        return add_to, t_start_fast
    d_raw = synth_device(ctx, 1, p0, np.asarray(vel_target_np, dtype=np.float64), t_vec, p_tx, p_rx, amp, t_fast_abs, k_rate, T_p,
                         C, FC)
    if device:
        return DeviceArray(d_raw, (t_vec.size, num_samples)), t_start_fast
    return _download(d_raw, (t_vec.size, num_samples)), t_start_fast
