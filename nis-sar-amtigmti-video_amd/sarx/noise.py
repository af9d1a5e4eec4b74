"""Radar-equation SNR and the ocean noise model of the reference scripts:
``calculate_snr_db`` / ``add_ocean_noise`` (sar_satellite_sim.py:319-344) and the in-place device form of
``generate_noise_tensor`` (sar_batch_sim.py:66-82).  The reference draws from unseeded global generators; here
every sample is a counter-based function of (seed, index), generated on the GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._ffi import check
from .engine import DeviceArray, DeviceBuffer, default_context

# sar_satellite_sim.py:307-317
P_TX, ANT_LENGTH, ANT_WIDTH, T_SYS, NF_DB, LOSS_DB, K_BOLTZ, SCR_DB, K_NU = 1000.0, 3.5, 0.5, 290.0, 5.0, 3.0, 1.380649e-23, 10.0, 1.0


def calculate_snr_db(r_slant, rcs, wavelength, bandwidth, t_int, p_tx=P_TX, ant_l=ANT_LENGTH, ant_w=ANT_WIDTH,
                     t_sys=T_SYS, nf_db=NF_DB, loss_db=LOSS_DB):
    """(snr_db, gain_db) of sar_satellite_sim.py:319-329."""
    ant_area = ant_l * ant_w * 0.6
    gain = 4 * np.pi * ant_area / (wavelength ** 2)
    gain_db = 10 * np.log10(gain)
    numerator = p_tx * (gain ** 2) * (wavelength ** 2) * rcs * t_int
    denominator = ((4 * np.pi) ** 3) * (r_slant ** 4) * K_BOLTZ * t_sys * bandwidth * (10 ** (loss_db / 10)) * (10 ** (nf_db / 10))
    return 10 * np.log10(numerator / denominator), gain_db


def power_stats(d_buf, n, ctx=None):
    """(max |x|^2, mean |x|^2) of a complex64 DeviceBuffer holding n samples."""
    ctx = ctx or d_buf.ctx
    mx, mean = C.c_double(), C.c_double()
    check(ctx.lib.sarx_power_stats_dev(ctx.h, d_buf.ptr, int(n), C.byref(mx), C.byref(mean)), ctx.h)
    return mx.value, mean.value


def add_noise_dev(d_buf, n, ref_power, snr_db, scr_db=SCR_DB, k_nu=K_NU, seed=0, ctx=None):
    """x += thermal + K-clutter in place on the GPU; powers relative to ``ref_power`` exactly as
    generate_noise_tensor (sar_batch_sim.py:67-78) and add_ocean_noise (sar_satellite_sim.py:334-343) set them.
    ``scr_db=None`` adds thermal noise only."""
    ctx = ctx or d_buf.ctx
    noise_std = np.sqrt(ref_power / (10 ** (snr_db / 10)) / 2)
    clutter_power = 0.0 if scr_db is None else ref_power / (10 ** (scr_db / 10))
    check(ctx.lib.sarx_add_ocean_noise_dev(ctx.h, d_buf.ptr, int(n), float(noise_std), float(clutter_power), float(k_nu),
                                           int(seed) & 0xFFFFFFFFFFFFFFFF), ctx.h)


def add_noise_rel_dev(d_buf, n, snr_db, scr_db=SCR_DB, k_nu=K_NU, seed=0, ref="max", ctx=None):
    """power_stats + add_noise_dev in one call that never visits the host: the levels are taken on the device relative to the
    buffer's own max |x|^2 (ref="max": sar_batch_sim.py:313-314) or mean |x|^2 (ref="mean": sar_satellite_sim.py:333), by the same
    partial sums and arithmetic as the two-call form - the samples are bit-identical - and the call only enqueues, so a frame loop
    can keep frames in flight (sarx_add_ocean_noise_rel_dev).  ``scr_db=None`` adds thermal noise only."""
    if ref not in ("max", "mean"):
        raise ValueError("ref must be 'max' or 'mean'")
    ctx = ctx or d_buf.ctx
    snr_lin = 10 ** (snr_db / 10)
    scr_lin = 0.0 if scr_db is None else 10 ** (scr_db / 10)
    check(ctx.lib.sarx_add_ocean_noise_rel_dev(ctx.h, d_buf.ptr, int(n), 1 if ref == "max" else 0, float(snr_lin), float(scr_lin),
                                               float(k_nu), int(seed) & 0xFFFFFFFFFFFFFFFF), ctx.h)


def add_ocean_noise(raw_data, snr_db, scr_db=SCR_DB, k_nu=K_NU, *, seed=0, ctx=None):
    """Drop-in for sar_satellite_sim.py:331-344 (noise relative to the MEAN signal power).  A NumPy array is
    returned as a new complex64 array; a DeviceBuffer is modified in place (pass its sample count as
    ``raw_data.nbytes // 8``) and returned."""
    if isinstance(raw_data, DeviceArray):                           # in place on the GPU, returned
        n = raw_data.shape[0] * raw_data.shape[1]
        _, mean = power_stats(raw_data, n, ctx or raw_data.ctx)
        add_noise_dev(raw_data, n, mean, snr_db, scr_db, k_nu, seed, ctx or raw_data.ctx)
        return raw_data
    if isinstance(raw_data, DeviceBuffer):
        n = raw_data.nbytes // 8
        _, mean = power_stats(raw_data, n, ctx)
        add_noise_dev(raw_data, n, mean, snr_db, scr_db, k_nu, seed, ctx)
        return raw_data
    ctx = ctx or default_context()
    x = np.ascontiguousarray(raw_data, dtype=np.complex64)
    d = ctx.to_device(x)
    _, mean = power_stats(d, x.size, ctx)
    add_noise_dev(d, x.size, mean, snr_db, scr_db, k_nu, seed, ctx)
    out = d.download(np.complex64, x.shape)
    d.release()
    return out
