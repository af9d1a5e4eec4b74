"""Range-Doppler focuser with the reference's signature:
``sar_focus_rda`` of sar_satellite_sim.py:356-448 (pasted again in sar_satellite_moving_sim.py:208 and
sar_vehicle_sim.py:182)."""
from __future__ import annotations

import ctypes as C

import os

import numpy as np

from . import _ffi
from ._ffi import check
from .engine import DeviceArray, default_context


class RdaPlan:
    """sarx_rda_plan: filter spectrum, window, axes and scratch for one (n_ranges, n_pulses, radar) combination."""

    def __init__(self, ctx, n_r, n_p, prm):
        self.ctx = ctx
        self.h = C.c_void_p()
        check(ctx.lib.sarx_rda_plan_create(ctx.h, n_r, n_p, C.byref(prm), C.byref(self.h)), ctx.h)
        ctx._plans.add(self)          # closed with the context, before sarx_destroy

    def close(self):
        if self.h and self.ctx.h is not None:
            self.ctx.lib.sarx_rda_plan_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_plans = {}


def _plan(ctx, n_r, n_p, prm):
    key = (id(ctx), n_r, n_p, os.environ.get("SARX_RDA_DIRECT")) + tuple(getattr(prm, f) for f, _ in prm._fields_)
    plan = _plans.get(key)
    if plan is None or plan.h is None:
        while len(_plans) >= 2:                       # plans hold several full-image buffers
            _plans.pop(next(iter(_plans))).close()
        plan = _plans[key] = RdaPlan(ctx, n_r, n_p, prm)
    return plan


VARIANTS = ("satellite", "moving", "vehicle")


def sar_focus_rda(phist, center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz, prf_hz,
                  platform_speed_mps, range_grp_m, *, ctx=None, intermediates=True, device_output=False, variant="satellite"):
    """phist: [num_ranges x num_pulses] complex (the scripts pass ``raw_data.T``); a NumPy array, or ``d.T`` of the
    [pulses x ranges] DeviceArray an echo generator returned with ``device=True`` (nothing is uploaded then).

    Returns the reference's 7-tuple (:447-448): (sar_image_mag.T [pulses x ranges], range_axis_centered,
    cross_range_m, phist_compressed, range_doppler, range_doppler_rcmc [ranges x pulses each], doppler_freq).
    Images are float32 / complex64.  ``intermediates=False`` skips downloading the three complex maps
    (they come back as None).  ``device_output=True`` (device input only) leaves the images on the GPU: the tuple then
    holds DeviceBuffers ([pulses x ranges] row-major: float32 magnitude, complex64 maps) and nothing is downloaded.

    ``variant`` selects which of the three pasted copies' return tuple comes back (the arithmetic is the same in all three):
    "satellite" the 7-tuple above; "moving" sar_satellite_moving_sim.py:208-285's 3-tuple (sar_image_mag.T,
    range_axis_centered, cross_range_m) - no intermediate is stored or downloaded; "vehicle" sar_vehicle_sim.py:182-273's
    8-tuple, the 7-tuple with range_doppler_filtered (the map after azimuth compression, :268) in front of doppler_freq.
    """
    if variant not in VARIANTS:
        raise ValueError(f"variant must be one of {VARIANTS}")
    if variant == "moving":
        intermediates = False
    n_maps = 4 if variant == "vehicle" else 3
    on_device = isinstance(phist, DeviceArray)
    if on_device:
        if not phist.transposed:
            raise ValueError("device input must be raw.T of a [pulses x ranges] DeviceArray, as the scripts pass raw_data.T")
        n_r, n_p = phist.shape
        ctx = ctx or phist.ctx
    else:
        a = np.asarray(phist)
        if a.ndim != 2:
            raise ValueError("phist must be 2-D [num_ranges x num_pulses]")
        n_r, n_p = a.shape
        ctx = ctx or default_context()
        # pulse-major memory: free when phist is the usual raw.T view
        x = np.ascontiguousarray(a.T, dtype=np.complex64)
    lib = ctx.lib
    prm = _ffi.RadarParams(center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz, prf_hz,
                           platform_speed_mps, range_grp_m, 0.0)
    plan = _plan(ctx, n_r, n_p, prm)
    if on_device:          # device in, device out: only what the caller wants is downloaded afterwards
        bufs = []
        handed_over = False
        try:
            d_mag = ctx.alloc(n_p * n_r * 4)
            bufs.append(d_mag)
            d_st = []
            for _ in range(n_maps):
                d_st.append(ctx.alloc(n_p * n_r * 8) if intermediates else None)
                bufs.append(d_st[-1])
            ptrs = [b.ptr if b is not None else None for b in d_st] + [None] * (4 - n_maps)
            check(lib.sarx_rda_focus_dev2(plan.h, phist.ptr, d_mag.ptr, *ptrs), ctx.h)
            if device_output:
                r_ax, c_ax, fd = np.empty(n_r), np.empty(n_p), np.empty(n_p)
                check(lib.sarx_rda_axes(plan.h, r_ax.ctypes.data, c_ax.ctypes.data, fd.ctypes.data), ctx.h)
                handed_over = True
                if variant == "moving":
                    return (d_mag, r_ax, c_ax)
                return (d_mag, r_ax, c_ax, *d_st, fd)
            mag = d_mag.download(np.float32, (n_p, n_r))
            stages = [b.download(np.complex64, (n_p, n_r)) if b is not None else None for b in d_st]
        finally:               # a failing focus or download must not keep up to four image-sized buffers
            if not handed_over:
                for b in bufs:
                    if b is not None:
                        b.release()
    else:
        if device_output:
            raise ValueError("device_output needs a device input (DeviceArray .T)")
        mag = ctx.pinned_empty((n_p, n_r), np.float32)             # large results: page-locked pool (one DMA, no first touch)
        stages = [ctx.pinned_empty((n_p, n_r), np.complex64) if intermediates else None for _ in range(n_maps)]
        ptr = [s.ctypes.data if s is not None else None for s in stages] + [None] * (4 - n_maps)
        check(lib.sarx_rda_focus_host2(plan.h, x.ctypes.data, mag.ctypes.data, *ptr), ctx.h)
    r_ax, c_ax, fd = np.empty(n_r), np.empty(n_p), np.empty(n_p)
    check(lib.sarx_rda_axes(plan.h, r_ax.ctypes.data, c_ax.ctypes.data, fd.ctypes.data), ctx.h)
    if variant == "moving":
        return mag, r_ax, c_ax
    maps = tuple(s.T if s is not None else None for s in stages)
    return (mag, r_ax, c_ax, *maps, fd)
