"""Host-side mirror of the reference's functions for the CSA / ATI-DPCA path.

Same names, argument order and return tuples as the reference scripts, so a
caller switches by changing an import.  All arithmetic happens in libsarx's
HIP kernels; this module only validates, stages buffers and shapes results.
"""
from __future__ import annotations

import numpy as np

from . import _ffi
from .engine import CsaPlan, DeviceArray, default_context

_plan_cache = {}
_PLAN_CACHE_MAX = 4


def _get_plan(ctx, n_az, n_rg, args, flags):
    key = (ctx.device_id, id(ctx), n_az, n_rg, tuple(float(a) for a in args), flags)
    plan = _plan_cache.get(key)
    if plan is None or plan.h is None:
        while len(_plan_cache) >= _PLAN_CACHE_MAX:       # plans hold full-image scratch
            _plan_cache.pop(next(iter(_plan_cache))).close()
        plan = _plan_cache[key] = CsaPlan(ctx, n_az, n_rg, *args, flags=flags)
    return plan


_host_ws = {}          # focus_ati_dpca on host arrays: the device buffers of the last size, kept from call to call


def _host_workspace(ctx, n_az, n_rg):
    """Device buffers for focus_ati_dpca's host-array path (both echoes, both images, the planes), kept for the LAST size only: a frame
    loop then pays neither hipMalloc / hipFree (the latter waits for the device) nor the first touch of fresh device pages (an
    upload into a new allocation ran 24-29 ms against 11.5 ms into a used one at 8192^2) per call.  clear_plan_cache() frees it."""
    key = (id(ctx), int(n_az), int(n_rg))
    ent = _host_ws.get(key)
    if ent is None or ctx.h is None or any(b.ptr is None for b in ent["all"]):
        _release_host_ws()
        ws = two_channel_workspace(ctx, n_az, n_rg)
        raw = [ctx.alloc(n_az * n_rg * 8), ctx.alloc(n_az * n_rg * 8)]
        ent = _host_ws[key] = {"ws": ws, "raw": raw, "all": raw + [v for k, v in ws.items() if k != "shape"]}
    return ent


def _release_host_ws():
    for ent in _host_ws.values():
        for b in ent["all"]:
            try:
                b.release()
            except Exception:
                pass
    _host_ws.clear()


def clear_plan_cache():
    _release_host_ws()
    for p in _plan_cache.values():
        p.close()
    _plan_cache.clear()


def sar_focus_csa(phist, center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz, prf_hz,
                  platform_speed_mps, range_ref_m, t_start_fast, *, ctx=None, fuse_range=True,
                  materialize_transpose=False, out=None):
    """Chirp Scaling focus on the GPU; drop-in for the reference function of the
    same name (sar_ati_dcpa_sim_csa.py:202-396).

    phist : [N_pulses x N_samples] complex (any complex/float dtype; computed as complex64), or the DeviceArray
            an echo generator returned with ``device=True``
    returns (img.T [N_rg x N_az] complex64, range_axis [N_rg], cross_range_axis [N_az])

    Like the reference, the image comes back as the transpose *view* of an
    [N_az x N_rg] row-major array (:396 returns ``img.T``);
    ``materialize_transpose=True`` corner-turns on the GPU instead and returns a
    C-contiguous [N_rg x N_az] array.  ``pulse_width_sec`` is accepted and
    unused, exactly as in the reference.

    The image array is allocated per call as in the reference; for results of
    64 MiB and more it sits on page-locked memory from a small per-context pool
    (Context.pinned_empty), so the download is one DMA and a second call pays no
    first touch.  ``out`` = the image a previous call returned (or any complex64
    array of that shape and layout) is overwritten and returned instead.
    """
    on_device = isinstance(phist, DeviceArray)          # echoes synthesised with device=True: nothing is uploaded
    if on_device:
        if phist.transposed:
            raise ValueError("a device phist must be the [N_pulses x N_samples] array itself, not its .T")
        a = phist
        ctx = ctx or phist.ctx
    else:
        a = np.asarray(phist)
        if a.ndim != 2:
            raise ValueError("phist must be 2-D [N_pulses x N_samples]")
        ctx = ctx or default_context()
    n_az, n_rg = a.shape
    flags = (_ffi.FUSE_RANGE if fuse_range else 0) | (_ffi.OUT_RG_MAJOR if materialize_transpose else 0)
    args = (center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz, prf_hz,
            platform_speed_mps, range_ref_m, t_start_fast)
    plan = _get_plan(ctx, n_az, n_rg, args, flags)
    if on_device:
        d_img = plan.result_buffer()                 # the plan's own, reused from call to call (the download below is blocking)
        plan.focus_dev(a, d_img)
        img = d_img.download(np.complex64, (n_rg, n_az) if materialize_transpose else (n_az, n_rg))
    else:
        dst = None
        if out is not None:
            dst = out if materialize_transpose else out.T      # the [n_az x n_rg] memory behind the img.T view
        img = plan.focus_host(a, out=dst)      # complex128 input is narrowed inside the library while it is staged
    range_axis, cross_range_axis = plan.axes()
    return (img if materialize_transpose else img.T), range_axis, cross_range_axis


class FocusFuture:
    """Result of sar_focus_csa_async: result() waits for the frame and returns what sar_focus_csa returns."""

    def __init__(self, plan, ticket, img, transposed_view):
        self._plan, self._ticket, self._img, self._view = plan, ticket, img, transposed_view
        self._axes = None

    def done(self):
        return self._ticket is None

    def __del__(self):                          # dropped without result(): the frame leaves the plan's two-deep pipeline all the same
        try:
            if self._ticket is not None and self._plan.h is not None:
                self._plan.focus_host_end(self._ticket)
        except Exception:
            pass

    def result(self):
        if self._ticket is not None:
            ticket, self._ticket = self._ticket, None
            self._plan.focus_host_end(ticket)
        if self._axes is None:
            self._axes = self._plan.axes()
        return (self._img.T if self._view else self._img), self._axes[0], self._axes[1]


def sar_focus_csa_async(phist, center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz, prf_hz,
                        platform_speed_mps, range_ref_m, t_start_fast, *, ctx=None, fuse_range=True,
                        materialize_transpose=False, out=None):
    """sar_focus_csa (sar_ati_dcpa_sim_csa.py:202-396) returning a FocusFuture at once: the frame's upload has happened, its focus
    and its download are in flight.  Calling it for the next frame BEFORE result() of this one overlaps that frame's upload with
    this frame's focus and download (PCIe is full duplex): the loop of sar_batch_sim.py:303-331 at ~41-46 ms per 16384^2 frame
    instead of 82-88.  At most two futures per (size, radar) may be pending; `phist` must not be modified before result().
    Same results, bit for bit, as sar_focus_csa."""
    a = np.asarray(phist)
    if a.ndim != 2:
        raise ValueError("phist must be 2-D [N_pulses x N_samples]")
    ctx = ctx or default_context()
    n_az, n_rg = a.shape
    flags = (_ffi.FUSE_RANGE if fuse_range else 0) | (_ffi.OUT_RG_MAJOR if materialize_transpose else 0)
    args = (center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz, prf_hz,
            platform_speed_mps, range_ref_m, t_start_fast)
    plan = _get_plan(ctx, n_az, n_rg, args, flags)
    dst = None
    if out is not None:
        dst = out if materialize_transpose else out.T
    ticket, img = plan.focus_host_begin(a, out=dst)
    return FocusFuture(plan, ticket, img, not materialize_transpose)


def focus_stream(frames, center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz, prf_hz,
                 platform_speed_mps, range_ref_m, t_start_fast, *, ctx=None, reserve=3, **kw):
    """Generator over an iterable of raw frames (host arrays of one shape): yields sar_focus_csa's tuple for every frame, in order,
    with frame i+1 uploading while frame i focuses and downloads (the frame loop of sar_batch_sim.py:303-331 on host arrays).
    `reserve` page-locked result blocks are set aside before the first frame (Context.reserve_pinned), so no call inside the loop
    meets hipHostMalloc; a consumer that keeps more than reserve - 2 results alive at once gets fresh blocks for the rest."""
    ctx = ctx or default_context()
    pending = None
    for raw in frames:
        raw = np.asarray(raw)
        if pending is None and reserve:
            ctx.reserve_pinned(raw.shape, np.complex64, reserve)
        fut = sar_focus_csa_async(raw, center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz, prf_hz,
                                  platform_speed_mps, range_ref_m, t_start_fast, ctx=ctx, **kw)
        if pending is not None:
            yield pending.result()
        pending = fut
    if pending is not None:
        yield pending.result()


def dpca_pulse_shift(raw_rx1, raw_rx2):
    """DPCA co-registration by one pulse (sar_ati_dcpa_sim_csa.py:402-403)."""
    return raw_rx1[1:, :], raw_rx2[:-1, :]


def _as_device_layout(x):
    """complex64 array whose memory can be handed over without a transpose copy.
    Returns (contiguous array, was_transposed)."""
    x = np.asarray(x)
    if x.ndim == 2 and x.flags.f_contiguous and not x.flags.c_contiguous:
        return np.ascontiguousarray(x.T, dtype=np.complex64), True
    return np.ascontiguousarray(x, dtype=np.complex64), False


def ati_dpca(slc1, slc2, mask_frac=0.05, cal_phase=0.0, *, ctx=None, complex_products=False,
             viewer_products=False):
    """ATI interferogram + DPCA difference in one GPU kernel.

    The reference has no function for this: the expressions are inline at
    sar_ati_dcpa_sim_csa.py:414-419 (products), :447-449 (mask) and restated in
    the viewer (sar_ati_dcpa_viewer_csa.py:42-52, which first applies
    ``slc2 * exp(1j*cal_phase)``).  Returns a dict with the reference's
    variable names: ati_phase, slc1_mag, dpca_mag, ati_phase_masked, mask,
    max_mag, sum_interf (= sum(slc1*conj(slc2)), the phase-balance input);
    optionally ati_interf / dpca_diff and the viewer's product names.
    """
    a, ta = _as_device_layout(slc1)
    b, tb = _as_device_layout(slc2)
    if a.shape != b.shape or ta != tb:
        raise ValueError("slc1 and slc2 must have the same shape and memory order")
    ctx = ctx or default_context()
    n = a.size
    d1, d2 = ctx.to_device(a), ctx.to_device(b)
    names = ["ati_phase", "slc1_mag", "dpca_mag"]
    if complex_products:
        names += ["ati_interf", "dpca_diff"]
    if viewer_products:
        names += ["slc2_mag", "slc1_phase", "slc2_phase", "dpca_phase"]
    outs = {k: ctx.alloc(n * (8 if k in ("ati_interf", "dpca_diff") else 4)) for k in names}
    max_mag, sum_interf = ctx.ati_dpca(d1, d2, n, cal_phase, outs)
    thr = np.float32(max_mag) * np.float32(mask_frac)
    d_masked = ctx.alloc(n * 4)
    ctx.mask_phase(outs["ati_phase"], outs["slc1_mag"], n, thr, d_masked)

    def get(buf, dt):
        arr = buf.download(dt, a.shape)
        return arr.T if ta else arr

    res = {k: get(v, np.complex64 if k in ("ati_interf", "dpca_diff") else np.float32) for k, v in outs.items()}
    res["ati_phase_masked"] = get(d_masked, np.float32)
    res["mask"] = res["slc1_mag"] > thr
    res["max_mag"] = max_mag
    res["sum_interf"] = sum_interf
    if viewer_products:     # viewer's dictionary keys (sar_ati_dcpa_viewer_csa.py:44-52)
        res.update({"Ch1 Magnitude": res["slc1_mag"], "Ch1 Phase": res["slc1_phase"],
                    "Ch2 Magnitude": res["slc2_mag"], "Ch2 Phase": res["slc2_phase"],
                    "DPCA Magnitude": res["dpca_mag"], "DPCA Phase": res["dpca_phase"],
                    "ATI Phase": res["ati_phase"]})
    for bufs in (d1, d2, d_masked, *outs.values()):
        bufs.release()
    return res


def phase_balance(slc1, slc2, *, ctx=None):
    """cal_phase = angle(mean(slc1*conj(slc2)))  (sar_ati_dcpa_viewer_csa.py:249-250),
    from the fp64 reduction fused into the ATI kernel."""
    r = ati_dpca(slc1, slc2, ctx=ctx)
    return float(np.angle(r["sum_interf"]))


def two_channel_workspace(ctx, n_az, n_rg):
    """Device buffers for focus_ati_dpca(..., workspace=...): both images, the three product planes and the max slot,
    allocated once and reused by every call of that size (a frame loop then allocates nothing).  release() each when done."""
    n = int(n_az) * int(n_rg)
    ws = {"slc1": ctx.alloc(n * 8), "slc2": ctx.alloc(n * 8), "d_max": ctx.alloc(_ffi.MAX_SLOT_BYTES)}
    for k in ("ati_phase_masked", "slc1_mag", "dpca_mag"):
        ws[k] = ctx.alloc(n * 4)
    ws["shape"] = (int(n_az), int(n_rg))
    return ws


def focus_ati_dpca(raw_rx1, raw_rx2, center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz,
                   prf_hz, platform_speed_mps, range_ref_m, t_start_fast, mask_frac=0.05, cal_phase=0.0, *,
                   ctx=None, pulse_shift=True, return_slc2=True, unmasked_phase=False, device_output=False,
                   workspace=None, fetch_stats=True):
    """The reference script's processing section in one call
    (sar_ati_dcpa_sim_csa.py:402-419,447-449): pulse shift, CSA focus of both
    channels, ATI/DPCA products, 5 % magnitude mask.  Nothing visits the host between the steps; with DeviceArray
    inputs (echo generators called with device=True) nothing is uploaded at all.

    The product stage runs the way bench.py's driver (sarx.batch.TwoChannelBatch) runs it: channel 1's focus leaves
    max|slc1| on the device while it writes the image (sarx_csa_plan_set_max_slot), and channel 2's last azimuth launch
    reads slc1 beside the slc2 samples it holds and writes the masked ATI phase, |slc1| and the DPCA magnitude
    (sarx_csa_plan_set_ati) - no host round trip for the threshold, no ATI or mask launch, neither image read again.
    Sizes without that epilogue (anything but a power of two or 7199 x 13200) run the separate ATI launch with the
    device-side threshold instead; the planes are bit-identical either way.

    Returns a dict with the reference's variable names: slc1, slc2 ([N_rg x N_az] views like sar_focus_csa's),
    slc1_mag, dpca_mag, ati_phase_masked, range_axis, cross_range, max_mag, sum_interf.
    return_slc2=False   : channel 2's image is never written (the reference saves it, :457-461, hence the default)
    unmasked_phase=True : also "ati_phase", the unmasked np.angle(slc1*conj(slc2)) of :415 (one more launch)
    device_output=True  : the images and planes stay on the GPU ([N_az x N_rg] row-major DeviceBuffers; release() them),
                          max_mag / sum_interf are still fetched (24 bytes) unless fetch_stats=False (then the call only
                          enqueues: no host synchronisation at all; Context.ati_stats() fetches them later)
    workspace           : two_channel_workspace(ctx, N_az, N_rg): buffers reused from call to call (with device_output the
                          returned buffers ARE the workspace's - do not release them per call)
    """
    ctx = ctx or default_context()
    on_device = isinstance(raw_rx1, DeviceArray) and isinstance(raw_rx2, DeviceArray)
    if on_device:                      # echoes synthesised on the GPU: the pulse shift is two views, nothing is uploaded
        r1, r2 = (raw_rx1.rows(1, None), raw_rx2.rows(0, -1)) if pulse_shift else (raw_rx1, raw_rx2)
    else:
        r1, r2 = (dpca_pulse_shift(raw_rx1, raw_rx2) if pulse_shift else (raw_rx1, raw_rx2))
        r1 = np.ascontiguousarray(r1, dtype=np.complex64)
        r2 = np.ascontiguousarray(r2, dtype=np.complex64)
    if r1.shape != r2.shape:
        raise ValueError("the two channels must have the same shape")
    n_az, n_rg = r1.shape
    args = (center_wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz, prf_hz,
            platform_speed_mps, range_ref_m, t_start_fast)
    plan = _get_plan(ctx, n_az, n_rg, args, _ffi.FUSE_RANGE)
    n = n_az * n_rg
    host_ent = None
    if workspace is None and not on_device and not device_output:
        host_ent = _host_workspace(ctx, n_az, n_rg)            # host arrays in, host arrays out: buffers kept from the last call of this size
        workspace = host_ent["ws"]
    if workspace is not None:
        if workspace.get("shape") != (n_az, n_rg):
            raise ValueError("workspace was made for another size")
        bufs = {k: v for k, v in workspace.items() if k != "shape"}
        keep = set(bufs)                                    # never released here
    else:
        bufs = {"slc1": ctx.alloc(n * 8), "slc2": ctx.alloc(n * 8), "d_max": ctx.alloc(_ffi.MAX_SLOT_BYTES)}
        for k in ("ati_phase_masked", "slc1_mag", "dpca_mag"):
            bufs[k] = ctx.alloc(n * 4)
        keep = set()
    if host_ent is not None:
        ctx.sync()                                          # the kept buffers are idle (a previous call's downloads have been waited for anyway)
        d_raw, d_raw2 = host_ent["raw"]
    else:
        d_raw = None if on_device else ctx.alloc(n * 8)
        d_raw2 = None if on_device else ctx.alloc(n * 8)   # channel 2 uploads while channel 1 focuses: a buffer of its own
    early = {}                                             # downloads started before the chain is complete (slc1 during channel 2's upload)
    try:
        fused = False
        have_max = False
        try:                                                # one guard around both focuses and both uploads: the cached plan
            try:                                            # never keeps a pointer to a buffer released below
                plan.set_max_slot(bufs["d_max"])            # channel 1: image + max|image| out of the same launch
                have_max = True
            except _ffi.SarxError:
                have_max = False
            if on_device:
                plan.focus_dev(r1, bufs["slc1"])
            else:
                # PCIe is full duplex and neither copy needs the compute units: channel 2 uploads while channel 1 focuses, and slc1
                # (complete when channel 1's focus is) downloads while channel 2 still uploads - sar_ati_dcpa_sim_csa.py:410-411
                d_raw.upload_unordered(r1)                  # idle (fresh, or kept from a finished call): nothing enqueued touches it
                plan.focus_dev(d_raw, bufs["slc1"])
                if not device_output:
                    early["slc1"] = bufs["slc1"].download_begin(np.complex64, (n_az, n_rg))
                d_raw2.upload_unordered(r2)
            src2 = r2 if on_device else d_raw2
            if have_max and not unmasked_phase:
                try:                                        # channel 2: the products come out of its last azimuth launch
                    plan.set_ati(bufs["slc1"], bufs["d_max"], mask_frac, cal_phase, bufs["ati_phase_masked"], bufs["slc1_mag"],
                                 bufs["dpca_mag"], keep_image=return_slc2)
                    fused = True
                except _ffi.SarxError:
                    fused = False
            if have_max and not fused:
                # without the ATI epilogue armed a focus clears the slot and reduces max|its own image| into it: channel 2's
                # focus must not see it, or the 5 % threshold below would come from max|slc2| instead of max|slc1| (:447)
                plan.set_max_slot(None)
            plan.focus_dev(src2, bufs["slc2"])
        finally:
            plan.set_ati(None)
            plan.set_max_slot(None)
        if not fused:
            if unmasked_phase or not have_max:              # ATI launch, then the mask with the threshold taken on the device
                bufs["ati_phase"] = ctx.alloc(n * 4)
                outs = {"ati_phase": bufs["ati_phase"], "slc1_mag": bufs["slc1_mag"], "dpca_mag": bufs["dpca_mag"]}
                ctx.ati_dpca(bufs["slc1"], bufs["slc2"], n, cal_phase, outs, want_stats=False)
                ctx.mask_phase_frac(bufs["ati_phase"], bufs["slc1_mag"], n, mask_frac, bufs["ati_phase_masked"])
            else:                                           # mask inside the ATI launch, threshold from channel 1's focus
                outs = {"ati_phase": bufs["ati_phase_masked"], "slc1_mag": bufs["slc1_mag"], "dpca_mag": bufs["dpca_mag"]}
                ctx.ati_dpca_masked(bufs["slc1"], bufs["slc2"], n, cal_phase, bufs["d_max"], mask_frac, outs)
        max_mag, sum_interf = ctx.ati_stats() if (fetch_stats or not device_output) else (None, None)   # the only host synchronisation of the chain
        ra, ca = plan.axes()
        res = {"range_axis": ra, "cross_range": ca, "max_mag": max_mag, "sum_interf": sum_interf, "fused_products": fused}
        names = ["slc1"] + (["slc2"] if (return_slc2 or not fused) else []) + ["slc1_mag", "dpca_mag", "ati_phase_masked"] + \
                (["ati_phase"] if "ati_phase" in bufs else [])
        if device_output:
            for k in names:
                res[k] = bufs[k]
                keep.add(k)
        else:                                               # every plane's download in flight at once, then collected in order
            for k in names:
                if k not in early:
                    early[k] = bufs[k].download_begin(np.complex64 if k in ("slc1", "slc2") else np.float32, (n_az, n_rg))
            for k in names:
                res[k] = early.pop(k).result().T
        return res
    finally:
        for pend in early.values():                         # an exception on the way: no copy may outlive the buffers released below
            try:
                pend.result()
            except _ffi.SarxError:
                pass
        for k, b in bufs.items():
            if k not in keep:
                b.release()
        if host_ent is None:
            for b in (d_raw, d_raw2):
                if b is not None:
                    b.release()
