"""Thin object layer over the C ABI: Context (one per GPU), DeviceBuffer, CsaPlan."""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

import numpy as np

from . import _ffi
from ._ffi import SarxError, check


class DeviceBuffer:
    """HBM allocation owned by a Context (freed with it or on release())."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(ctx.lib.sarx_malloc(ctx.h, self.nbytes, C.byref(p)), ctx.h)
        self.ptr = p.value
        ctx._live[id(self)] = self.ptr

    def release(self):
        if self.ptr is not None and self.ctx.h is not None:
            self.ctx.lib.sarx_free(self.ctx.h, self.ptr)
            self.ctx._live.pop(id(self), None)
        self.ptr = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes > self.nbytes:
            raise ValueError("upload larger than buffer")
        check(self.ctx.lib.sarx_memcpy_h2d(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes), self.ctx.h)
        return self

    def upload_lane(self, arr):
        """upload() that waits for the current lane's enqueued work only (sarx_memcpy_h2d_lane): for a buffer only this lane's launches
        read, so that frames in flight on the other lanes keep running (upload() waits for every lane)."""
        arr = np.ascontiguousarray(arr)
        if arr.nbytes > self.nbytes:
            raise ValueError("upload larger than buffer")
        check(self.ctx.lib.sarx_memcpy_h2d_lane(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes), self.ctx.h)
        return self

    def upload_unordered(self, arr):
        """upload() that does not wait for GPU work already enqueued: the caller guarantees nothing enqueued touches this buffer, and
        the copy then overlaps whatever the GPU is doing (sarx_memcpy_h2d_unordered)."""
        arr = np.ascontiguousarray(arr)
        if arr.nbytes > self.nbytes:
            raise ValueError("upload larger than buffer")
        check(self.ctx.lib.sarx_memcpy_h2d_unordered(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes), self.ctx.h)
        return self

    def download_begin(self, dtype, shape, force=False):
        """Asynchronous download, ordered after everything enqueued so far on the current lane: returns a PendingDownload whose
        result() waits for the copy and returns the array.  The copy is asynchronous when the context's pool hands out a page-locked
        block for it - from a size's fourth request on, at once with force=True or after Context.reserve_pinned; otherwise (a one-shot
        script, which should not pay hipHostMalloc for arrays it downloads once) it is the ordinary blocking copy into a NumPy array."""
        out = self.ctx.pinned_empty(shape, dtype, force=force)
        if out.nbytes > self.nbytes:
            raise ValueError("download larger than buffer")
        return PendingDownload(self.ctx, out, self.ptr)

    def download(self, dtype, shape):
        out = self.ctx.pinned_empty(shape, dtype)            # large results: page-locked pool (one DMA, no first touch)
        if out.nbytes > self.nbytes:
            raise ValueError("download larger than buffer")
        check(self.ctx.lib.sarx_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes), self.ctx.h)
        return out


class PendingDownload:
    """A device-to-host copy in flight on the context's download stream (sarx_memcpy_d2h_begin); result() waits for it."""

    def __init__(self, ctx, out, src_ptr):
        self.ctx, self.out, self.slot = ctx, out, None
        if ctx.is_pinned(out):
            slot = C.c_int(-1)
            check(ctx.lib.sarx_memcpy_d2h_begin(ctx.h, out.ctypes.data, src_ptr, out.nbytes, C.byref(slot)), ctx.h)
            self.slot = slot.value
        else:                                   # no page-locked block to be had (pool budget): the ordinary blocking copy
            check(ctx.lib.sarx_memcpy_d2h(ctx.h, out.ctypes.data, src_ptr, out.nbytes), ctx.h)

    def result(self):
        if self.slot is not None:
            slot, self.slot = self.slot, None
            check(self.ctx.lib.sarx_memcpy_d2h_end(self.ctx.h, slot), self.ctx.h)
        return self.out

    def __del__(self):                          # dropped without result(): the copy is waited for and its slot returned to the context
        try:
            if self.slot is not None and self.ctx.h is not None:
                self.ctx.lib.sarx_memcpy_d2h_end(self.ctx.h, self.slot)
        except Exception:
            pass


def download_block(ctx, ptr, ld_elems, row0, n_rows, col0, n_cols, dtype=np.complex64):
    """Rows [row0, row0+n_rows) x columns [col0, col0+n_cols) of a row-major device image with leading dimension
    ``ld_elems`` (strided copy, blocking): how tests sample columns of a full-size image without downloading it."""
    item = np.dtype(dtype).itemsize
    out = np.empty((n_rows, n_cols), dtype=dtype)
    src = ptr + (int(row0) * int(ld_elems) + int(col0)) * item
    check(ctx.lib.sarx_memcpy2d_d2h(ctx.h, out.ctypes.data, n_cols * item, src, int(ld_elems) * item, n_cols * item,
                                    int(n_rows)), ctx.h)
    return out


def upload_block(ctx, ptr, ld_elems, row0, col0, block):
    """Inverse of download_block: writes a host [n_rows x n_cols] block into the device image."""
    block = np.ascontiguousarray(block)
    item = block.dtype.itemsize
    dst = ptr + (int(row0) * int(ld_elems) + int(col0)) * item
    check(ctx.lib.sarx_memcpy2d_h2d(ctx.h, dst, int(ld_elems) * item, block.ctypes.data, block.shape[1] * item,
                                    block.shape[1] * item, block.shape[0]), ctx.h)


class DeviceArray:
    """A row-major 2-D complex64 array living in a DeviceBuffer: what the echo generators return with ``device=True``
    and what the focusers accept in place of a NumPy array, so a scene can go from synthesis to products without
    visiting the host.  ``rows(a, b)`` is a view (the DPCA pulse shift of sar_ati_dcpa_sim_csa.py:402-403 is two
    such views)."""

    def __init__(self, buf, shape, offset=0, owner=True, transposed=False):
        self.buf, self.shape, self.offset, self.owner = buf, (int(shape[0]), int(shape[1])), int(offset), owner
        self.ctx = buf.ctx
        self.transposed = transposed       # .T of a row-major array: same memory, shape reversed (like a NumPy view)

    @property
    def T(self):
        return DeviceArray(self.buf, self.shape[::-1], self.offset, owner=False, transposed=not self.transposed)

    @property
    def ptr(self):
        return self.buf.ptr + self.offset

    @property
    def nbytes(self):
        return self.shape[0] * self.shape[1] * 8

    def rows(self, a, b):
        if self.transposed:
            raise ValueError("rows() of a transposed view is not contiguous")
        a, b, _ = slice(a, b).indices(self.shape[0])
        return DeviceArray(self.buf, (max(b - a, 0), self.shape[1]), self.offset + a * self.shape[1] * 8, owner=False)

    def numpy(self):
        out = np.empty(self.shape[::-1] if self.transposed else self.shape, dtype=np.complex64)
        check(self.ctx.lib.sarx_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes), self.ctx.h)
        return out.T if self.transposed else out

    def release(self):
        if self.owner:
            self.buf.release()


class _PinnedBlock:
    """A page-locked host block (sarx_host_alloc) behind the array interface.  NumPy keeps the object providing the interface
    alive as the base of every array and view made on it; when the last of them dies the block goes back to its context's pool."""

    def __init__(self, ctx, ptr, nbytes):
        self._ctx, self._ptr, self._nbytes = ctx, ptr, nbytes
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            self._ctx._pinned_return(self._ptr, self._nbytes)
        except Exception:
            pass


class Context:
    """sarx_ctx wrapper: one per GPU, owns a compute stream and a comm stream."""

    PINNED_MIN_BYTES = 64 << 20        # smaller results are plain NumPy arrays (the library copies them with one hipMemcpy anyway)
    PINNED_FREE_PER_SIZE = 4           # free blocks kept per size (a two-channel call returns three planes of one size); a caller that holds more results alive gets fresh blocks
    PINNED_FROM_REQUEST = 4            # a size earns page-locked blocks from its fourth request on: hipHostMalloc costs 0.15 s per GiB,
                                       # up to 2.6 x the first touch of a pageable result, so a one-shot script (two images, three planes,
                                       # three intermediate maps of one size at most) never pays it; a frame loop pays it twice

    def __init__(self, device_id=0):
        self.lib = _ffi.load()
        h = C.c_void_p()
        check(self.lib.sarx_init(int(device_id), C.byref(h)), None)
        self.h = h.value
        self.device_id = int(device_id)
        self._live = {}
        self._plans = weakref.WeakSet()
        self._lane_sets, self._lane_ratios = {}, {}
        self._scratch = {}             # (tag, lane) -> DeviceBuffer
        self._lane = 0
        self._pinned_free = {}         # nbytes -> [ptr, ...]
        self._pinned_seen = {}         # nbytes -> requests so far
        self._pinned_total = 0         # bytes handed out + bytes kept free
        self._pinned_cap = int(float(os.environ.get("SARX_PINNED_POOL_GIB", "16")) * 2 ** 30)
        self._pinned_lock = threading.RLock()         # re-entrant: a block may be finalised (GC) inside pinned_empty on the same thread

    def close(self):
        if self.h is not None:
            for p in list(self._plans):
                p.close()
            for ptr in list(self._live.values()):
                self.lib.sarx_free(self.h, ptr)
            self._live.clear()
            with self._pinned_lock:
                for ptrs in self._pinned_free.values():
                    for ptr in ptrs:
                        self.lib.sarx_host_free(self.h, ptr)
                self._pinned_free.clear()
            self.lib.sarx_destroy(self.h)
            self.h = None

    # -- small device scratch kept across calls --
    def scratch(self, tag, nbytes):
        """A device buffer of at least nbytes that lives as long as the context, one per (tag, current lane): tables a frame loop
        uploads every frame (target positions, platform track, ...) reuse it instead of a hipMalloc / hipFree pair per call."""
        key = (tag, self._lane)
        b = self._scratch.get(key)
        if b is None or b.ptr is None or b.nbytes < nbytes:
            if b is not None:
                b.release()
            b = self._scratch[key] = DeviceBuffer(self, max(int(nbytes), 256))
        return b

    def scratch_upload(self, tag, arr):
        """The (tag, current lane) scratch buffer is read by this lane's launches only: the upload waits for this lane alone.  (Waiting
        for every lane drained the GPU at each of a frame's seven table uploads: 364 idle gaps, 7 % of the VideoSAR loop, against 35 gaps
        and 0.8 % - tools/trace_busy.py on profiles/r05_bm_* / r05_bo_*.)"""
        arr = np.ascontiguousarray(arr)
        return self.scratch(tag, arr.nbytes).upload_lane(arr)

    # -- pooled page-locked result arrays --
    def last_error(self):
        """Text of the last error a call on this context set (sarx_last_error)."""
        return self.lib.sarx_last_error(self.h).decode(errors="replace")

    def reserve_pinned(self, shape, dtype, count=1):
        """Page-lock `count` result blocks of this shape now, so that a frame loop never meets hipHostMalloc (0.15 s per GiB) in the
        middle: the blocks wait in the pool and pinned_empty hands them out from the first request on.  Returns how many blocks of
        that size the pool holds free afterwards (fewer than asked for if the pool budget SARX_PINNED_POOL_GIB is exhausted)."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        with self._pinned_lock:
            free = self._pinned_free.setdefault(nbytes, [])
            while len(free) < int(count) and self._pinned_total + nbytes <= self._pinned_cap and self.h is not None:
                out = C.c_void_p()
                if self.lib.sarx_host_alloc(self.h, nbytes, C.byref(out)) != 0 or not out.value:
                    break
                free.append(out.value)
                self._pinned_total += nbytes
            self._pinned_seen[nbytes] = max(self._pinned_seen.get(nbytes, 0), self.PINNED_FROM_REQUEST)
            return len(free)

    def is_pinned(self, arr):
        """True if the array's memory is a block of this context's page-locked pool."""
        base = arr
        while isinstance(base, np.ndarray) and base.base is not None:
            base = base.base
        return isinstance(base, _PinnedBlock)

    def pinned_empty(self, shape, dtype, force=False):
        """np.empty(shape, dtype) on page-locked memory from a per-context pool (results of the *_host entry points): the
        download is one DMA at the PCIe rate and a repeated call of the same size pays neither hipHostMalloc nor the first touch
        of fresh pages.  The array is the caller's like any NumPy result; its block returns to the pool when the array and all
        views of it are gone.  Small results, the first three requests of a size (a one-shot script should not pay 0.15 s per GiB
        of hipHostMalloc for arrays it downloads once), an exhausted pool budget (SARX_PINNED_POOL_GIB, default 16) or a failed
        pinned allocation give an ordinary np.empty."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if nbytes < self.PINNED_MIN_BYTES or self.h is None:
            return np.empty(shape, dtype)
        with self._pinned_lock:
            seen = self._pinned_seen[nbytes] = self._pinned_seen.get(nbytes, 0) + 1
            free = self._pinned_free.get(nbytes)
            ptr = free.pop() if free else None
            if ptr is None:
                if (seen < self.PINNED_FROM_REQUEST and not force) or self._pinned_total + nbytes > self._pinned_cap:
                    return np.empty(shape, dtype)
                out = C.c_void_p()
                if self.lib.sarx_host_alloc(self.h, nbytes, C.byref(out)) != 0 or not out.value:
                    return np.empty(shape, dtype)
                ptr = out.value
                self._pinned_total += nbytes
        return np.asarray(_PinnedBlock(self, ptr, nbytes)).view(dtype).reshape(shape)

    def _pinned_return(self, ptr, nbytes):
        with self._pinned_lock:
            if self.h is None:             # context closed first: the block outlives it (freed with the process)
                return
            free = self._pinned_free.setdefault(nbytes, [])
            if len(free) < self.PINNED_FREE_PER_SIZE:
                free.append(ptr)
            else:
                self.lib.sarx_host_free(self.h, ptr)
                self._pinned_total -= nbytes

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory / sync / timing --
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return self.alloc(arr.nbytes).upload(arr)

    def select_lane(self, lane):
        """Every later enqueue of this context goes to compute stream `lane` (0..3): consecutive independent frames on
        alternating lanes - each with its own plan and buffers - overlap on the GPU (sarx_select_lane)."""
        check(self.lib.sarx_select_lane(self.h, int(lane)), self.h)
        self._lane = int(lane)

    def set_range_cus(self, cus):
        """Persistent range launches size their grid for `cus` compute units (0 = all): with frames in flight the rest of the
        chip stays available to the other lane's azimuth launches (sarx_set_range_cus)."""
        check(self.lib.sarx_set_range_cus(self.h, int(cus)), self.h)

    def probe_lanes(self, a, b, us=300):
        """together / alone time of two small launches on lanes a and b: ~1 = the lanes run side by side, ~2 = they share a hardware
        queue and take turns (sarx_probe_lanes)."""
        r = C.c_double()
        check(self.lib.sarx_probe_lanes(self.h, int(a), int(b), int(us), C.byref(r)), self.h)
        return r.value

    def concurrent_lanes(self, k=2):
        """k lane numbers (lane 0 first) that run side by side, for frames in flight: HIP may put two of a context's streams on one
        hardware queue, and frames in flight on such a pair gain nothing.  Probed once per context (a few milliseconds)."""
        k = max(1, min(int(k), 4))
        if k == 1:
            return [0]
        if self._lane_sets.get(k) is None:
            chosen, ratios = [0], {}
            for cand in (1, 2, 3):
                if len(chosen) == k:
                    break
                ratios[cand] = max(self.probe_lanes(x, cand) for x in chosen)
                if ratios[cand] < 1.5:
                    chosen.append(cand)
            for cand in sorted(ratios, key=ratios.get):          # not enough independent lanes: take the least bad ones
                if len(chosen) == k:
                    break
                if cand not in chosen:
                    chosen.append(cand)
            self._lane_sets[k] = chosen
            self._lane_ratios = ratios
        return list(self._lane_sets[k])

    def lanes_join(self):
        """On the device: every lane waits for everything enqueued so far on every lane."""
        check(self.lib.sarx_lanes_join(self.h), self.h)

    def sync(self):
        check(self.lib.sarx_sync(self.h), self.h)

    def record(self, slot):
        check(self.lib.sarx_event_record(self.h, slot), self.h)

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        check(self.lib.sarx_event_elapsed_ms(self.h, a, b, C.byref(ms)), self.h)
        return ms.value

    def info(self):
        name = C.create_string_buffer(256)
        arch = C.create_string_buffer(64)
        cus = C.c_int()
        hbm = C.c_uint64()
        check(self.lib.sarx_device_info(self.h, name, 256, C.byref(cus), C.byref(hbm), arch, 64), self.h)
        return {"name": name.value.decode(), "arch": arch.value.decode(), "compute_units": cus.value,
                "hbm_bytes": hbm.value}

    # -- kernels that are not tied to a plan --
    def fill_noise(self, buf, n, seed):
        check(self.lib.sarx_fill_noise_c64(self.h, buf.ptr, int(n), int(seed)), self.h)

    def echo_synth(self, d_tau_pb, d_amp, d_t_fast, n_pulses, n_targets, n_samples, kr, t_p, d_raw):
        check(self.lib.sarx_echo_synth_dev(self.h, d_tau_pb.ptr, d_amp.ptr, d_t_fast.ptr, int(n_pulses), int(n_targets),
                                           int(n_samples), float(kr), float(t_p), d_raw.ptr, 0), self.h)

    def corner_turn(self, src, dst, rows, cols):
        check(self.lib.sarx_corner_turn_dev(self.h, src.ptr, dst.ptr, int(rows), int(cols)), self.h)

    def multilook(self, src, dst, rows, cols, looks):
        check(self.lib.sarx_multilook_dev(self.h, src.ptr, dst.ptr, int(rows), int(cols), int(looks)), self.h)

    def mask_phase(self, phase, mag, n, thr, out):
        check(self.lib.sarx_mask_phase_dev(self.h, phase.ptr, mag.ptr, int(n), float(thr), out.ptr), self.h)

    def mask_phase_frac(self, phase, mag, n, frac, out):
        """Mask with thr = frac * max|slc1| of the most recent ati_dpca launch, taken on the device (no host sync)."""
        check(self.lib.sarx_mask_phase_frac_dev(self.h, phase.ptr, mag.ptr, int(n), float(frac), out.ptr), self.h)

    def ati_stats(self):
        """(max|slc1|, sum slc1*conj(slc2)) of the most recent ati_dpca launch (blocking)."""
        mx = C.c_double()
        sm = (C.c_double * 2)()
        check(self.lib.sarx_ati_stats(self.h, C.byref(mx), sm), self.h)
        return mx.value, complex(sm[0], sm[1])

    def magnitude(self, src, dst, n):
        check(self.lib.sarx_magnitude_dev(self.h, src.ptr, dst.ptr, int(n)), self.h)

    def ati_dpca(self, slc1, slc2, n, cal_phase, outs, want_stats=True):
        """outs: dict name -> DeviceBuffer for fields of sarx_ati_outputs."""
        o = _ffi.AtiOutputs()
        for k, _ in _ffi.AtiOutputs._fields_:
            b = outs.get(k)
            setattr(o, k, b.ptr if b is not None else None)
        mx = C.c_double()
        sm = (C.c_double * 2)()
        check(self.lib.sarx_ati_dpca_dev(self.h, slc1.ptr, slc2.ptr, int(n), float(cal_phase), C.byref(o),
                                         C.byref(mx) if want_stats else None, sm if want_stats else None), self.h)
        return (mx.value, complex(sm[0], sm[1])) if want_stats else None

    def ati_dpca_masked(self, slc1, slc2, n, cal_phase, d_max, mask_frac, outs):
        """The same launch with the magnitude mask applied on the way out: outs['ati_phase'] receives the masked phase
        (sar_ati_dcpa_sim_csa.py:447-449); d_max = device float holding max|slc1| (CsaPlan.set_max_slot).  Only enqueues."""
        o = _ffi.AtiOutputs()
        for k, _ in _ffi.AtiOutputs._fields_:
            b = outs.get(k)
            setattr(o, k, b.ptr if b is not None else None)
        check(self.lib.sarx_ati_dpca_masked_dev(self.h, slc1.ptr, slc2.ptr, int(n), float(cal_phase), d_max.ptr, float(mask_frac),
                                                C.byref(o)), self.h)

    # -- RCCL --
    @staticmethod
    def rccl_info():
        """{'path', 'version', 'header_version'} of the RCCL the collectives run on (loads it)."""
        path = C.create_string_buffer(1024)
        ver, hdr = C.c_int(), C.c_int()
        check(_ffi.load().sarx_rccl_info(path, 1024, C.byref(ver), C.byref(hdr)), None)
        return {"path": path.value.decode(), "version": ver.value, "header_version": hdr.value}

    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(_ffi.COMM_ID_BYTES)
        check(_ffi.load().sarx_comm_unique_id(buf), None)
        return buf.raw

    def comm_init(self, uid, n_ranks, rank):
        check(self.lib.sarx_comm_init(self.h, uid, int(n_ranks), int(rank)), self.h)

    def allgather(self, send, recv, bytes_per_rank):
        check(self.lib.sarx_allgather_dev(self.h, send.ptr, recv.ptr, int(bytes_per_rank)), self.h)

    def max_abs(self, buf, n, d_max):
        """*d_max = max(*d_max, max|buf[:n]|) over an fp32 device buffer (sarx_max_abs_f32_dev); clear d_max first."""
        check(self.lib.sarx_max_abs_f32_dev(self.h, buf.ptr, int(n), d_max.ptr), self.h)

    def allreduce_max(self, buf, count=1):
        """In-place RCCL all-reduce(max) of `count` floats on the comm stream, ordered after the compute stream."""
        check(self.lib.sarx_allreduce_max_dev(self.h, buf.ptr, int(count)), self.h)

    def comm_fence_compute(self):
        """Later compute-stream work waits (on the device) for every gather enqueued so far."""
        check(self.lib.sarx_comm_fence_compute(self.h), self.h)

    def comm_sync(self):
        check(self.lib.sarx_comm_sync(self.h), self.h)


class CsaPlan:
    """sarx_plan wrapper for one (n_az, n_rg, radar) geometry."""

    def __init__(self, ctx, n_az, n_rg, wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz,
                 prf_hz, platform_speed_mps, range_ref_m, t_start_fast, flags=_ffi.OUT_AZ_MAJOR):
        self.ctx = ctx
        self.n_az, self.n_rg, self.flags = int(n_az), int(n_rg), int(flags)
        self.params = _ffi.RadarParams(wavelength_m, pulse_width_sec, chirp_rate_hzpsec, sample_rate_hz, prf_hz,
                                       platform_speed_mps, range_ref_m, t_start_fast)
        h = C.c_void_p()
        check(ctx.lib.sarx_csa_plan_create(ctx.h, self.n_az, self.n_rg, C.byref(self.params), self.flags,
                                           C.byref(h)), ctx.h)
        self.h = h.value
        ctx._plans.add(self)

    def close(self):
        if self.h is not None and self.ctx.h is not None:
            self.ctx.lib.sarx_csa_plan_destroy(self.h)
        self.h = None
        buf, self._result_buf = getattr(self, "_result_buf", None), None
        if buf is not None:
            buf.release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def result_buffer(self):
        """A device image buffer owned by the plan and kept until it is closed: where the facade focuses a device-resident echo
        before the blocking download (a fresh allocation per call costs its first touch - a download out of fresh device pages
        ran 62 ms against 38 ms for 1.75 GiB - and hipFree waits for the device)."""
        if getattr(self, "_result_buf", None) is None or self._result_buf.ptr is None:
            self._result_buf = self.ctx.alloc(self.n_az * self.n_rg * 8)
        return self._result_buf

    @property
    def rg_major(self):
        return bool(self.flags & _ffi.OUT_RG_MAJOR)

    def axes(self):
        ra = np.empty(self.n_rg, dtype=np.float64)
        ca = np.empty(self.n_az, dtype=np.float64)
        check(self.ctx.lib.sarx_csa_axes(self.h, ra.ctypes.data, ca.ctypes.data), self.ctx.h)
        return ra, ca

    def scratch_bytes(self):
        b = C.c_uint64()
        check(self.ctx.lib.sarx_csa_plan_bytes(self.h, C.byref(b)), self.ctx.h)
        return b.value

    def focus_host(self, phist_c64, out=None):
        """[n_az x n_rg] complex64 host array -> focused image in the plan's layout (host).  The result array comes from the
        context's page-locked pool (Context.pinned_empty) unless `out` - a C-contiguous complex64 array of the result's shape,
        e.g. an earlier result - is given."""
        a = np.asarray(phist_c64)
        wide = a.dtype == np.complex128 and a.flags.c_contiguous        # the reference's dtype: narrowed by the library's copy threads
        if not wide:
            a = np.ascontiguousarray(a, dtype=np.complex64)
        if a.shape != (self.n_az, self.n_rg):
            raise ValueError(f"phist shape {a.shape} != plan ({self.n_az}, {self.n_rg})")
        shape = (self.n_rg, self.n_az) if self.rg_major else (self.n_az, self.n_rg)
        if out is None:
            out = self.ctx.pinned_empty(shape, np.complex64)
        elif not (isinstance(out, np.ndarray) and out.dtype == np.complex64 and out.shape == shape and out.flags.c_contiguous
                  and out.flags.writeable):
            raise ValueError(f"out must be a writeable C-contiguous complex64 array of shape {shape}")
        fn = self.ctx.lib.sarx_csa_focus_host_c128 if wide else self.ctx.lib.sarx_csa_focus_host
        check(fn(self.h, a.ctypes.data, out.ctypes.data), self.ctx.h)
        return out

    def focus_host_begin(self, phist, out=None):
        """First half of focus_host as a two-deep pipeline (sarx_csa_focus_host_begin): uploads this frame without waiting for the one
        that is still focusing / downloading, enqueues its focus and the download of its image into `out` (default: a page-locked
        array from the context's pool) and returns (ticket, out).  focus_host_end(ticket) waits for the image.  `phist` must stay
        untouched until then; at most two frames per plan are in flight."""
        a = np.ascontiguousarray(phist, dtype=np.complex64)
        if a.shape != (self.n_az, self.n_rg):
            raise ValueError(f"phist shape {a.shape} != plan ({self.n_az}, {self.n_rg})")
        shape = (self.n_rg, self.n_az) if self.rg_major else (self.n_az, self.n_rg)
        if out is None:
            out = self.ctx.pinned_empty(shape, np.complex64, force=True)
        elif not (isinstance(out, np.ndarray) and out.dtype == np.complex64 and out.shape == shape and out.flags.c_contiguous
                  and out.flags.writeable):
            raise ValueError(f"out must be a writeable C-contiguous complex64 array of shape {shape}")
        ticket = C.c_int(-1)
        check(self.ctx.lib.sarx_csa_focus_host_begin(self.h, a.ctypes.data, out.ctypes.data, C.byref(ticket)), self.ctx.h)
        return (ticket.value, a), out           # the ticket keeps the (possibly converted) input alive until _end

    def focus_host_end(self, ticket):
        check(self.ctx.lib.sarx_csa_focus_host_end(self.h, int(ticket[0])), self.ctx.h)

    def mark_range(self, slot_start=-1, slot_stop=-1):
        """focus_dev records ctx events around its range pass(es) (roofline kernel timing)."""
        check(self.ctx.lib.sarx_csa_plan_mark_range(self.h, int(slot_start), int(slot_stop)), self.ctx.h)

    def stamp_range(self, pair_ptr=None):
        """focus_dev's fused range launch leaves {first workgroup start, last workgroup end} (100 MHz ticks) in the two uint64 at
        the device address pair_ptr, which the caller has initialised to {2**64 - 1, 0}; None switches it off (sarx_csa_plan_stamp_range)."""
        check(self.ctx.lib.sarx_csa_plan_stamp_range(self.h, pair_ptr if pair_ptr else None), self.ctx.h)

    def set_look_slot(self, looks, slot_ptr):
        """Every later focus_dev also writes the looks x looks multilook of |image|^2 to the device address slot_ptr
        (None / 0 switches it off): the VideoSAR stack slot without reading the image again."""
        check(self.ctx.lib.sarx_csa_plan_set_look_slot(self.h, int(looks), slot_ptr if slot_ptr else None), self.ctx.h)

    def set_max_slot(self, d_max):
        """Every later focus_dev also leaves max|image| (fp32) in the device buffer d_max (None switches it off): the threshold
        input of Context.ati_dpca_masked without a further pass over the image."""
        check(self.ctx.lib.sarx_csa_plan_set_max_slot(self.h, d_max.ptr if d_max is not None else None), self.ctx.h)

    def set_ati(self, d_slc1, d_max=None, mask_frac=0.05, cal_phase=0.0, masked=None, slc1_mag=None, dpca_mag=None, keep_image=False):
        """Every later focus_dev (the SECOND channel's) emits the ATI / DPCA products of (d_slc1, its own image) from its last
        azimuth launch into the three fp32 device planes; d_slc1=None switches it off (sarx_csa_plan_set_ati)."""
        if d_slc1 is None:
            check(self.ctx.lib.sarx_csa_plan_set_ati(self.h, None, None, 0.0, 0.0, None, None, None, 0), self.ctx.h)
            return
        check(self.ctx.lib.sarx_csa_plan_set_ati(self.h, d_slc1.ptr, d_max.ptr, float(mask_frac), float(cal_phase), masked.ptr,
                                                 slc1_mag.ptr, dpca_mag.ptr, int(bool(keep_image))), self.ctx.h)

    def focus_dev(self, d_phist, d_image):
        check(self.ctx.lib.sarx_csa_focus_dev(self.h, d_phist.ptr, d_image.ptr), self.ctx.h)

    def run_pass(self, pass_id, d_in, d_out):
        check(self.ctx.lib.sarx_csa_pass(self.h, int(pass_id), d_in.ptr, d_out.ptr), self.ctx.h)


class FocusLanes:
    """A frame loop's focuser with frames in flight: consecutive focus_dev calls go to alternating compute lanes of the context,
    each lane with its own CsaPlan (scratch), so the launches of neighbouring frames share the GPU (sarx_select_lane; DESIGN.md
    4.8).  The caller gives every frame in flight its own input and output buffers; results equal CsaPlan.focus_dev bit for bit.

        fl = sarx.FocusLanes(ctx, n_az, n_rg, *focus_args)            # frames are independent (sar_batch_sim.py:303-331)
        for f in range(n_frames):
            fl.focus_dev(d_echo[f], d_image[f])                        # only enqueues
        fl.finish()                                                    # lane 0 selected again, every lane joined; ctx.sync() to wait

    lanes=None: two for frames of 4096^2 samples and more, one below (small frames are launch-bound)."""

    def __init__(self, ctx, n_az, n_rg, *focus_args, lanes=None, flags=_ffi.FUSE_RANGE, range_cus=192):
        if lanes is None:
            lanes = 2 if int(n_az) * int(n_rg) >= 4096 * 4096 else 1
        self.ctx, self.lanes, self.range_cus = ctx, max(1, min(int(lanes), 4)), int(range_cus)
        self.plans = [CsaPlan(ctx, n_az, n_rg, *focus_args, flags=flags) for _ in range(self.lanes)]
        self.lane_ids = ctx.concurrent_lanes(self.lanes)       # lanes that really run side by side (probed once per context)
        self._i = 0

    def focus_dev(self, d_phist, d_image):
        lane = self._i % self.lanes
        self._i += 1
        if self.lanes > 1:
            self.ctx.select_lane(self.lane_ids[lane])
            self.ctx.set_range_cus(self.range_cus)     # the persistent range launch leaves CUs to the other lane's azimuth tiles
        try:
            self.plans[lane].focus_dev(d_phist, d_image)
        except BaseException:                          # the CU share and the lane are context state: not left behind by a failed enqueue
            if self.lanes > 1:
                self.ctx.select_lane(0)
                self.ctx.set_range_cus(0)
            raise
        return lane

    def finish(self):
        """Back to lane 0 with every lane joined (device-side): whatever is enqueued next sees all frames finished."""
        if self.lanes > 1:
            self.ctx.select_lane(0)
            self.ctx.set_range_cus(0)
            self.ctx.lanes_join()
        self._i = 0

    def close(self):
        self.finish()
        for p in self.plans:
            p.close()
        self.plans = []


_default_ctx = {}


def device_count():
    """Number of HIP devices visible to this process (does not create a context)."""
    n = C.c_int()
    check(_ffi.load().sarx_device_count(C.byref(n)), None)
    return n.value


def default_context(device_id=0):
    """Process-wide Context per device (created on first use)."""
    c = _default_ctx.get(device_id)
    if c is None or c.h is None:
        c = _default_ctx[device_id] = Context(device_id)
    return c


__all__ = ["Context", "CsaPlan", "DeviceBuffer", "SarxError", "default_context", "download_block", "upload_block"]
