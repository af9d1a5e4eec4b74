"""ctypes binding of libsarx.so (include/sarx.h).  No torch, no cffi.

The library is built in-tree by ``csrc/Makefile`` (``__graft_entry__.build()``).
There is no CPU fallback: every compute entry point needs a gfx950 device and
raises :class:`SarxError` otherwise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SARX_LIB") or os.path.join(_HERE, "libsarx.so")

MAX_SLOT_BYTES = 32768          # SARX_MAX_SLOT_BYTES: 256 partial maxima, 32 floats apart
COMM_ID_BYTES = 128
OUT_AZ_MAJOR, OUT_RG_MAJOR, FUSE_RANGE = 0, 1, 2
PASS_AZ_FFT_PHI1, PASS_RG_FFT_PHI2, PASS_RG_IFFT_PHI3, PASS_AZ_IFFT, PASS_RG_FUSED_23 = 1, 2, 3, 4, 23
PASS_TEST_RG_FFT, PASS_TEST_RG_IFFT = 100, 101
PASS_RG_FFT_PHI2_PERM, PASS_RG_IFFT_PHI3_PERM = 12, 13      # n_rg = 16384: spectrum order P[(k // 16 // 64) * 1024 + (k % 16) * 64 + (k // 16) % 64] = X[k]


class SarxError(RuntimeError):
    """Raised with sarx_last_error() text when a libsarx call fails."""

    def __init__(self, code, msg):
        super().__init__(f"libsarx error {code}: {msg}")
        self.code = code


class RadarParams(C.Structure):
    """sarx_radar_params: positional args of sar_focus_csa after phist
    (sar_ati_dcpa_sim_csa.py:202)."""
    _fields_ = [(n, C.c_double) for n in (
        "wavelength_m", "pulse_width_s", "chirp_rate_hz_s", "sample_rate_hz", "prf_hz",
        "platform_speed_mps", "range_ref_m", "t_start_fast_s")]


class TdbpParams(C.Structure):
    """sarx_tdbp_params: the module constants tdbp_gpu reads (sar_batch_sim.py:13,20,23-25)."""
    _fields_ = [(n, C.c_double) for n in ("c", "fc", "fs", "t_p", "k_rate")]


class AtiOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ati_phase", "slc1_mag", "dpca_mag", "ati_interf", "dpca_diff", "slc2_mag", "slc1_phase",
        "slc2_phase", "dpca_phase")]


# name -> (restype, argtypes): every symbol declared in include/sarx.h
_vp, _i, _sz, _u64, _d, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_uint64, C.c_double, C.c_float
_P = C.POINTER
SIGNATURES = {
    "sarx_version": (_i, []),
    "sarx_init": (_i, [_i, _P(_vp)]),
    "sarx_destroy": (_i, [_vp]),
    "sarx_last_error": (C.c_char_p, [_vp]),
    "sarx_device_count": (_i, [_P(_i)]),
    "sarx_device_info": (_i, [_vp, C.c_char_p, _sz, _P(_i), _P(_u64), C.c_char_p, _sz]),
    "sarx_malloc": (_i, [_vp, _sz, _P(_vp)]),
    "sarx_free": (_i, [_vp, _vp]),
    "sarx_host_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "sarx_host_free": (_i, [_vp, _vp]),
    "sarx_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "sarx_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "sarx_memcpy_d2d": (_i, [_vp, _vp, _vp, _sz]),
    "sarx_memcpy_h2d_unordered": (_i, [_vp, _vp, _vp, _sz]),
    "sarx_memcpy_h2d_lane": (_i, [_vp, _vp, _vp, _sz]),
    "sarx_memcpy_d2h_begin": (_i, [_vp, _vp, _vp, _sz, _P(_i)]),
    "sarx_memcpy_d2h_end": (_i, [_vp, _i]),
    "sarx_memcpy2d_d2h": (_i, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    "sarx_memcpy2d_h2d": (_i, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    "sarx_memset": (_i, [_vp, _vp, _i, _sz]),
    "sarx_persistent_grid": (_i, [_i, _i, _i]),
    "sarx_sync": (_i, [_vp]),
    "sarx_select_lane": (_i, [_vp, _i]),
    "sarx_lanes_join": (_i, [_vp]),
    "sarx_set_range_cus": (_i, [_vp, _i]),
    "sarx_probe_lanes": (_i, [_vp, _i, _i, _i, _P(_d)]),
    "sarx_event_record": (_i, [_vp, _i]),
    "sarx_event_elapsed_ms": (_i, [_vp, _i, _i, _P(_f)]),
    "sarx_csa_plan_create": (_i, [_vp, _i, _i, _P(RadarParams), C.c_uint, _P(_vp)]),
    "sarx_csa_plan_destroy": (_i, [_vp]),
    "sarx_csa_axes": (_i, [_vp, _vp, _vp]),
    "sarx_csa_focus_host": (_i, [_vp, _vp, _vp]),
    "sarx_csa_focus_host_begin": (_i, [_vp, _vp, _vp, _P(_i)]),
    "sarx_csa_focus_host_end": (_i, [_vp, _i]),
    "sarx_csa_focus_host_c128": (_i, [_vp, _vp, _vp]),
    "sarx_csa_focus_dev": (_i, [_vp, _vp, _vp]),
    "sarx_csa_pass": (_i, [_vp, _i, _vp, _vp]),
    "sarx_csa_plan_mark_range": (_i, [_vp, _i, _i]),
    "sarx_csa_plan_stamp_range": (_i, [_vp, _vp]),
    "sarx_csa_plan_set_look_slot": (_i, [_vp, _i, _vp]),
    "sarx_csa_plan_set_max_slot": (_i, [_vp, _vp]),
    "sarx_csa_plan_set_ati": (_i, [_vp, _vp, _vp, _f, _d, _vp, _vp, _vp, _i]),
    "sarx_csa_plan_bytes": (_i, [_vp, _P(_u64)]),
    "sarx_rda_plan_create": (_i, [_vp, _i, _i, _P(RadarParams), _P(_vp)]),
    "sarx_rda_plan_destroy": (_i, [_vp]),
    "sarx_rda_focus_host": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "sarx_rda_focus_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "sarx_rda_focus_host2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sarx_rda_focus_dev2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sarx_rda_axes": (_i, [_vp, _vp, _vp, _vp]),
    "sarx_ati_dpca_dev": (_i, [_vp, _vp, _vp, _sz, _d, _P(AtiOutputs), _P(_d), _P(_d)]),
    "sarx_ati_stats": (_i, [_vp, _P(_d), _P(_d)]),
    "sarx_ati_dpca_masked_dev": (_i, [_vp, _vp, _vp, _sz, _d, _vp, _f, _P(AtiOutputs)]),
    "sarx_mask_phase_frac_dev": (_i, [_vp, _vp, _vp, _sz, _f, _vp]),
    "sarx_magnitude_dev": (_i, [_vp, _vp, _vp, _sz]),
    "sarx_max_abs_f32_dev": (_i, [_vp, _vp, _sz, _vp]),
    "sarx_mask_phase_dev": (_i, [_vp, _vp, _vp, _sz, _f, _vp]),
    "sarx_corner_turn_dev": (_i, [_vp, _vp, _vp, _i, _i]),
    "sarx_multilook_dev": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "sarx_fill_noise_c64": (_i, [_vp, _vp, _sz, _u64]),
    "sarx_add_ocean_noise_dev": (_i, [_vp, _vp, _sz, _d, _d, _d, _u64]),
    "sarx_add_ocean_noise_rel_dev": (_i, [_vp, _vp, _sz, _i, _d, _d, _d, _u64]),
    "sarx_power_stats_dev": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "sarx_echo_synth_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _d, _d, _vp, _i]),
    "sarx_echo_geometry_dev": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _d, _vp, _vp]),
    "sarx_echo_spotlight_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _d, _d, _vp]),
    "sarx_tdbp_plan_create": (_i, [_vp, _i, _i, _i, _i, _P(TdbpParams), _P(_vp)]),
    "sarx_tdbp_plan_destroy": (_i, [_vp]),
    "sarx_tdbp_last_window": (_i, [_vp, _vp, _vp]),
    "sarx_tdbp_focus_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _vp, _d, _vp]),
    "sarx_tdbp_focus_host": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _vp, _d, _vp, _vp]),
    "sarx_comm_unique_id": (_i, [_vp]),
    "sarx_rccl_info": (_i, [C.c_char_p, _sz, _P(_i), _P(_i)]),
    "sarx_comm_init": (_i, [_vp, _vp, _i, _i]),
    "sarx_allgather_dev": (_i, [_vp, _vp, _vp, _sz]),
    "sarx_allreduce_max_dev": (_i, [_vp, _vp, _sz]),
    "sarx_comm_sync": (_i, [_vp]),
    "sarx_comm_fence_compute": (_i, [_vp]),
    "sarx_comm_mark": (_i, [_vp, _i]),
    "sarx_comm_wait_mark": (_i, [_vp, _i]),
    "sarx_comm_destroy": (_i, [_vp]),
}

_lib = None


def load():
    """Load libsarx.so once and attach prototypes.  Fails loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SarxError(-3, f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C nis-sar-amtigmti-video_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, ctx=None):
    if rc != 0:
        msg = load().sarx_last_error(ctx)
        raise SarxError(rc, msg.decode() if msg else "unknown")
