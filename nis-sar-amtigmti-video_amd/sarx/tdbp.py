"""VideoSAR back-projection path with the reference's names and argument order:
``run_physics_spotlight`` (sar_batch_sim.py:85-169), ``tdbp_gpu`` (:171-238), ``calculate_raw_snr_db`` (:54-64),
``generate_noise_tensor`` (:66-82) and the orbit arc of ``main`` (:258-268).

The reference functions read module globals; here they are the keyword argument ``consts`` (a dict with the
reference's names C, FC, FS, T_P, K_RATE, R0, Lambda ...), defaulting to ``batch_constants()`` = the literal block
:12-50.  Geometry per pulse and target (fp64), the sample loop and the whole back-projection run in HIP kernels.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import check
from .echo import synth_device
from .engine import DeviceBuffer, default_context

def batch_constants():
    """sar_batch_sim.py:12-50."""
    k = {"C": 299792458.0, "Re": 6371000.0, "h": 350000.0, "GM": 3.986004418e14}
    k["R_sat"] = k["Re"] + k["h"]
    k["V_sat"] = np.sqrt(k["GM"] / k["R_sat"])
    k["FC"], k["BW"], k["T_P"], k["FS"], k["PRF"] = 9.65e9, 500e6, 20e-6, 600e6, 5000.0
    k["Lambda"] = k["C"] / k["FC"]
    k["K_RATE"] = k["BW"] / k["T_P"]
    look = np.radians(45.0)
    gamma = np.arcsin((k["R_sat"] / k["Re"]) * np.sin(look)) - look
    k["S0_from_C"] = np.array([0, -k["R_sat"] * np.sin(gamma), k["R_sat"] * np.cos(gamma)])
    k["V_unit"] = np.array([1.0, 0.0, 0.0])
    k["C_offset"] = np.array([0, 0, -k["Re"]])
    k["R0"] = np.linalg.norm(k["S0_from_C"] + k["C_offset"])
    k.update(P_TX=1000.0, ANT_WIDTH=0.5, T_SYS=290.0, NF_DB=5.0, LOSS_DB=3.0, K_BOLTZ=1.380649e-23, SCR_DB=10.0,
             K_NU=1.0, SNR_BOOST_DB=26.0)
    return k


def orbit_arc(t_vec, consts=None):
    """Platform positions and velocities on the circular orbit (sar_batch_sim.py:262-268): ([n x 3], [n x 3])."""
    k = consts or batch_constants()
    omega = k["V_sat"] / k["R_sat"]
    wt = omega * np.asarray(t_vec, dtype=np.float64)[:, None]
    pos = k["S0_from_C"][None, :] * np.cos(wt) + (k["R_sat"] * k["V_unit"])[None, :] * np.sin(wt) + k["C_offset"][None, :]
    vel = (k["V_sat"] * k["V_unit"])[None, :] * np.cos(wt) - (k["S0_from_C"] * omega)[None, :] * np.sin(wt)
    return pos, vel


def calculate_raw_snr_db(r_slant, rcs, wavelength, bandwidth, ant_l, p_tx=None, ant_w=None, t_sys=None, nf_db=None,
                         loss_db=None, *, consts=None):
    """Radar-equation SNR of the raw data in dB (sar_batch_sim.py:54-64)."""
    k = consts or batch_constants()
    p_tx = k["P_TX"] if p_tx is None else p_tx
    ant_w = k["ANT_WIDTH"] if ant_w is None else ant_w
    t_sys = k["T_SYS"] if t_sys is None else t_sys
    nf_db = k["NF_DB"] if nf_db is None else nf_db
    loss_db = k["LOSS_DB"] if loss_db is None else loss_db
    gain = 4 * np.pi * (ant_l * ant_w * 0.6) / (wavelength ** 2)
    numerator = p_tx * (gain ** 2) * (wavelength ** 2) * rcs
    denominator = ((4 * np.pi) ** 3) * (r_slant ** 4) * k["K_BOLTZ"] * t_sys * bandwidth * (10 ** (loss_db / 10)) * (10 ** (nf_db / 10))
    return 10 * np.log10(numerator / denominator)


def generate_noise_tensor(shape, ref_power, snr_db, scr_db=None, k_nu=None, *, seed=None, consts=None):
    """Thermal noise + K-distributed sea clutter (sar_batch_sim.py:66-82), complex64 on the host.
    The reference draws from torch's global generator; here a NumPy Generator seeded by ``seed``."""
    k = consts or batch_constants()
    scr_db = k["SCR_DB"] if scr_db is None else scr_db
    k_nu = k["K_NU"] if k_nu is None else k_nu
    rng = np.random.default_rng(seed)
    noise_std = np.sqrt(ref_power / (10 ** (snr_db / 10)) / 2)
    thermal = noise_std * (rng.standard_normal(shape, dtype=np.float32) + 1j * rng.standard_normal(shape, dtype=np.float32))
    clutter_power = ref_power / (10 ** (scr_db / 10))
    texture = rng.gamma(k_nu, 1.0 / k_nu, shape)                     # torch Gamma(concentration=k, rate=k)
    speckle = rng.exponential(1.0, shape)
    amp = np.sqrt(clutter_power * texture * speckle)
    phase = rng.random(shape) * 2 * np.pi
    return (thermal + amp * np.exp(1j * phase)).astype(np.complex64)


def spotlight_window(consts=None):
    """(num_samples, t_start, t_fast_abs) of sar_batch_sim.py:86-91."""
    k = consts or batch_constants()
    win_len = (2000.0 / k["C"]) + k["T_P"] + 10e-6
    num_samples = int(np.ceil(win_len * k["FS"]))
    if num_samples % 2 != 0:
        num_samples += 1
    t_start = 2 * k["R0"] / k["C"] - win_len / 2
    return num_samples, t_start, t_start + np.arange(num_samples) / k["FS"]


def run_physics_spotlight(base_targets, t_vec, pos_sat, vel_sat, heading_deg, speed, l_ant, *, consts=None, ctx=None,
                          device=False, out=None, sync=True):
    """Spotlight echo of a rigid target moving at ``speed`` along ``heading_deg``; drop-in for
    sar_batch_sim.py:85-169.  returns (raw [len(t_vec) x num_samples] complex64, t_start, num_samples, v_tgt).
    ``device=True`` leaves raw on the GPU (a DeviceBuffer) for ``tdbp_gpu``; ``out`` (implies device) is a
    DeviceBuffer of at least len(t_vec) * num_samples * 8 bytes to fill instead of allocating one per call; ``sync=False``
    (device output only) returns as soon as the kernels are enqueued on the current lane."""
    k = consts or batch_constants()
    Cc, FC, T_P, K_RATE, Lambda = k["C"], k["FC"], k["T_P"], k["K_RATE"], k["Lambda"]
    ctx = ctx or default_context()
    num_samples, t_start, t_fast_abs = spotlight_window(k)
    phi = np.radians(heading_deg)
    v_tgt = np.array([speed * np.cos(phi), speed * np.sin(phi), 0])
    c, s = np.cos(phi), np.sin(phi)
    r_mat = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    p0 = np.array([r_mat @ np.asarray(t["position"], dtype=np.float64) for t in base_targets])      # :101
    rcs = np.array([t["rcs"] for t in base_targets], dtype=np.float64)
    t_vec = np.asarray(t_vec, dtype=np.float64)
    pos_sat = np.asarray(pos_sat, dtype=np.float64)
    vel_sat = np.asarray(vel_sat, dtype=np.float64)
    # per pulse and target: moved target (:127), bistatic delay with the receiver displaced by v_sat * 2 d_tx / C
    # (:128-133), antenna pattern (:134-144), amplitude rcs * gain (:150): geometry kernel, then the sample kernel
    d_raw = synth_device(ctx, 2, p0, v_tgt, t_vec, pos_sat, vel_sat, rcs, t_fast_abs, K_RATE, T_P, Cc, FC, l_ant=l_ant,
                         wavelength=Lambda, out=out, sync=sync or not (device or out is not None))
    if device or out is not None:
        return d_raw, t_start, num_samples, v_tgt
    raw = d_raw.download(np.complex64, (t_vec.size, num_samples))
    d_raw.release()
    return raw, t_start, num_samples, v_tgt


class TdbpPlan:
    """sarx_tdbp_plan: reference-chirp spectrum and scratch for one (n_pulses, num_samples, nx, ny)."""

    def __init__(self, ctx, n_pulses, num_samples, nx, ny, consts):
        self.ctx = ctx
        self.shape = (int(n_pulses), int(num_samples), int(nx), int(ny))
        prm = _ffi.TdbpParams(consts["C"], consts["FC"], consts["FS"], consts["T_P"], consts["K_RATE"])
        self.h = C.c_void_p()
        check(ctx.lib.sarx_tdbp_plan_create(ctx.h, *self.shape, C.byref(prm), C.byref(self.h)), ctx.h)
        ctx._plans.add(self)          # closed with the context, before sarx_destroy
        self._d_img = None            # the device image of the device-in path, kept across frames (a frame loop allocated and freed it per frame)

    def focus(self, raw, pos_plat, vel_plat, t_start, vel_focus, t_pulses, scene_size, want_rc=False, d_image=None):
        n_p, n_s, nx, ny = self.shape
        pos = np.ascontiguousarray(pos_plat, dtype=np.float64)
        vel = np.ascontiguousarray(vel_plat, dtype=np.float64)
        tp = np.ascontiguousarray(t_pulses, dtype=np.float64)
        vf = np.ascontiguousarray(vel_focus, dtype=np.float64)
        if pos.shape != (n_p, 3) or vel.shape != (n_p, 3) or tp.shape != (n_p,) or vf.shape != (3,):
            raise ValueError("pos_plat/vel_plat must be [n_pulses x 3], t_pulses [n_pulses], vel_focus [3]")
        img = np.empty((ny, nx), dtype=np.complex128)
        lib, ctx = self.ctx.lib, self.ctx
        if isinstance(raw, DeviceBuffer) and d_image is not None:      # device in, device out: only enqueues, returns nothing
            check(lib.sarx_tdbp_focus_dev(self.h, raw.ptr, pos.ctypes.data, vel.ctypes.data, tp.ctypes.data, float(t_start),
                                          vf.ctypes.data, float(scene_size), d_image.ptr), ctx.h)
            return None
        if d_image is not None:
            raise ValueError("d_image needs a device input")
        if isinstance(raw, DeviceBuffer):
            if self._d_img is None:
                self._d_img = ctx.alloc(img.nbytes)
            check(lib.sarx_tdbp_focus_dev(self.h, raw.ptr, pos.ctypes.data, vel.ctypes.data, tp.ctypes.data, float(t_start),
                                          vf.ctypes.data, float(scene_size), self._d_img.ptr), ctx.h)
            return self._d_img.download(np.complex128, (ny, nx))
        x = np.ascontiguousarray(raw, dtype=np.complex64)
        if x.shape != (n_p, n_s):
            raise ValueError(f"raw must be [{n_p} x {n_s}]")
        rc = np.empty((n_p, n_s), dtype=np.complex64) if want_rc else None
        check(lib.sarx_tdbp_focus_host(self.h, x.ctypes.data, pos.ctypes.data, vel.ctypes.data, tp.ctypes.data,
                                       float(t_start), vf.ctypes.data, float(scene_size), img.ctypes.data,
                                       rc.ctypes.data if want_rc else None), ctx.h)
        return (img, rc) if want_rc else img

    def last_window(self):
        """[lo, hi) samples of each pulse that the last focus call range-compressed."""
        lo, hi = C.c_int(), C.c_int()
        check(self.ctx.lib.sarx_tdbp_last_window(self.h, C.byref(lo), C.byref(hi)), self.ctx.h)
        return lo.value, hi.value

    def close(self):
        if self.h and self.ctx.h is not None:
            self.ctx.lib.sarx_tdbp_plan_destroy(self.h)
            if self._d_img is not None:
                self._d_img.release()
        self._d_img = None
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_plans = {}


def tdbp_gpu(raw_t, pos_plat, vel_plat, t_start, num_samples, vel_focus, t_pulses, scene_size, nx=512, ny=512, *,
             consts=None, ctx=None, d_image=None):
    """Back-projection image of one CPI; drop-in for sar_batch_sim.py:171-238.
    raw_t: [n_pulses x num_samples] complex (NumPy array, or the DeviceBuffer of run_physics_spotlight(device=True)).
    returns complex128 [ny x nx].  ``d_image`` (device input only): a device buffer / address holder for the complex128 [ny x nx]
    image - the call then only enqueues on the current lane and returns None (a frame loop with frames in flight; every lane gets
    its own plan, i.e. its own scratch)."""
    k = consts or batch_constants()
    ctx = ctx or default_context()
    n_p = len(t_pulses)
    key = (id(ctx), getattr(ctx, "_lane", 0), n_p, int(num_samples), int(nx), int(ny), k["C"], k["FC"], k["FS"], k["T_P"], k["K_RATE"])
    plan = _plans.get(key)
    if plan is None or plan.h is None:         # closed together with its Context: never reuse a dead handle
        _plans.pop(key, None)
        if len(_plans) >= 8:
            _plans.pop(next(iter(_plans))).close()
        plan = _plans[key] = TdbpPlan(ctx, n_p, num_samples, nx, ny, k)
    return plan.focus(raw_t, pos_plat, vel_plat, t_start, vel_focus, t_pulses, scene_size, d_image=d_image)
