"""sarx: MI355X-native backend for the CSA focus + ATI/DPCA path.

Python host code with the reference's function signatures over a ctypes C ABI
(include/sarx.h) to hand-written HIP kernels (csrc/).  No torch on this path.
"""
from ._ffi import SarxError
from .engine import Context, CsaPlan, DeviceArray, DeviceBuffer, FocusLanes, default_context, device_count
from .echo import run_bistatic_physics_gpu, run_custom_physics, run_moving_physics, run_physics_engine
from .rda import sar_focus_rda
from .noise import calculate_snr_db, add_ocean_noise, add_noise_dev, add_noise_rel_dev, power_stats
from .tdbp import (tdbp_gpu, run_physics_spotlight, calculate_raw_snr_db, generate_noise_tensor, batch_constants,
                   orbit_arc, TdbpPlan)
from .focus import (FocusFuture, ati_dpca, clear_plan_cache, dpca_pulse_shift, focus_ati_dpca, focus_stream, phase_balance,
                    sar_focus_csa, sar_focus_csa_async, two_channel_workspace)

__all__ = ["SarxError", "Context", "CsaPlan", "FocusLanes", "add_noise_rel_dev", "DeviceArray", "DeviceBuffer", "default_context", "device_count", "sar_focus_csa", "sar_focus_csa_async", "focus_stream", "FocusFuture", "ati_dpca",
           "dpca_pulse_shift", "phase_balance", "focus_ati_dpca", "two_channel_workspace", "clear_plan_cache", "run_physics_engine",
           "run_bistatic_physics_gpu", "run_moving_physics", "run_custom_physics", "sar_focus_rda", "calculate_snr_db", "add_ocean_noise", "add_noise_dev", "power_stats", "tdbp_gpu", "run_physics_spotlight", "calculate_raw_snr_db",
           "generate_noise_tensor", "batch_constants", "orbit_arc", "TdbpPlan"]
