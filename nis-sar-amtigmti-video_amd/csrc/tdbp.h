// Time-domain back-projection (sar_batch_sim.py:171-238); see tdbp.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/sarx.h"

namespace sarx {
struct Tdbp;
Tdbp* tdbp_create(int n_pulses, int num_samples, int nx, int ny, const sarx_tdbp_params* k, const float2* tw_all, std::string& err);
void tdbp_destroy(Tdbp* t);
hipError_t tdbp_focus(Tdbp* t, const float2* raw_dev, const double* pos, const double* vel, const double* t_pulses, double t_start,
                      const double* vel_focus, double scene_size, bool all_samples, hipStream_t st);
void tdbp_window(const Tdbp* t, int* lo, int* hi);   // samples range-compressed by the last call
const double2* tdbp_image(const Tdbp* t);     // device, complex128 [ny][nx]
const float2* tdbp_rc(const Tdbp* t);         // device, complex64 [n_pulses][num_samples]
uint64_t tdbp_bytes(const Tdbp* t);
}  // namespace sarx
