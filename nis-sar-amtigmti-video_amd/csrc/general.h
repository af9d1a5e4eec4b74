// Any-size CSA focus (chirp-z over the power-of-two kernels); see general.hip.
#pragma once
#include <string>

#include "../../include/sarx.h"
#include "csa_kernels.h"

namespace sarx {
struct GeneralCsa;
GeneralCsa* general_csa_create(int n_az, int n_rg, const sarx_radar_params* prm, const float2* tw_all, std::string& err);
void general_csa_destroy(GeneralCsa* g);
hipError_t general_csa_focus(GeneralCsa* g, const float2* d_in, float2* d_out, hipStream_t st);
uint64_t general_csa_bytes(const GeneralCsa* g);

// Range-Doppler focuser (sar_satellite_sim.py:356-448); params.range_ref_m carries range_grp_m
struct Rda;
Rda* rda_create(int n_ranges, int n_pulses, const sarx_radar_params* prm, const float2* tw_all, std::string& err);
void rda_destroy(Rda* r);
hipError_t rda_focus(Rda* r, const float2* d_in_pulse_major, hipStream_t st);
const float* rda_mag(const Rda* r);                 // [n_pulses x n_ranges] = the reference's sar_image_mag.T
const float2* rda_stage(const Rda* r, int which);   // 0 range-compressed, 1 range-Doppler, 2 after RCMC; [n_pulses x n_ranges]
void rda_axes(const Rda* r, double* range_centered, double* cross_range, double* doppler);
uint64_t rda_bytes(const Rda* r);
}  // namespace sarx
