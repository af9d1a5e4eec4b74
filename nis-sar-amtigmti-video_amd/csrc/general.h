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
}  // namespace sarx
