// Any-size CSA focus (chirp-z over the power-of-two kernels); see general.hip.
#pragma once
#include <complex>
#include <string>
#include <vector>

#include "../../include/sarx.h"
#include "csa_kernels.h"

namespace sarx {
struct GeneralCsa;
// csa_tables: also build the per-row range-convolution spectra the CSA path uses when n_rg is not a power of two
// cus: compute units of the device (persistent range kernels size their grid from it)
GeneralCsa* general_csa_create(int n_az, int n_rg, const sarx_radar_params* prm, const float2* tw_all, std::string& err,
                               bool csa_tables = true, int cus = 0);
void general_csa_destroy(GeneralCsa* g);
hipError_t general_csa_focus(GeneralCsa* g, const float2* d_in, float2* d_out, hipStream_t st);
// max |image| slot ([MAX_SHARDS x 32] floats, AzArgs::max_out) filled by every later focus; false = this plan's route has no such epilogue
bool general_csa_set_max_slot(GeneralCsa* g, unsigned* slot);
// ATI / DPCA products out of the last inverse launch of every later focus (ati->part must hold the returned number of partials);
// ati NULL or ati->s1 NULL = off (returns 0); -1 = this plan's route has no such epilogue
struct AtiFuse;
int general_csa_set_ati(GeneralCsa* g, const AtiFuse* ati);
int general_csa_ati_parts(const GeneralCsa* g);
uint64_t general_csa_bytes(const GeneralCsa* g);
// one range pass (RangeMode) on a dense [n_az x n_rg] image; hipErrorNotSupported unless the range extent has a direct
// mixed-radix line kernel (range_mixed.hip)
hipError_t general_csa_range_pass(GeneralCsa* g, int mode, const float2* in, float2* out, hipStream_t st);
// one azimuth pass (forward + Phi_1 / inverse) on dense images; hipErrorNotSupported unless n_az = 7199 on the direct route
hipError_t general_csa_az_pass(GeneralCsa* g, bool inv, const float2* in, float2* out, hipStream_t st);

// building blocks shared with tdbp.hip
// in-place line FFTs of `rows` contiguous lines of length m (power of two, 16..32768); the inverse carries 1/m.
// m == 32768 leaves the spectrum in the split order of to_split_order(); the inverse expects that order.
// mulvec (forward only, optional): the spectrum is multiplied by mulvec[k] (device order) in the transform's epilogue
hipError_t line_fft_pow2(const float2* tw_all, float2* buf, int rows, int m, bool inv, hipStream_t st,
                         const float2* mulvec = nullptr);
void host_fft_pow2(std::vector<std::complex<double>>& a);         // forward, in place
void to_split_order(std::vector<std::complex<double>>& a);        // 32768-point spectrum: natural -> device order
// out[r][c] = (r < in_rows && c < in_cols ? in[r][c] : 0) * colvec[c] * scalar   (colvec optional)
hipError_t scale_copy_cols(const float2* in, int in_rows, int in_cols, size_t in_ld, float2* out, int out_rows, int out_cols,
                           size_t out_ld, const float2* colvec, float scalar, hipStream_t st);

// Range-Doppler focuser (sar_satellite_sim.py:356-448); params.range_ref_m carries range_grp_m
struct Rda;
Rda* rda_create(int n_ranges, int n_pulses, const sarx_radar_params* prm, const float2* tw_all, std::string& err, int cus = 0);
void rda_destroy(Rda* r);
// mag_out: device buffer for the [n_pulses x n_ranges] magnitude (nullptr: the object's own, see rda_mag);
// want_rc: also keep the RCMC map (rda_stage(r, 2)); the other two intermediates are pipeline buffers and always valid
// want_ac: also keep the azimuth-compressed map (rda_stage(r, 3); sar_vehicle_sim.py:268 range_doppler_filtered)
hipError_t rda_focus(Rda* r, const float2* d_in_pulse_major, hipStream_t st, float* mag_out = nullptr, bool want_rc = true,
                     bool want_ac = false);
const float* rda_mag(const Rda* r);                 // [n_pulses x n_ranges] = the reference's sar_image_mag.T
const float2* rda_stage(const Rda* r, int which);   // 0 range-compressed, 1 range-Doppler, 2 after RCMC, 3 after azimuth compression (want_ac); [n_pulses x n_ranges]
void rda_axes(const Rda* r, double* range_centered, double* cross_range, double* doppler);
uint64_t rda_bytes(const Rda* r);
}  // namespace sarx
