// Internal launch interface between the C ABI (sarx_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace sarx {

enum RangeMode { RG_FFT = 0, RG_IFFT = 1, RG_FFT_PHI2 = 2, RG_IFFT_PHI3 = 3, RG_FUSED = 4,
                 RG_CONV = 5 };   // range_mixed.hip: zero-padded FFT . mulvec . IFFT / N, cropped (a circular convolution that holds a 'same' convolution)
enum AzEpilogue { AZ_EPI_NONE = 0, AZ_EPI_TWIDDLE = 1, AZ_EPI_PHI1 = 2, AZ_EPI_SCALE = 3,
                  AZ_EPI_TWCOL = 4,      // * W_M^(col*m_out), M = 1/tw_scale (32768-point line split, forward)
                  AZ_EPI_PROCOL = 5,     // inputs * W_M^(col*m_in) first, outputs * scale (its inverse)
                  AZ_EPI_ROWVEC = 6,     // * rowvec[output row] (Bluestein filter spectrum fused into the forward transform)
                  AZ_EPI_TWIDDLE_PADIN = 7,   // inputs from a smaller [io_rows x io_cols] array (ld io_ld), zero outside, * rowvec[input row]; then TWIDDLE
                  AZ_EPI_CROPOUT = 8,
                  AZ_EPI_TWIDDLE_ROWSIN = 9,  // TWIDDLE, but input rows >= io_rows are zeros that are not read (chirp-z padding)
                  AZ_EPI_SCALE_ROWSOUT = 10,
                  AZ_EPI_CROPOUT_PHI1 = 11,
                  AZ_EPI_CROPOUT_MAG = 12,
                  AZ_EPI_SCALE_LOOK = 13,
                  AZ_EPI_SCALE_ATI = 14 };    // SCALE, and the ATI / DPCA products of (ati_s1, this image) go out instead of (or beside) the image     // SCALE, and the row's sums of |x|^2 over `look` consecutive columns go to look_part (VideoSAR stack slot fused into the focus)    // CROPOUT, but the magnitude goes to out_mag (fp32) instead of the complex value to out   // CROPOUT, then * Phi_1(output row, col): the forward azimuth chirp-z ends in the CSA's first phase  // SCALE, but output rows >= io_rows are not written (only the cropped part is used)       // outputs * rowvec[output row] * scale, written to a [io_rows x io_cols] array (ld io_ld) only inside it

struct RangeArgs {
    const float2* in;
    float2* out;
    const float2* tw;     // exp(-2 pi i m / n_rg), m < n_rg
    const double2* c2;    // per azimuth bin: {0.5/(Kr(1+Cs)), 2 R_ref Cs / c}
    const double2* c3;    // per azimuth bin: {c D / lam, -0.5 Kr Cs (1+Cs)}
    double df;            // 1/(n_rg*dt): fftfreq step (numpy.fft.fftfreq)
    double dt;            // 1/fs
    double t_start;       // t_start_fast
    double t0;            // 2 R_ref / c
    float inv_n;          // 1/n_rg
    int n_az;
    const float2* mulvec; // RG_FFT only, optional: out[line][k] *= mulvec[(line % mul_period) * n_rg + k]
    int mul_period;
    // line -> image row.  row_inner == 0: the launch covers rows 0 .. n_az-1.  Otherwise line i of the n_az lines of this
    // launch is row  row0 + i % row_inner + (i / row_inner) * row_stride  (the rows of a group of azimuth tiles: slab mode)
    int row0, row_inner, row_stride;
    // RG_CONV: the input line holds conv_valid samples (leading dimension conv_in_ld), the rest of the transform length is zero;
    // of the result the samples [conv_crop0, conv_crop0 + conv_out) go to the output line (leading dimension conv_out_ld)
    int conv_valid, conv_crop0, conv_out;
    size_t conv_in_ld, conv_out_ld;
    // RG_CONV on the power-of-two line kernel only: conv_wrap_n > 0 reads input sample j of a line from column (conv_wrap_c0 + j) mod
    // conv_wrap_n of the dense line (a wrapped segment: the overlap-save blocks of a circular correlation, tdbp.hip)
    int conv_wrap_n, conv_wrap_c0;
    // optional {first workgroup start, last workgroup end} of this launch in s_memrealtime ticks (100 MHz), reduced with
    // atomic min / max by one lane per workgroup (sarx_csa_plan_stamp_range; range_fused_wl_kernel only): the launch's
    // execution span while other launches share the GPU, which an event pair on the stream cannot separate from queueing
    unsigned long long* stamp;
};
__host__ __device__ inline int range_row(const RangeArgs& a, int line) {
    return a.row_inner ? a.row0 + line % a.row_inner + (line / a.row_inner) * a.row_stride : line;
}

// max |image| slot of sarx_csa_plan_set_max_slot: MAX_SHARDS partial maxima, one per 128-byte line (32 floats apart)
constexpr unsigned MAX_SHARDS = 256;
constexpr size_t MAX_SLOT_BYTES = MAX_SHARDS * 32 * sizeof(float);
struct AzArgs {
    const float2* in;
    float2* out;
    const float2* tw_r;   // exp(-2 pi i m / R), m < R      (in-tile FFT)
    const float2* tw_n;   // exp(-2 pi i m / n_az), m < n_az (four-step twiddle)
    const double2* c1;    // per azimuth bin: {-0.5 Kr Cs, tau_ref}
    const float2* rowvec; // AZ_EPI_ROWVEC / CROPOUT: per output row; TWIDDLE_PADIN: per input row (optional)
    size_t io_ld;         // TWIDDLE_PADIN / CROPOUT: leading dimension and extents of the smaller array
    int io_rows, io_cols;
    int io_shift_in;      // TWIDDLE_PADIN: sequence element r is source row (r + io_shift_in) mod io_rows (fftshift bookkeeping)
    float hamming_inv;    // TWIDDLE_PADIN, > 0: source row i is weighted 0.54 - 0.46 cos(2 pi i hamming_inv), hamming_inv = 1/(io_rows - 1):
                          // the azimuth window of sar_focus_rda (:396) evaluated in the kernel - as a rowvec table it was a second
                          // vector-memory instruction per row in a launch bound by their issue (0.316 against 0.214 ms at 32768 x 2048)
    int io_shift_out;     // CROPOUT*: sequence element r goes to destination row (r + io_shift_out) mod io_rows
    float* out_mag;       // CROPOUT_MAG: [io_rows x io_cols] fp32 magnitudes (leading dimension io_ld)
    bool nt;              // nontemporal image loads / stores (images too large to be re-read from cache before they are evicted)
    float* look_part;     // SCALE_LOOK: [n_az x n_rg/look] row-wise partial sums of |x|^2 over `look` columns (a power of two <= tile width)
    int look;
    // SCALE_ATI: slc1 = ati_s1 [n_az x n_rg], slc2 = this launch's output; masked phase, |slc1|, |slc1 - slc2 e^(i cal)| planes; partial sums of
    // slc1 conj(slc2) per wave ([tiles x waves], fixed order: reproducible); threshold ati_frac * max over the MAX_SHARDS shards of ati_thr
    const float2* ati_s1; float *ati_phase, *ati_m1, *ati_dm; double2* ati_part; const float* ati_thr;
    float ati_cc, ati_cs, ati_frac; int ati_keep_image;
    unsigned* max_out;    // SCALE / SCALE_LOOK: [MAX_SHARDS x 32] bits of partial maxima of |x| over the image as written (hypotf; non-negative floats order like their bits); NULL = off
    int valid_len;        // TWCOL / PROCOL (split lines): only the first valid_len samples of a line are read (rest = 0) / written; 0 = all
    double dt, t_start;
    float scale;          // 1/n_az for the inverse's last step
    float tw_scale;       // 1/M for the column-indexed twiddles of the 32768-point line split
    int n_rg;
    // row of tile element m for tile q: q*q_stride + m*m_stride;  q = blockIdx.y + q0 (a launch may cover a range of tiles)
    int in_q_stride, in_m_stride, out_q_stride, out_m_stride;
    int q0;
};

// Grid of a persistent kernel: wgs_per_cu resident workgroups on each of `cus` compute units, never more workgroups
// than work items, at least one.  cus comes from the ctx of the device being launched on (never a process-wide cache).
inline int persistent_grid(int wgs_per_cu, int cus, int work_items) {
    if (wgs_per_cu < 1) wgs_per_cu = 1;
    if (cus < 1) cus = 1;
    long long g = (long long)wgs_per_cu * cus;
    if (g > work_items) g = work_items;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_range_pass(int n_rg, int mode, const RangeArgs& a, hipStream_t st);
// range_v2.hip: 32 points/thread, split re/im exchange (two lines resident per CU at 16384)
bool range_v2_supported(int n_rg, int mode);
hipError_t launch_range_pass_v2(int n_rg, int mode, const RangeArgs& a, int cus, hipStream_t st);
// range_fused_wl.hip: fused FFT.Phi2.IFFT.Phi3 at 16384 with wave-private sub-transforms
bool range_fused_wl_supported(int n_rg);
// alone: the launch has the chip to itself (no frames in flight on other lanes): next-line touch prefetch on
hipError_t launch_range_fused_wl(const RangeArgs& a, int cus, hipStream_t st, bool alone = false);
// range_mixed.hip: direct mixed-radix lines (13200 = 24 * 22 * 25, the reference's native range extent), all modes
bool range_mixed_supported(int n_rg);
hipError_t launch_range_mixed(int n_rg, int mode, const RangeArgs& a, int cus, hipStream_t st);
bool range_conv_supported(int m);
hipError_t launch_range_conv(int m, const RangeArgs& a, int cus, hipStream_t st);
// r: FFT length of the tile (2..128), w: tile width in range samples (16 or 32), nq: tiles along azimuth
hipError_t launch_az_tile(int r, int w, bool inv, int epi, const AzArgs& a, int nq, hipStream_t st);

// Middle launch of a two-step chirp-z column transform of length M = ra * s (general.hip): tile q holds rows q*s + m
// (m < s) of a [M x n_rg] array; FFT_s over m gives bins q + ra*k2, times bhat[q + ra*k2] (a.rowvec, natural order),
// inverse FFT_s, times the conjugate four-step twiddle W_M^(+q m) (a.tw_scale = 1/M), in place.  The forward
// transform's second step and the inverse transform's first step on one HBM round trip.
hipError_t launch_az_conv(int s, int ra, const AzArgs& a, hipStream_t st);

// az_pfa.hip: prime-factor (23 x 313) azimuth transforms of the reference's native 7199 pulses, Rader for the 313
// ATI / DPCA products emitted by a focus's last azimuth launch (sarx_csa_plan_set_ati): what that launch needs
struct AtiFuse {
    const float2* s1;                 // first channel's image [n_az x n_rg] (dense); NULL = off
    float *phase, *m1, *dm;           // masked ATI phase, |slc1|, DPCA magnitude planes [n_az x n_rg]
    double2* part;                    // one partial of sum slc1 conj(slc2) per wave of the launch
    const float* thr;                 // MAX_SHARDS partial maxima of |slc1| (32 floats apart)
    float cc, cs, frac;               // exp(i cal), mask fraction
    int keep_image;
};
struct PfaArgs {
    const float2* in; size_t in_ld; int in_cols;      // dense source [7199 x in_cols]
    float2* u; size_t u_ld; int u_cols;                // intermediate, rows n1*313 + k2
    float2* out; size_t out_ld; int out_cols;         // dense destination, rows in natural bin / pulse order
    const unsigned* offin; const unsigned* offu;       // [312] byte offsets of 23 g^q source rows / g^-m intermediate rows
    unsigned off0in, off0u;                            // bytes of 313 rows of the source / intermediate
    unsigned pitch_u, pitch_out;                       // bytes of one row of the intermediate / destination
    const float2* bspec;                               // spectrum of the Rader kernel w' / 312, for the transform's sign
    const double2* c1; double dt, t_start;             // Phi_1 epilogue
    float scale;                                       // inverse: 1/7199
    int c1k, c2k;                                      // output map k = (c1k k1 + c2k k2) mod 7199
    bool nt;                                           // nontemporal image loads / stores
    unsigned* max_out;                                 // inverse epilogue: [MAX_SHARDS x 32] partial maxima of |out| (AzArgs::max_out); NULL = off
    AtiFuse ati;                                       // inverse epilogue 3: the products of (ati.s1, this image)
};
struct AzPfa;
bool az_pfa_supported(int n_az);
// tables are built for fixed leading dimensions (elements) of the source, the intermediate and the destination
AzPfa* az_pfa_create(size_t in_ld, size_t u_ld, size_t out_ld, hipError_t* err);
void az_pfa_destroy(AzPfa* z);
// epi: 0 none, 1 times Phi_1 (forward), 2 times scale (inverse).  src may equal dst; u is a [7199 x u_ld] work array
hipError_t az_pfa_run(const AzPfa* z, bool inv, const float2* src, size_t src_ld, int src_cols, float2* u, size_t u_ld,
                      float2* dst, size_t dst_ld, int dst_cols, int epi, const double2* c1, double dt, double t_start,
                      float scale, hipStream_t st, unsigned* max_out = nullptr, const AtiFuse* ati = nullptr);
// partial sums the products epilogue of az_pfa_run writes for an image of dst_cols columns (one per wave)
int az_pfa_ati_parts(int dst_cols);

// az_pfa7200.hip: prime-factor (32 x 225) pulse-axis transforms of the Range-Doppler focuser at 7200 pulses
bool az_pfa7200_supported(int n);
hipError_t az_pfa7200_run(bool inv, const float2* src, size_t src_ld, int cols, float2* u, size_t u_ld, float2* dst, float* dst_mag,
                          size_t dst_ld, int shift_in, int shift_out, const float* pre, float scale, hipStream_t st);
// range_wp.hip: sixteen-wave range passes at 16384 samples with the spectrum in permuted order between FFT+Phi2 and IFFT+Phi3
bool range_wp_supported(int n_rg);
hipError_t launch_range_wp(int mode, const RangeArgs& a, int cus, hipStream_t st);

// products.hip
struct AtiArgs {
    const float2* s1;
    const float2* s2;
    size_t n;
    float cal_c, cal_s;   // exp(i*cal_phase)
    float* ati_phase;
    float* mag1;
    float* dpca_mag;
    float2* interf;
    float2* diff;
    float* mag2;
    float* ph1;
    float* ph2;
    float* dpca_phase;
    bool nt;              // nontemporal loads of the two images (their last use)
    const float* thr_max; // masked variant: ati_phase gets phase where |slc1| > mask_frac * max(thr_max shards), 0 elsewhere (:447-449); NULL = plain
    float mask_frac;
    float* part_max;      // [blocks]
    double2* part_sum;    // [blocks]
};
int ati_blocks(size_t n);
hipError_t launch_ati_dpca(const AtiArgs& a, hipStream_t st);
hipError_t launch_ati_finish(const float* part_max, const double2* part_sum, int blocks, double* out3, hipStream_t st);
// scratch: 128 double2 (the first level's results)
hipError_t launch_ati_finish_sums(const double2* part_sum, int n, const float* max_shards, double2* scratch, double* out3, hipStream_t st);
hipError_t launch_mask_phase(const float* phase, const float* mag, size_t n, float thr, float* out, hipStream_t st);
// thr = frac * out3[0] on the device (out3 = {max|slc1|, sum re, sum im} of the last ATI launch)
hipError_t launch_mask_phase_frac(const float* phase, const float* mag, size_t n, float frac, const double* out3, float* out,
                                  hipStream_t st);
hipError_t launch_magnitude(const float2* in, float* out, size_t n, hipStream_t st);
hipError_t launch_spin(int blocks, unsigned long long cycles, unsigned* sink, hipStream_t st);
hipError_t launch_max_abs_f32(const float* x, size_t n, float* out, hipStream_t st);   // *out = max(*out, max |x|)
hipError_t launch_corner_turn(const float2* in, float2* out, int rows, int cols, hipStream_t st);
hipError_t launch_multilook(const float2* in, float* out, int rows, int cols, int looks, hipStream_t st);
// out[R][C] = (sum over r < looks of part[(looks R + r) * cols + C]) / looks^2: second half of the multilook fused into the focus
hipError_t launch_look_finish(const float* part, float* out, int out_rows, int cols, int looks, hipStream_t st);
hipError_t launch_fill_noise(float2* buf, size_t n, uint64_t seed, hipStream_t st);

hipError_t launch_ocean_noise(float2* buf, size_t n, float sigma, float clutter_power, float nu, uint64_t seed, hipStream_t st,
                              const float* levels = nullptr);
hipError_t launch_noise_levels(const double* part, int blocks, size_t n, int ref_is_max, double snr_lin, double scr_lin, float* levels,
                               hipStream_t st);
hipError_t launch_power_stats(const float2* buf, size_t n, double* part, int blocks, hipStream_t st);

// echo.hip
struct EchoArgs {
    const double2* tau_pb;   // [n_pulses][n_targets] {delay s, carrier phase in revolutions}
    const float* amp;        // [n_targets] sqrt(rcs)
    const float* amp_pt;     // optional [n_pulses][n_targets]: per-pulse amplitude (antenna pattern), replaces amp
    const double* t_fast;    // [n_samples] absolute fast time of each sample
    float2* out;             // [n_pulses][n_samples]
    double kr, t_p, u_off;   // u = t_fast - tau - u_off
    int n_pulses, n_targets, n_samples;
    int accumulate;          // out += the sum (a second target set into the same pulses) instead of out =
};
hipError_t launch_echo_synth(const EchoArgs& a, hipStream_t st);
struct EchoGeoArgs {
    int model;                // 0 monostatic, 1 bistatic, 2 spotlight
    int n_pulses, n_targets;
    const double* tgt_pos;    // [n_targets][3] at t = 0
    const double* tgt_vel;    // [3] common target velocity (models 1, 2)
    const double* t_pulse;    // [n_pulses] slow time (models 1, 2)
    const double* tx_pos;     // [n_pulses][3] transmitter / platform position
    const double* aux;        // [n_pulses][3]: receiver position (model 1) or platform velocity (model 2)
    const double* rcs;        // [n_targets] (model 2)
    double c, fc, l_ant, lambda;
    double2* tau_pb;          // out [n_pulses][n_targets]
    float* amp_pt;            // out [n_pulses][n_targets] (model 2)
};
hipError_t launch_echo_geometry(const EchoGeoArgs& a, hipStream_t st);

}  // namespace sarx
