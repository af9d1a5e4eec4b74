// CSA focus for sizes that are not powers of two (the reference's native scene is
// 7199 pulses x 13200 samples, sar_ati_dcpa_sim_csa.py:47,111,402).
//
// Any-length DFTs are Bluestein chirp-z transforms over the power-of-two kernels:
//   X[k] = c[k] * sum_n (x[n] c[n]) conj(c)[k-n],   c[n] = exp(-i pi n^2 / N),
// i.e. pad to M >= 2N-1 (power of two), FFT_M, multiply by the precomputed spectrum of conj(c),
// IFFT_M, crop.  Chirp and kernel tables are evaluated in fp64 on the host (n^2 reduced mod 2N in
// integers).  Along range M reaches 32768 (N = 13200), one size beyond the single-launch line FFT:
// a 32768-point line runs as 128 x 256 over the azimuth tile kernel and the 256-point line kernel,
// its spectrum staying in the permuted order the inverse consumes.
// This path is correctness-first: phases are separate element-wise launches and every transform
// makes several HBM round trips.  The power-of-two path (sarx_api.hip) is the tuned one.
#include "general.h"

#include <cmath>
#include <complex>
#include <cstdlib>
#include <vector>

#include "fft_core.hpp"
#include "phase.hpp"

namespace sarx {

// ---- element-wise kernels --------------------------------------------------------------------------
// out[r][c] = (r < in_rows && c < in_cols ? in[s][c] : 0) * rowvec[r] * rowvec_src[s] * colvec[c] * scalar,
// s = (r + row_shift) mod in_rows  (a circular shift of the rows: fftshift / ifftshift bookkeeping of the RDA)
// (in may equal out when row_shift == 0: each thread reads and writes the same element)
__global__ __launch_bounds__(256) void scale_copy_2d_kernel(const cf* in, int in_rows, int in_cols, size_t in_ld,
                                                            cf* out, int out_rows, int out_cols, size_t out_ld,
                                                            const cf* __restrict__ rowvec, const cf* __restrict__ colvec,
                                                            float scalar, int row_shift, const cf* __restrict__ rowvec_src) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= out_cols) return;
    const bool cin = c < in_cols;
    cf cv = make_float2(scalar, 0.f);
    if (colvec && cin) { const cf w = colvec[c]; cv = make_float2(w.x * scalar, w.y * scalar); }
    for (int r = blockIdx.y; r < out_rows; r += gridDim.y) {
        cf x = make_float2(0.f, 0.f);
        if (cin && r < in_rows) {
            int s = r + row_shift;
            if (s >= in_rows) s -= in_rows;
            x = cmul(in[(size_t)s * in_ld + c], cv);
            if (rowvec) x = cmul(x, rowvec[r]);
            if (rowvec_src) x = cmul(x, rowvec_src[s]);
        }
        out[(size_t)r * out_ld + c] = x;
    }
}

static hipError_t scale_copy(const cf* in, int in_rows, int in_cols, size_t in_ld, cf* out, int out_rows, int out_cols,
                             size_t out_ld, const cf* rowvec, const cf* colvec, float scalar, hipStream_t st,
                             int row_shift = 0, const cf* rowvec_src = nullptr) {
    dim3 grid((out_cols + 255) / 256, out_rows < 16384 ? out_rows : 16384);
    hipLaunchKernelGGL(scale_copy_2d_kernel, grid, dim3(256), 0, st, in, in_rows, in_cols, in_ld, out, out_rows, out_cols,
                       out_ld, rowvec, colvec, scalar, row_shift, rowvec_src);
    return hipGetLastError();
}

// ---- host-side tables ----------------------------------------------------------------------------
typedef std::complex<double> zd;
static void host_fft(std::vector<zd>& a) {           // iterative radix-2, forward, in place
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const zd w = std::polar(1.0, ang * (double)k);
                const zd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}
static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
// forward DFT of any length on the host, fp64: Bluestein's chirp-z over the radix-2 transform above
static void host_dft_any(std::vector<zd>& a) {
    const size_t n = a.size();
    if (n && (n & (n - 1)) == 0) { host_fft(a); return; }
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    std::vector<zd> c(n), x(m, zd(0, 0)), b(m, zd(0, 0));
    for (size_t k = 0; k < n; ++k) {
        const unsigned long long k2 = ((unsigned long long)k * k) % (2ull * n);
        c[k] = std::polar(1.0, -M_PI * (double)k2 / (double)n);
        x[k] = a[k] * c[k];
        b[k] = std::conj(c[k]);
        if (k) b[m - k] = std::conj(c[k]);
    }
    host_fft(x);
    host_fft(b);
    for (size_t i = 0; i < m; ++i) x[i] = std::conj(x[i] * b[i]);      // inverse transform = conj(FFT(conj(.))) / m
    host_fft(x);
    for (size_t k = 0; k < n; ++k) a[k] = std::conj(x[k]) / (double)m * c[k];
}

struct Axis {
    int n = 0, m = 0;            // length, convolution length (m == n: direct power of two)
    bool direct = false;
    cf *chirp_f = nullptr, *chirp_i = nullptr, *bhat_f = nullptr, *bhat_i = nullptr;   // device
};
static const int SPLIT_B = 256;                       // lines beyond 16384 points: 32768 = 128 x 256, 65536 = 256 x 256
static inline int split_a(int m) { return m / SPLIT_B; }

static hipError_t upload(const std::vector<zd>& h, cf** d) {
    std::vector<cf> f(h.size());
    for (size_t i = 0; i < h.size(); ++i) f[i] = make_float2((float)h[i].real(), (float)h[i].imag());
    hipError_t e = hipMalloc(d, f.size() * sizeof(cf));
    if (e != hipSuccess) return e;
    return hipMemcpy(*d, f.data(), f.size() * sizeof(cf), hipMemcpyHostToDevice);
}
// direct_max: largest power of two transformed without chirp-z; split_order: a 32768-point spectrum is stored in the
// order of the split line FFT (range axis), the column transforms (azimuth axis) keep natural order
static hipError_t axis_init(Axis& ax, int n, int m_max, int direct_max, bool split_order) {
    ax.n = n;
    if (is_pow2(n) && n >= 16 && n <= direct_max) { ax.direct = true; ax.m = n; return hipSuccess; }
    int m = 16;
    while (m < 2 * n - 1) m <<= 1;
    if (m > m_max) return hipErrorInvalidValue;
    ax.m = m;
    std::vector<zd> cf_(n), ci_(n), b(m, zd(0, 0)), bi(m, zd(0, 0));
    for (int k = 0; k < n; ++k) {
        const long long k2 = ((long long)k * k) % (2LL * n);          // exact reduction of k^2 mod 2N
        const double ang = M_PI * (double)k2 / (double)n;
        cf_[k] = std::polar(1.0, -ang);
        ci_[k] = std::polar(1.0, ang);
        b[k] = ci_[k];                 // conj(c) for the forward transform
        bi[k] = cf_[k];
        if (k) { b[m - k] = ci_[k]; bi[m - k] = cf_[k]; }
    }
    host_fft(b);
    host_fft(bi);
    if (m > 16384 && split_order) {    // spectrum order of the split line FFT: position k1*256 + k2 holds bin k1 + (m/256)*k2
        to_split_order(b);
        to_split_order(bi);
    }
    hipError_t e;
    if ((e = upload(cf_, &ax.chirp_f)) != hipSuccess) return e;
    if ((e = upload(ci_, &ax.chirp_i)) != hipSuccess) return e;
    if ((e = upload(b, &ax.bhat_f)) != hipSuccess) return e;
    return upload(bi, &ax.bhat_i);
}
static void axis_free(Axis& ax) { hipFree(ax.chirp_f); hipFree(ax.chirp_i); hipFree(ax.bhat_f); hipFree(ax.bhat_i); }

struct GeneralCsa {
    AtiFuse ati{};                    // general_csa_set_ati: products out of the last inverse launch (prime-factor route)
    unsigned* max_slot = nullptr;     // general_csa_set_max_slot: partial maxima of |image| from the last inverse launch (prime-factor route)
    int n_az = 0, n_rg = 0, ldc = 0;
    sarx_radar_params p{};
    const cf* tw_all = nullptr;
    Axis az, rg;
    double2 *c1 = nullptr, *c2 = nullptr, *c3 = nullptr;
    cf *data = nullptr, *work_a = nullptr, *work_b = nullptr;
    size_t work_elems = 0;
    cf* ktab = nullptr;          // [n_az x rg.m]: spectrum of the range convolution kernel IFFT_N(Phi_2) per azimuth bin (see below)
    bool rg_mixed = false;       // the range extent has a direct mixed-radix line kernel (range_mixed.hip): no chirp-z along range
    AzPfa* pfa = nullptr;        // n_az = 7199: prime-factor azimuth transforms (az_pfa.hip) instead of chirp-z, on the rg_mixed route
    int cus = 256;               // compute units of the device (persistent grids)
    uint64_t bytes = 0;
};

// ---- power-of-two transforms on work arrays ---------------------------------------------------------
// mulvec (forward only): the m-point spectrum is multiplied by mulvec[k] (device order beyond 16384) on the way out;
// mul_rows > 0: mulvec is a [mul_rows x m] table, one spectrum per line (line r uses row r mod mul_rows)
// valid_len > 0 (m > 16384 only): forward reads only the first valid_len samples of each line (the rest are zeros that
// need not exist in memory), inverse writes only those
static hipError_t rows_pow2(GeneralCsa* g, cf* buf, int rows, int m, bool inv, hipStream_t st, const cf* mulvec = nullptr,
                            int mul_rows = 0, int valid_len = 0) {
    RangeArgs a{};
    hipError_t e;
    if (m <= 16384) {
        a.in = buf; a.out = buf; a.tw = g->tw_all + m; a.inv_n = 1.0f / (float)m; a.n_az = rows;
        a.mulvec = inv ? nullptr : mulvec; a.mul_period = mul_rows > 0 ? mul_rows : 1;
        return launch_range_pass(m, inv ? RG_IFFT : RG_FFT, a, st);
    }
    // 32768 = 128 x 256 (65536 = 256 x 256) on the [(rows*SA) x 256] view of the lines
    const int SPLIT_A = split_a(m);
    AzArgs z{};
    z.in = buf; z.out = buf; z.tw_r = g->tw_all + SPLIT_A; z.n_rg = SPLIT_B; z.tw_scale = 1.0f / (float)m;
    z.scale = 1.0f / (float)SPLIT_A;
    z.in_q_stride = SPLIT_A; z.in_m_stride = 1; z.out_q_stride = SPLIT_A; z.out_m_stride = 1;
    z.valid_len = valid_len;
    a.in = buf; a.out = buf; a.tw = g->tw_all + SPLIT_B; a.inv_n = 1.0f / (float)SPLIT_B; a.n_az = rows * SPLIT_A;
    if (!inv) {
        if ((e = launch_az_tile(SPLIT_A, 32, false, AZ_EPI_TWCOL, z, rows, st)) != hipSuccess) return e;
        a.mulvec = mulvec; a.mul_period = (mul_rows > 0 ? mul_rows : 1) * SPLIT_A;   // line L = row*SA + k1 holds positions k1*256 + k2
        return launch_range_pass(SPLIT_B, RG_FFT, a, st);
    }
    if ((e = launch_range_pass(SPLIT_B, RG_IFFT, a, st)) != hipSuccess) return e;
    return launch_az_tile(SPLIT_A, 32, true, AZ_EPI_PROCOL, z, rows, st);
}

// column FFT of length n (power of two, 16..16384) on a [n x ld] array, ld a multiple of 32; in -> out via tmp
// Optional fused ends of a two-step column transform (n > 128): the first step reads a smaller dense array (zero
// outside it, times a per-row vector) instead of `in`; the last step of an inverse writes times a per-row vector and a
// scale into a smaller dense array instead of `out`.  One HBM pass each instead of a separate copy.
// shift: circular row shift while copying in (sequence element r = source row (r + shift) mod rows; rowvec is indexed by
// r) / out (sequence element r -> destination row (r + shift) mod rows); mag: write |.| there (fp32, leading
// dimension ld) instead of the complex value to p
struct ColsSrc { const cf* p; size_t ld; int rows, cols; const cf* rowvec; int shift; };
struct ColsDst { cf* p; size_t ld; int rows, cols; const cf* rowvec; float scale; int shift; float* mag; };
static bool cols_two_step(int n) { return n > 128; }

static hipError_t cols_pow2(GeneralCsa* g, const cf* in, cf* tmp, cf* out, int n, int ld, bool inv, hipStream_t st,
                            const cf* rowvec = nullptr, const ColsSrc* src = nullptr, const ColsDst* dst = nullptr,
                            int rows_valid = 0) {       // forward: input rows beyond are zeros, not read; inverse: output rows beyond not written
    int l2 = 0;
    while ((1 << l2) < n) ++l2;
    const int S = (n <= 128) ? n : (1 << (l2 / 2)), RA = n / S;
    AzArgs a{};
    a.scale = 1.0f / (float)n;
    a.n_rg = ld;
    a.tw_n = n <= 16384 ? g->tw_all + n : nullptr;      // table twiddles exist up to 16384; the kernels use v_sin/v_cos
    const int epi_last = inv ? AZ_EPI_SCALE : (rowvec ? AZ_EPI_ROWVEC : AZ_EPI_NONE);
    a.rowvec = rowvec;
    if (S == n) {
        a.in = in; a.out = out; a.tw_r = g->tw_all + n;
        a.in_q_stride = 0; a.in_m_stride = 1; a.out_q_stride = 0; a.out_m_stride = 1;
        return launch_az_tile(n, 32, inv, epi_last, a, 1, st);
    }
    a.in = in; a.out = tmp; a.tw_r = g->tw_all + RA;
    a.in_q_stride = 1; a.in_m_stride = S; a.out_q_stride = 1; a.out_m_stride = S;
    int epi_first = AZ_EPI_TWIDDLE;
    if (!src && !inv && rows_valid > 0 && rows_valid < n) { epi_first = AZ_EPI_TWIDDLE_ROWSIN; a.io_rows = rows_valid; }
    if (src) {
        if (inv) return hipErrorInvalidValue;
        epi_first = AZ_EPI_TWIDDLE_PADIN;
        a.in = src->p; a.io_ld = src->ld; a.io_rows = src->rows; a.io_cols = src->cols; a.rowvec = src->rowvec;
    }
    hipError_t e = launch_az_tile(RA, 32, inv, epi_first, a, S, st);
    if (e != hipSuccess) return e;
    a.in = tmp; a.out = out; a.tw_r = g->tw_all + S; a.rowvec = rowvec;
    a.in_q_stride = S; a.in_m_stride = 1; a.out_q_stride = 1; a.out_m_stride = RA;
    if (dst) {
        if (!inv) return hipErrorInvalidValue;
        a.out = dst->p; a.io_ld = dst->ld; a.io_rows = dst->rows; a.io_cols = dst->cols; a.rowvec = dst->rowvec;
        a.scale = dst->scale / (float)n;
        return launch_az_tile(S, 32, true, AZ_EPI_CROPOUT, a, RA, st);
    }
    if (!dst && inv && rows_valid > 0 && rows_valid < n) {
        a.io_rows = rows_valid;
        return launch_az_tile(S, 32, true, AZ_EPI_SCALE_ROWSOUT, a, RA, st);
    }
    return launch_az_tile(S, 32, inv, epi_last, a, RA, st);
}

hipError_t line_fft_pow2(const float2* tw_all, float2* buf, int rows, int m, bool inv, hipStream_t st, const float2* mulvec) {
    GeneralCsa g;
    g.tw_all = tw_all;
    return rows_pow2(&g, buf, rows, m, inv, st, mulvec);
}
void host_fft_pow2(std::vector<zd>& a) { host_fft(a); }
void to_split_order(std::vector<zd>& a) {
    std::vector<zd> p(a.size());
    const int SPLIT_A = split_a((int)a.size());
    for (int k1 = 0; k1 < SPLIT_A; ++k1)
        for (int k2 = 0; k2 < SPLIT_B; ++k2) p[k1 * SPLIT_B + k2] = a[k1 + SPLIT_A * k2];
    a.swap(p);
}
hipError_t scale_copy_cols(const float2* in, int in_rows, int in_cols, size_t in_ld, float2* out, int out_rows, int out_cols,
                           size_t out_ld, const float2* colvec, float scalar, hipStream_t st) {
    return scale_copy(in, in_rows, in_cols, in_ld, out, out_rows, out_cols, out_ld, nullptr, colvec, scalar, st);
}

// ---- any-length transforms of the dense [n_az x n_rg] image `d` (in place) ----------------------------------
#define GCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

// in_shift / out_shift: circular row shifts applied while copying in / out (element r of the transformed sequence is
// source row (r + in_shift) mod n; destination row r' receives sequence element (r' + out_shift) mod n);
// pre: optional per-source-row factor (azimuth window)
static hipError_t fft_cols(GeneralCsa* g, const cf* src, cf* dst, bool inv, hipStream_t st, int in_shift = 0,
                           const cf* pre = nullptr, int out_shift = 0) {
    const Axis& ax = g->az;
    const int n = g->n_az, C = g->n_rg, ld = g->ldc, m = ax.m;
    cf *wa = g->work_a, *wb = g->work_b;
    if (ax.direct) {
        GCK(scale_copy(src, n, C, C, wa, n, ld, ld, nullptr, nullptr, 1.0f, st, in_shift, pre));
        GCK(cols_pow2(g, wa, wa, wb, n, ld, inv, st));                     // step A in place on wa, result in wb
        return scale_copy(wb, n, C, ld, dst, n, C, C, nullptr, nullptr, 1.0f, st, out_shift, nullptr);
    }
    const cf* chirp = inv ? ax.chirp_i : ax.chirp_f;
    // rows n .. m of the padded sequence: zeros that are neither written nor read (two-step transforms), unused on the way out
    const int rv = cols_two_step(m) ? n : 0;
    GCK(scale_copy(src, n, C, C, wa, rv ? n : m, ld, ld, chirp, nullptr, 1.0f, st, in_shift, pre));
    GCK(cols_pow2(g, wa, wa, wb, m, ld, false, st, inv ? ax.bhat_i : ax.bhat_f, nullptr, nullptr, rv));   // filter spectrum in the epilogue
    GCK(cols_pow2(g, wb, wb, wa, m, ld, true, st, nullptr, nullptr, nullptr, rv));                          // result in wa
    // the chirp belongs to the sequence index, i.e. to the source row of this copy
    return scale_copy(wa, n, C, ld, dst, n, C, C, nullptr, nullptr, inv ? 1.0f / (float)n : 1.0f, st, out_shift, chirp);
}

// ---- fused hand-over between two transforms ---------------------------------------------------------
// out[r][c] = in[r][c] * row_vec[r] * col_vec[c] * scale * Phi_WHICH(r, c)   for r < n_az, c < n_rg, else 0
// The post-chirp of the Bluestein transform that just finished, the CSA phase, and the pre-chirp / zero padding of
// the next transform in ONE pass (three before).  in may equal out when the leading dimensions agree.
struct BridgeArgs {
    const cf* in; size_t in_ld;
    cf* out; size_t out_ld;
    int n_az, n_rg, out_rows, out_cols;
    const cf *row_vec, *col_vec;
    float scale;
    const double2* c;
    double dt, t_start, t0, df;
};
template <int WHICH> __global__ __launch_bounds__(256) void bridge_kernel(BridgeArgs a) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.out_cols) return;
    const bool cin = j < a.n_rg;
    const double tau = __dadd_rn(a.t_start, __dmul_rn((double)j, a.dt));
    const int ks = (j < (a.n_rg + 1) / 2) ? j : j - a.n_rg;            // numpy.fft.fftfreq order, any parity
    const double f = (double)ks * a.df;
    cf cv = make_float2(a.scale, 0.f);
    if (a.col_vec && cin) { const cf w = a.col_vec[j]; cv = make_float2(w.x * a.scale, w.y * a.scale); }
    for (int i = blockIdx.y; i < a.out_rows; i += gridDim.y) {
        cf x = make_float2(0.f, 0.f);
        if (cin && i < a.n_az) {
            const double2 c = a.c[i];
            double p;
            if (WHICH == 1) { const double d = tau - c.y; p = c.x * d * d; }                          // :272
            else if (WHICH == 2) p = f * fma(c.x, f, c.y);                                            // :318-324
            else { const double d = tau - a.t0; p = fma(c.x, tau, c.y * d * d); }                     // :359,375-380
            x = cmul(a.in ? cmul(a.in[(size_t)i * a.in_ld + j], cv) : cv, cis_rev(p));
            if (a.row_vec) x = cmul(x, a.row_vec[i]);
        }
        a.out[(size_t)i * a.out_ld + j] = x;
    }
}
template <int WHICH>
static hipError_t bridge(GeneralCsa* g, const cf* in, size_t in_ld, cf* out, size_t out_ld, int out_rows, int out_cols,
                         const cf* row_vec, const cf* col_vec, float scale, hipStream_t st) {
    BridgeArgs a{};
    a.in = in; a.in_ld = in_ld; a.out = out; a.out_ld = out_ld;
    a.n_az = g->n_az; a.n_rg = g->n_rg; a.out_rows = out_rows; a.out_cols = out_cols;
    a.row_vec = row_vec; a.col_vec = col_vec; a.scale = scale;
    a.c = (WHICH == 1) ? g->c1 : (WHICH == 2) ? g->c2 : g->c3;
    a.dt = 1.0 / g->p.sample_rate_hz; a.t_start = g->p.t_start_fast_s;
    a.t0 = 2.0 * g->p.range_ref_m / 299792458.0;
    a.df = 1.0 / ((double)g->n_rg * a.dt);
    dim3 grid((out_cols + 255) / 256, out_rows < 16384 ? out_rows : 16384);
    hipLaunchKernelGGL(bridge_kernel<WHICH>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

// column transform of x [m_az x ldc] (already chirped and padded when the axis is not direct); x and y are both
// overwritten; returns the buffer holding the result through *res
// src: the transform's input comes from there (x is then scratch only); dst: an inverse writes its result there (*res = null)
static hipError_t cols_core(GeneralCsa* g, cf* x, cf* y, bool inv, hipStream_t st, cf** res, const ColsSrc* src = nullptr,
                            const ColsDst* dst = nullptr) {
    const Axis& ax = g->az;
    if (ax.direct) { *res = dst ? nullptr : y; return cols_pow2(g, x, x, y, g->n_az, g->ldc, inv, st, nullptr, src, dst); }
    *res = dst ? nullptr : x;
    // rows n_az .. m of the padded sequence are zeros on the way in (never read) and unused on the way out (never written)
    GCK(cols_pow2(g, x, x, y, ax.m, g->ldc, false, st, inv ? ax.bhat_i : ax.bhat_f, src, nullptr, g->n_az));
    return cols_pow2(g, y, y, x, ax.m, g->ldc, true, st, nullptr, nullptr, dst, g->n_az);
}
// Chirp-z column transform of the dense [n_az x n_rg] array src into the dense array dst in THREE launches on one
// [M x ldc] work array x (M = ra * s > 128):  (A) pre-chirp, zero padding and the first four-step stage while copying in,
// (B) second stage . filter spectrum . first inverse stage in place (launch_az_conv), (C) last inverse stage, post-chirp,
// scale and crop while copying out - optionally times Phi_1 (epi_last = AZ_EPI_CROPOUT_PHI1).  The padded rows are
// never read in (A) nor written in (C).  Five launches and two work arrays before.
static hipError_t cols_bluestein3(GeneralCsa* g, cf* x, bool inv, hipStream_t st, const ColsSrc& src, const ColsDst& dst,
                                  int epi_last) {
    const Axis& ax = g->az;
    const int m = ax.m, ld = g->ldc;
    int l2 = 0;
    while ((1 << l2) < m) ++l2;
    const int S = 1 << (l2 / 2), RA = m / S;
    AzArgs a{};
    a.n_rg = ld;
    a.tw_n = m <= 16384 ? g->tw_all + m : nullptr;
    a.c1 = g->c1; a.dt = 1.0 / g->p.sample_rate_hz; a.t_start = g->p.t_start_fast_s;
    // (A) rows q + m S: FFT over m (length RA), twiddle W_M^(-q m'), same rows of x
    a.in = src.p; a.out = x; a.tw_r = g->tw_all + RA; a.scale = 1.0f / (float)m;
    a.io_ld = src.ld; a.io_rows = src.rows; a.io_cols = src.cols; a.rowvec = src.rowvec; a.io_shift_in = src.shift;
    a.in_q_stride = 1; a.in_m_stride = S; a.out_q_stride = 1; a.out_m_stride = S;
    GCK(launch_az_tile(RA, 32, false, AZ_EPI_TWIDDLE_PADIN, a, S, st));
    // (B) rows q S + m in place
    a.in = x; a.out = x; a.tw_r = g->tw_all + S; a.rowvec = inv ? ax.bhat_i : ax.bhat_f; a.tw_scale = 1.0f / (float)m;
    GCK(launch_az_conv(S, RA, a, st));
    // (C) rows q + m' S: inverse FFT over m' (length RA) -> sequence index q + S m, cropped into dst
    a.in = x; a.out = dst.p; a.tw_r = g->tw_all + RA; a.rowvec = dst.rowvec; a.scale = dst.scale / (float)m;
    a.io_ld = dst.ld; a.io_rows = dst.rows; a.io_cols = dst.cols; a.io_shift_in = 0; a.io_shift_out = dst.shift; a.out_mag = dst.mag;
    a.in_q_stride = 1; a.in_m_stride = S; a.out_q_stride = 1; a.out_m_stride = S;
    return launch_az_tile(RA, 32, true, dst.mag ? AZ_EPI_CROPOUT_MAG : epi_last, a, S, st);
}

// Direct two-step column transform of a power-of-two length n > 128 with both ends fused: the first step reads the dense
// array src (rows rotated by src.shift, times src.rowvec by sequence index) - or, src.p == nullptr, the work array x
// itself, already in sequence order -, the last step writes the dense array dst (rows rotated by dst.shift, scaled) or its
// magnitude.  Two launches, x is the only intermediate.
static hipError_t cols_pow2_ends(GeneralCsa* g, cf* x, int n, bool inv, hipStream_t st, const ColsSrc& src, const ColsDst& dst,
                                 float hamming_inv = 0.f) {       // > 0: Hamming weight by SOURCE row, computed in the first step (AzArgs)
    const int ld = g->ldc;
    int l2 = 0;
    while ((1 << l2) < n) ++l2;
    const int S = 1 << (l2 / 2), RA = n / S;
    AzArgs a{};
    a.n_rg = ld;
    a.scale = 1.0f / (float)n;                           // four-step twiddle argument q m / n
    a.tw_n = n <= 16384 ? g->tw_all + n : nullptr;
    a.in = src.p ? src.p : x; a.out = x; a.tw_r = g->tw_all + RA;
    a.in_q_stride = 1; a.in_m_stride = S; a.out_q_stride = 1; a.out_m_stride = S;
    if (src.p) {
        if (inv) return hipErrorInvalidValue;            // the inverse's copy-in form is not instantiated
        a.io_ld = src.ld; a.io_rows = src.rows; a.io_cols = src.cols; a.rowvec = src.rowvec; a.io_shift_in = src.shift;
        a.hamming_inv = hamming_inv;
        GCK(launch_az_tile(RA, 32, false, AZ_EPI_TWIDDLE_PADIN, a, S, st));
        a.hamming_inv = 0.f;
    } else {
        GCK(launch_az_tile(RA, 32, inv, AZ_EPI_TWIDDLE, a, S, st));
    }
    a.in = x; a.out = dst.p; a.tw_r = g->tw_all + S; a.rowvec = dst.rowvec;
    a.in_q_stride = S; a.in_m_stride = 1; a.out_q_stride = 1; a.out_m_stride = RA;
    a.io_ld = dst.ld; a.io_rows = dst.rows; a.io_cols = dst.cols; a.io_shift_in = 0; a.io_shift_out = dst.shift; a.out_mag = dst.mag;
    a.scale = dst.scale;
    if (dst.mag && !inv) return hipErrorInvalidValue;
    return launch_az_tile(S, 32, inv, dst.mag ? AZ_EPI_CROPOUT_MAG : AZ_EPI_CROPOUT, a, RA, st);
}

// line transform in place on w [n_az x m_rg] (chirped and padded when the axis is not direct)
static hipError_t rows_core(GeneralCsa* g, cf* w, bool inv, hipStream_t st) {
    const Axis& ax = g->rg;
    if (ax.direct) return rows_pow2(g, w, g->n_az, g->n_rg, inv, st);
    GCK(rows_pow2(g, w, g->n_az, ax.m, false, st, inv ? ax.bhat_i : ax.bhat_f));         // * filter spectrum in the epilogue
    return rows_pow2(g, w, g->n_az, ax.m, true, st);
}

// ---- range FFT . Phi_2 . IFFT as ONE convolution (range extent not a power of two) ---------------------------------
// FFT_N, * Phi_2, IFFT_N along a line is the circular convolution of the line with phi_i = IFFT_N(Phi_2[i, :]).  Two chirp-z
// transforms would cost four M-point FFTs per line; the convolution itself needs two: pad the line to M >= 2N - 1, FFT_M,
// multiply by K_i = FFT_M(g_i) with g_i[lag mod M] = phi_i[lag mod N] for lags -(N-1)..N-1, IFFT_M, keep N samples.  K is a
// [n_az x M] table built once per plan with the chirp-z machinery itself (1.9 GB for 7199 x 13200; read in the forward
// transform's epilogue).
__global__ __launch_bounds__(256) void kext_kernel(const cf* phi, size_t phi_ld, const cf* chirp, float scale, int n, int m, int rows, cf* out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    int idx = -1;
    if (j < n) idx = j;                              // lags 0 .. N-1
    else if (j >= m - n + 1) idx = j - m + n;        // lags -(N-1) .. -1  ->  phi[lag + N]
    cf w = make_float2(0.f, 0.f);
    if (idx >= 0) { const cf c = chirp[idx]; w = make_float2(c.x * scale, c.y * scale); }
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        out[(size_t)r * m + j] = idx >= 0 ? cmul(phi[(size_t)r * phi_ld + idx], w) : make_float2(0.f, 0.f);
}

static hipError_t build_range_kernel_table(GeneralCsa* g, hipStream_t st) {
    const Axis& rg = g->rg;
    const int n_az = g->n_az, n_rg = g->n_rg, m = rg.m;
    cf* lines = g->work_a;
    // Phi_2 * pre-chirp of the inverse chirp-z, padded; inverse core; post-chirp / N and the lag layout; FFT_M
    GCK(bridge<2>(g, nullptr, 0, lines, m, n_az, m, nullptr, rg.chirp_i, 1.0f, st));
    GCK(rows_pow2(g, lines, n_az, m, false, st, rg.bhat_i));
    GCK(rows_pow2(g, lines, n_az, m, true, st));
    dim3 grid((m + 255) / 256, n_az < 16384 ? n_az : 16384);
    hipLaunchKernelGGL(kext_kernel, grid, dim3(256), 0, st, lines, (size_t)m, rg.chirp_i, 1.0f / (float)n_rg, n_rg, m, n_az, g->ktab);
    GCK(hipGetLastError());
    GCK(rows_pow2(g, g->ktab, n_az, m, false, st));
    return hipStreamSynchronize(st);
}

// one range pass of the plan's geometry on a dense [n_az x n_rg] image (direct mixed-radix line lengths only)
hipError_t general_csa_range_pass(GeneralCsa* g, int mode, const float2* in, float2* out, hipStream_t st) {
    if (!g->rg_mixed) return hipErrorNotSupported;
    RangeArgs a{};
    a.in = in; a.out = out; a.c2 = g->c2; a.c3 = g->c3;
    a.dt = 1.0 / g->p.sample_rate_hz; a.df = 1.0 / ((double)g->n_rg * a.dt);
    a.t_start = g->p.t_start_fast_s; a.t0 = 2.0 * g->p.range_ref_m / 299792458.0;
    a.inv_n = 1.0f / (float)g->n_rg; a.n_az = g->n_az;
    return launch_range_mixed(g->n_rg, mode, a, g->cus, st);
}

// one azimuth pass (forward + Phi_1, or inverse with 1/n_az) of the plan's geometry on dense [n_az x n_rg] images
// (prime-factor pulse counts only; `in` is not modified, in != out)
hipError_t general_csa_az_pass(GeneralCsa* g, bool inv, const float2* in, float2* out, hipStream_t st) {
    if (!g->pfa) return hipErrorNotSupported;
    const int n_rg = g->n_rg;
    if (!inv)
        return az_pfa_run(g->pfa, false, in, n_rg, n_rg, g->work_a, g->ldc, out, n_rg, n_rg, 1, g->c1, 1.0 / g->p.sample_rate_hz,
                          g->p.t_start_fast_s, 1.0f, st);
    return az_pfa_run(g->pfa, true, in, n_rg, n_rg, g->work_a, g->ldc, out, n_rg, n_rg, 2, nullptr, 0.0, 0.0, 1.0f / (float)g->n_az, st);
}

// sar_focus_csa (:233-385) at any size.  Buffers: wa/wb [m_az x ldc] for the column transforms, work_a doubles as
// the [n_az x m_rg] line-transform array when the range axis is not a power of two (else the dense image `data`).
hipError_t general_csa_focus(GeneralCsa* g, const float2* d_in, float2* d_out, hipStream_t st) {
    const Axis &az = g->az, &rg = g->rg;
    const int n_az = g->n_az, n_rg = g->n_rg, ld = g->ldc;
    cf *wa = g->work_a, *wb = g->work_b;
    if (g->rg_mixed) {
        // the range extent has a direct line kernel (13200): no chirp-z along range, one fused launch for :278-382
        const bool z3 = !az.direct && cols_two_step(az.m);
        if (g->pfa) {                 // 7199 = 23 x 313 pulses: prime-factor transforms, two launches each, no padding at all
            const double dt = 1.0 / g->p.sample_rate_hz;
            GCK(az_pfa_run(g->pfa, false, d_in, n_rg, n_rg, wa, ld, g->data, n_rg, n_rg, 1, g->c1, dt, g->p.t_start_fast_s, 1.0f, st));
            GCK(general_csa_range_pass(g, RG_FUSED, g->data, g->data, st));
            return az_pfa_run(g->pfa, true, g->data, n_rg, n_rg, wa, ld, d_out, n_rg, n_rg, 2, nullptr, 0.0, 0.0, 1.0f / (float)n_az, st, g->ati.s1 ? nullptr : g->max_slot, &g->ati);
        }
        if (z3) {                     // azimuth FFT (:233) as a three-launch chirp-z with Phi_1 (:272-274) in its last epilogue
            const ColsSrc src{d_in, (size_t)n_rg, n_az, n_rg, az.chirp_f};
            const ColsDst dst{g->data, (size_t)n_rg, n_az, n_rg, az.chirp_f, 1.0f};
            GCK(cols_bluestein3(g, wa, false, st, src, dst, AZ_EPI_CROPOUT_PHI1));
        } else {
            cf *first = az.direct ? wa : wb, *other = az.direct ? wb : wa, *res = nullptr;
            if (cols_two_step(az.m)) {
                const ColsSrc src{d_in, (size_t)n_rg, n_az, n_rg, az.direct ? nullptr : az.chirp_f};
                GCK(cols_core(g, first, other, false, st, &res, &src));
            } else {
                GCK(scale_copy(d_in, n_az, n_rg, n_rg, first, az.m, ld, ld, az.direct ? nullptr : az.chirp_f, nullptr, 1.0f, st));
                GCK(cols_core(g, first, other, false, st, &res));
            }
            GCK(bridge<1>(g, res, ld, g->data, n_rg, n_az, n_rg, az.direct ? nullptr : az.chirp_f, nullptr, 1.0f, st));
        }
        GCK(general_csa_range_pass(g, RG_FUSED, g->data, g->data, st));   // range FFT . Phi_2 . IFFT . Phi_3 in place on the dense image
        if (z3) {                     // azimuth IFFT (:385)
            const ColsSrc src{g->data, (size_t)n_rg, n_az, n_rg, az.chirp_i};
            const ColsDst dst{d_out, (size_t)n_rg, n_az, n_rg, az.chirp_i, 1.0f / (float)n_az};
            return cols_bluestein3(g, wa, true, st, src, dst, AZ_EPI_CROPOUT);
        }
        cf* res = nullptr;
        GCK(scale_copy(g->data, n_az, n_rg, n_rg, wb, cols_two_step(az.m) ? n_az : az.m, ld, ld, az.direct ? nullptr : az.chirp_i,
                       nullptr, 1.0f, st));
        if (cols_two_step(az.m)) {
            const ColsDst dst{d_out, (size_t)n_rg, n_az, n_rg, az.direct ? nullptr : az.chirp_i, az.direct ? 1.0f : 1.0f / (float)n_az};
            return cols_core(g, wb, wa, true, st, &res, nullptr, &dst);
        }
        GCK(cols_core(g, wb, wa, true, st, &res));
        return scale_copy(res, n_az, n_rg, ld, d_out, n_az, n_rg, n_rg, nullptr, nullptr, az.direct ? 1.0f : 1.0f / (float)n_az, st, 0,
                          az.direct ? nullptr : az.chirp_i);
    }
    cf* lines = rg.direct ? g->data : g->work_a;            // where the range transforms run
    const size_t lines_ld = rg.direct ? (size_t)n_rg : (size_t)rg.m;
    const int lines_cols = rg.direct ? n_rg : rg.m;
    // azimuth FFT (:233): d_in -> (chirp, pad) -> wb
    // (the result must land in wb: work_a may become the line array next)
    cf *first = az.direct ? wa : wb, *other = az.direct ? wb : wa, *res = nullptr;
    const bool fused_ends = cols_two_step(az.m);             // copy-in / copy-out folded into the first / last tile launch
    if (fused_ends) {
        const ColsSrc src{d_in, (size_t)n_rg, n_az, n_rg, az.direct ? nullptr : az.chirp_f};
        GCK(cols_core(g, first, other, false, st, &res, &src));
    } else {
        GCK(scale_copy(d_in, n_az, n_rg, n_rg, first, az.m, ld, ld, az.direct ? nullptr : az.chirp_f, nullptr, 1.0f, st));
        GCK(cols_core(g, first, other, false, st, &res));
    }
    if (g->ktab) {
        // * azimuth post-chirp * Phi_1 (:272-274), zero-padded lines; range FFT . Phi_2 . IFFT (:278-331) as one convolution
        const int vlen = rg.m > 16384 ? n_rg : 0;       // split lines: the zero padding is neither written nor read
        GCK(bridge<1>(g, res, ld, lines, lines_ld, n_az, vlen ? n_rg : lines_cols, az.direct ? nullptr : az.chirp_f, nullptr, 1.0f, st));
        GCK(rows_pow2(g, lines, n_az, rg.m, false, st, g->ktab, n_az, vlen));
        GCK(rows_pow2(g, lines, n_az, rg.m, true, st, nullptr, 0, vlen));
        // * Phi_3 (:359-382) * azimuth pre-chirp, into wb (work_a holds the lines)
        GCK(bridge<3>(g, lines, lines_ld, wb, ld, fused_ends ? n_az : az.m, ld, az.direct ? nullptr : az.chirp_i, nullptr, 1.0f, st));
    } else {
        // * azimuth post-chirp * Phi_1 (:272-274) * range pre-chirp, into the line array
        GCK(bridge<1>(g, res, ld, lines, lines_ld, n_az, lines_cols, az.direct ? nullptr : az.chirp_f, rg.direct ? nullptr : rg.chirp_f,
                      1.0f, st));
        GCK(rows_core(g, lines, false, st));                                              // :278
        // post-chirp of the forward and pre-chirp of the inverse are conjugates: only Phi_2 (:318-326) and the zero padding remain
        GCK(bridge<2>(g, lines, lines_ld, lines, lines_ld, n_az, lines_cols, nullptr, nullptr, 1.0f, st));
        GCK(rows_core(g, lines, true, st));                                               // :331
        // * range post-chirp / n_rg * Phi_3 (:359-382) * azimuth pre-chirp, into wb (work_a may hold the lines)
        GCK(bridge<3>(g, lines, lines_ld, wb, ld, fused_ends ? n_az : az.m, ld, az.direct ? nullptr : az.chirp_i,
                      rg.direct ? nullptr : rg.chirp_i, rg.direct ? 1.0f : 1.0f / (float)n_rg, st));
    }
    if (fused_ends) {                                                                 // :385, post-chirp / n_az and crop in the last launch
        const ColsDst dst{d_out, (size_t)n_rg, n_az, n_rg, az.direct ? nullptr : az.chirp_i, az.direct ? 1.0f : 1.0f / (float)n_az};
        return cols_core(g, wb, wa, true, st, &res, nullptr, &dst);
    }
    GCK(cols_core(g, wb, wa, true, st, &res));                                        // :385
    return scale_copy(res, n_az, n_rg, ld, d_out, n_az, n_rg, n_rg, nullptr, nullptr, az.direct ? 1.0f : 1.0f / (float)n_az, st, 0,
                      az.direct ? nullptr : az.chirp_i);
}

// =====================================================================================================
// Range-Doppler focuser: sar_focus_rda, sar_satellite_sim.py:356-448 (pasted again at
// sar_satellite_moving_sim.py:208 and sar_vehicle_sim.py:182).  Internal layout [pulse][range] (range
// contiguous): the transpose of the reference's [range][pulse] argument, which the scripts themselves
// obtain as raw.T, so the host passes the underlying memory through unchanged.
//   1 range compression: linear convolution with the Hamming-weighted unit-norm chirp, 'same' window
//     (:375-392) as FFT_M . H . IFFT_M along the lines, M >= n_r + L - 1
//   2 Hamming over pulses, fftshift . FFT . fftshift along azimuth (:396-399)
//   3 RCMC: the profile sampled at r(1 - a_k) is read back at r by linear interpolation, 0 outside (:411-427)
//   4 azimuth compression exp(-i pi fd^2 / Ka(r)) (:431-435)
//   5 ifftshift . IFFT . ifftshift, magnitude (:438-440)
// =====================================================================================================
struct RdaArgsDev {
    const cf* in;
    cf* out;
    float* mag;
    const double* fd;      // [n_p] Doppler axis (fftshift order, :402-405)
    const double* r_axis;  // [n_r] range axis in metres (:407)
    int n_p, n_r;
    double k_rcmc;         // lambda^2 / (8 Vr^2)
    double k_ac;           // lambda / (2 Vr^2):  1/Ka = k_ac * r
    const double2* rowc;   // [n_p] {s_k = 1 - alpha_k, 1 / s_k} (rda_rcmc_azcomp_rows_kernel: no fp64 division per pixel)
    double inv_dr;         // (n_r - 1) / (r_axis[n_r - 1] - r_axis[0])
    // rda_rcmc_azcomp*_kernel: Doppler row k goes to row (k + out_shift) mod n_p of `out` (leading dimension out_ld): the
    // ifftshift in front of the inverse transform folded into this store (power-of-two direct route); 0 / n_r otherwise
    int out_shift; size_t out_ld;
    cf* ac_out;            // optional: the azimuth-compressed map in Doppler order, dense (sar_vehicle_sim.py:268 range_doppler_filtered)
};
// out[k][j] = lerp of in[k][.] at u = (r_j/(1-a_k) - r_0)/dr, zero outside the sampled span
__global__ __launch_bounds__(256) void rda_rcmc_kernel(RdaArgsDev a) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.n_r) return;
    const double r0 = a.r_axis[0], rj = a.r_axis[j];
    const double dr = (a.n_r > 1) ? (a.r_axis[a.n_r - 1] - r0) / (double)(a.n_r - 1) : 1.0;
    for (int k = blockIdx.y; k < a.n_p; k += gridDim.y) {
        const double alpha = a.fd[k] * a.fd[k] * a.k_rcmc;            // delta_R = r * alpha  (:414)
        const cf* row = a.in + (size_t)k * a.n_r;
        cf y = make_float2(0.f, 0.f);
        if (a.n_r == 1) {
            y = row[0];
        } else {
            const double s = 1.0 - alpha;                            // sample positions x_j' = r_j' * s
            const double x_first = r0 * s, x_last = a.r_axis[a.n_r - 1] * s;
            if (rj >= x_first && rj <= x_last) {
                double u = (rj / s - r0) / dr;
                int j0 = (int)floor(u);
                if (j0 < 0) j0 = 0;
                if (j0 > a.n_r - 2) j0 = a.n_r - 2;
                // guard the fp rounding of u against the exact sample positions
                while (j0 > 0 && a.r_axis[j0] * s > rj) --j0;
                while (j0 < a.n_r - 2 && a.r_axis[j0 + 1] * s <= rj) ++j0;
                const double xa = a.r_axis[j0] * s, xb = a.r_axis[j0 + 1] * s;
                const float f = (float)((rj - xa) / (xb - xa));
                const cf p = row[j0], q = row[j0 + 1];
                y = make_float2(fmaf(f, q.x - p.x, p.x), fmaf(f, q.y - p.y, p.y));
            }
        }
        a.out[(size_t)k * a.n_r + j] = y;
    }
}
// out = in * exp(-i pi fd_k^2 * k_ac * r_j)
__global__ __launch_bounds__(256) void rda_azcomp_kernel(RdaArgsDev a) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.n_r) return;
    const double g = -0.5 * a.k_ac * a.r_axis[j];                     // revolutions per Hz^2
    for (int k = blockIdx.y; k < a.n_p; k += gridDim.y) {
        const size_t i = (size_t)k * a.n_r + j;
        a.out[i] = cmul(a.in[i], cis_rev(g * a.fd[k] * a.fd[k]));
    }
}
// RCMC (:411-427) and azimuth compression (:431-435) in one pass over the range-Doppler map: the migrated sample is
// interpolated (rda_rcmc_kernel's arithmetic), stored to the RCMC map only when the caller asked for that
// intermediate, multiplied by the compression phase and stored once.  One HBM round trip instead of two.
__global__ __launch_bounds__(256) void rda_rcmc_azcomp_kernel(RdaArgsDev a, cf* rc_out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.n_r) return;
    const double r0 = a.r_axis[0], rj = a.r_axis[j];
    const double dr = (a.n_r > 1) ? (a.r_axis[a.n_r - 1] - r0) / (double)(a.n_r - 1) : 1.0;
    const double g = -0.5 * a.k_ac * rj;
    for (int k = blockIdx.y; k < a.n_p; k += gridDim.y) {
        const double fd2 = a.fd[k] * a.fd[k];
        const double alpha = fd2 * a.k_rcmc;
        const cf* row = a.in + (size_t)k * a.n_r;
        cf y = make_float2(0.f, 0.f);
        if (a.n_r == 1) {
            y = row[0];
        } else {
            const double s = 1.0 - alpha;
            const double x_first = r0 * s, x_last = a.r_axis[a.n_r - 1] * s;
            if (rj >= x_first && rj <= x_last) {
                double u = (rj / s - r0) / dr;
                int j0 = (int)floor(u);
                if (j0 < 0) j0 = 0;
                if (j0 > a.n_r - 2) j0 = a.n_r - 2;
                while (j0 > 0 && a.r_axis[j0] * s > rj) --j0;
                while (j0 < a.n_r - 2 && a.r_axis[j0 + 1] * s <= rj) ++j0;
                const double xa = a.r_axis[j0] * s, xb = a.r_axis[j0 + 1] * s;
                const float f = (float)((rj - xa) / (xb - xa));
                const cf p = row[j0], q = row[j0 + 1];
                y = make_float2(fmaf(f, q.x - p.x, p.x), fmaf(f, q.y - p.y, p.y));
            }
        }
        const size_t i = (size_t)k * a.n_r + j;
        if (rc_out) rc_out[i] = y;
        const cf z = cmul(y, cis_rev(g * fd2));
        if (a.ac_out) a.ac_out[i] = z;
        int ko = k + a.out_shift;
        if (ko >= a.n_p) ko -= a.n_p;
        a.out[(size_t)ko * a.out_ld + j] = z;
    }
}
// The same pass with one Doppler row per blockIdx.y and four range samples per thread: everything that depends on the row
// only (s, 1/s, the span of the migrated axis) comes from a table built at plan creation, so a pixel costs one fused
// multiply-add for its fractional position instead of three fp64 divisions (0.86 -> 0.37 ms at 13200 x 7200: profiles/r03_f_*).
struct __attribute__((aligned(8))) RcmcF4 { float x, y, z, w; };  // two neighbouring samples as ONE 16-byte access (8-byte aligned: the hardware takes an unaligned dwordx4)
__global__ __launch_bounds__(256) void rda_rcmc_azcomp_rows_kernel(RdaArgsDev a, cf* rc_out) {
    const int k = blockIdx.y;
    const double2 sc = a.rowc[k];
    const double s = sc.x, r0 = a.r_axis[0];
    const double x_first = r0 * s, x_last = a.r_axis[a.n_r - 1] * s;
    const double c_a = sc.y * a.inv_dr, c_b = r0 * a.inv_dr;      // u = r_j / (s dr) - r_0 / dr;  1 / (x_b - x_a) = 1 / (s dr)
    const double fd2 = a.fd[k] * a.fd[k];
    const cf* row = a.in + (size_t)k * a.n_r;
    const int last = a.n_r - 1;
    int ko = k + a.out_shift;
    if (ko >= a.n_p) ko -= a.n_p;
    // The range axis is affine in the sample index to an ulp (:370-373,407), so the positions r_j and the bracket's two positions
    // come from r_0 + j dr instead of the table: per pixel the table cost 24 bytes from L2 beside the 16 bytes of the bracket's
    // samples, in a launch that moves 16 bytes per pixel through HBM.  Linear interpolation is continuous in the position, so a
    // bracket chosen one ulp differently at a tie gives the same value; only the two ENDS of the span decide between a value
    // and zero (np.interp left = right = 0), and they are taken from the table itself (x_first, x_last above; the last
    // sample's own position below).  A thread's four gathers are issued before anything waits for them.
    const double dr = 1.0 / a.inv_dr, r_last = a.r_axis[last];
    double rj[4], x0[4];
    int jc[4];
    RcmcF4 pq[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int j = blockIdx.x * 1024 + m * 256 + threadIdx.x;
        const int jj = j < a.n_r ? j : last;
        rj[m] = jj == last ? r_last : fma((double)jj, dr, r0);
        int c = (int)floor(fma(rj[m], c_a, -c_b));
        c = c < 0 ? 0 : (c > last - 1 ? last - 1 : c);
        double xa = fma((double)c, dr, r0) * s;
        const double xb = (c + 1 == last ? r_last : fma((double)(c + 1), dr, r0)) * s;
        if (c > 0 && xa > rj[m]) { --c; xa = fma((double)c, dr, r0) * s; }            // u rounded across an integer
        else if (c < last - 1 && xb <= rj[m]) { ++c; xa = xb; }
        jc[m] = c; x0[m] = xa;
        pq[m] = *reinterpret_cast<const RcmcF4*>(row + c);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int j = blockIdx.x * 1024 + m * 256 + threadIdx.x;
        if (j >= a.n_r) continue;
        cf y = make_float2(0.f, 0.f);
        if (rj[m] >= x_first && rj[m] <= x_last) {
            const float f = (float)((rj[m] - x0[m]) * c_a);
            y = make_float2(fmaf(f, pq[m].z - pq[m].x, pq[m].x), fmaf(f, pq[m].w - pq[m].y, pq[m].y));
        }
        const size_t i = (size_t)k * a.n_r + j;
        if (rc_out) rc_out[i] = y;
        const cf z = cmul(y, cis_rev(-0.5 * a.k_ac * rj[m] * fd2));
        if (a.ac_out) a.ac_out[i] = z;
        a.out[(size_t)ko * a.out_ld + j] = z;
    }
}
__global__ __launch_bounds__(256) void rda_mag_kernel(const cf* in, float* mag, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) mag[i] = hypotf(in[i].x, in[i].y);
}

struct Rda {
    GeneralCsa* g = nullptr;      // buffers, azimuth axis (n_az = pulses, n_rg = ranges)
    int n_p = 0, n_r = 0, m_c = 0, l_mf = 0;
    cf *hhat = nullptr, *win = nullptr;        // filter spectrum [m_c] (split order at 32768), azimuth window [n_p]
    cf* pre_f = nullptr;                       // chirp-z azimuth: pre-chirp[r] * window[(r + shift) mod n_p], per sequence index r
    // direct route (13200 ranges x 7200 pulses, the satellite scripts' size): range compression as ONE circular convolution of
    // length m_conv (range_mixed.hip, RG_CONV), pulse-axis transforms by the 32 x 225 prime-factor kernels (az_pfa7200.hip)
    int m_conv = 0; cf* hhat_conv = nullptr;   // filter spectrum at m_conv points, natural order
    bool pfa72 = false; float* winf = nullptr; // azimuth window as floats [n_p]
    // power-of-two direct route (the airborne script's 2048 ranges x 32768 pulses, sar_vehicle_sim.py:41,86): range compression as
    // ONE circular convolution of power-of-two length m_conv <= 16384 in registers / LDS (range_pass_kernel<m_conv, RG_CONV>),
    // pulse-axis transforms as the two four-step launches with window, both fftshifts and the magnitude in their first / last step
    bool az2 = false;
    cf* ac = nullptr;                          // azimuth-compressed map (the airborne script's eighth output), allocated on first request
    double2* rowc = nullptr; double inv_dr = 0;  // per Doppler row {1 - alpha, 1 / (1 - alpha)} for the row-wise RCMC kernel
    int cus = 256;
    double *fd = nullptr, *r_axis = nullptr;
    cf *pc = nullptr, *rd = nullptr, *rc = nullptr;   // the three intermediates the reference returns
    float* mag = nullptr;
    std::vector<double> h_fd, h_r;
    double lam = 0, vr = 0, prf = 0;
    uint64_t bytes = 0;
};

void rda_destroy(Rda* r) {
    if (!r) return;
    general_csa_destroy(r->g);
    hipFree(r->hhat); hipFree(r->win); hipFree(r->pre_f); hipFree(r->fd); hipFree(r->r_axis); hipFree(r->hhat_conv); hipFree(r->winf); hipFree(r->rowc);
    hipFree(r->pc); hipFree(r->rd); hipFree(r->rc); hipFree(r->mag); hipFree(r->ac);
    delete r;
}

Rda* rda_create(int n_r, int n_p, const sarx_radar_params* prm, const float2* tw_all, std::string& err, int cus) {
    // prm: wavelength, pulse width, chirp rate, sample rate, prf, platform speed, range_ref = range_grp_m
    const double fs = prm->sample_rate_hz, tp = prm->pulse_width_s, kr = prm->chirp_rate_hz_s;
    const int l_mf = (int)floor(tp / (1.0 / fs)) + 1;                 // :377-378
    if (l_mf < 1 || n_r < 2 || n_p < 2) { err = "RDA needs n_ranges, n_pulses >= 2 and a positive pulse width"; return nullptr; }
    int m_c = 16;
    while (m_c < n_r + l_mf - 1) m_c <<= 1;
    if (m_c > 65536) { err = "range samples + matched-filter taps - 1 must be <= 65536"; return nullptr; }
    Rda* r = new Rda();
    r->n_p = n_p; r->n_r = n_r; r->m_c = m_c; r->l_mf = l_mf;
    r->lam = prm->wavelength_m; r->vr = prm->platform_speed_mps; r->prf = prm->prf_hz;
    if (cus > 0) r->cus = cus;
    {   // direct route switches (SARX_RDA_DIRECT=0 keeps the chirp-z / padded power-of-two route for A/B)
        const char* de = getenv("SARX_RDA_DIRECT");
        const int direct = de ? atoi(de) : 1;
        // a circular convolution of length M holds the 'same' window [c0, c0 + n_r) of the n_r + l_mf - 1 sample full
        // convolution when the wrapped ends miss it: M >= full - c0 and M >= c0 + n_r
        const int c0 = (l_mf - 1) / 2, full = n_r + l_mf - 1, m_need = std::max(full - c0, c0 + n_r);
        if (direct && m_c > 16384 && range_conv_supported(19683) && m_need <= 19683 && n_r <= 19683) r->m_conv = 19683;
        if (direct && !r->m_conv) {      // shortest power of two that holds the 'same' window; the line kernels go up to 16384
            int mp = 16;
            while (mp < m_need) mp <<= 1;
            if (mp <= 16384) r->m_conv = mp;
        }
        r->pfa72 = direct && az_pfa7200_supported(n_p);
        r->az2 = direct && !r->pfa72 && is_pow2(n_p) && cols_two_step(n_p) && n_p <= 32768;
    }
    r->g = general_csa_create(n_p, n_r, prm, tw_all, err, false, r->cus);      // buffers and the azimuth axis only
    if (!r->g) { delete r; return nullptr; }
    auto bail = [&](const char* what, hipError_t e) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        rda_destroy(r);
        return (Rda*)nullptr;
    };
    hipError_t e;
    // the convolution needs [n_p x m_c] work arrays
    const size_t need = r->m_conv ? 0 : (size_t)n_p * m_c;
    if (need > r->g->work_elems) {
        hipFree(r->g->work_a); hipFree(r->g->work_b);
        r->g->work_a = r->g->work_b = nullptr;
        if ((e = hipMalloc(&r->g->work_a, need * sizeof(cf))) != hipSuccess) return bail("hipMalloc work", e);
        if ((e = hipMalloc(&r->g->work_b, need * sizeof(cf))) != hipSuccess) return bail("hipMalloc work", e);
        r->g->work_elems = need;
    }
    // matched filter: conj chirp * Hamming, unit norm (:379-385); its spectrum at length m_c
    std::vector<zd> h(m_c, zd(0, 0));
    double nrm = 0;
    for (int k = 0; k < l_mf; ++k) {
        const double t = (l_mf > 1) ? -tp / 2 + (double)k * (tp / (double)(l_mf - 1)) : -tp / 2;
        const double w = (l_mf > 1) ? 0.54 - 0.46 * cos(2.0 * M_PI * (double)k / (double)(l_mf - 1)) : 1.0;
        h[k] = std::polar(w, -M_PI * kr * t * t);
        nrm += w * w;
    }
    nrm = sqrt(nrm);
    for (int k = 0; k < l_mf; ++k) h[k] /= nrm;
    if (r->m_conv) {                 // the same taps at the circular length of the direct kernel
        std::vector<zd> hc(r->m_conv, zd(0, 0));
        for (int k = 0; k < l_mf; ++k) hc[k] = h[k];
        if (is_pow2(r->m_conv)) host_fft(hc); else host_dft_any(hc);
        if ((e = upload(hc, &r->hhat_conv)) != hipSuccess) return bail("upload filter", e);
    }
    host_fft(h);
    if (m_c > 16384) to_split_order(h);
    if ((e = upload(h, &r->hhat)) != hipSuccess) return bail("upload filter", e);
    std::vector<zd> win(n_p);
    for (int i = 0; i < n_p; ++i) win[i] = zd(n_p > 1 ? 0.54 - 0.46 * cos(2.0 * M_PI * (double)i / (double)(n_p - 1)) : 1.0, 0.0);
    if ((e = upload(win, &r->win)) != hipSuccess) return bail("upload window", e);
    if (r->pfa72) {
        std::vector<float> wf(n_p);
        for (int i = 0; i < n_p; ++i) wf[i] = (float)win[i].real();
        if ((e = hipMalloc(&r->winf, n_p * sizeof(float))) != hipSuccess) return bail("hipMalloc window", e);
        if ((e = hipMemcpy(r->winf, wf.data(), n_p * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return bail("upload window", e);
    }
    if (r->az2) {
        // (the Hamming window over pulses, :396, is evaluated by the first launch itself: AzArgs::hamming_inv)
        // pad columns of the work arrays (ldc > n_r) are transformed along with the rest and never copied out: keep them finite
        if ((e = hipMemset(r->g->work_a, 0, r->g->work_elems * sizeof(cf))) != hipSuccess) return bail("hipMemset", e);
        if ((e = hipMemset(r->g->work_b, 0, r->g->work_elems * sizeof(cf))) != hipSuccess) return bail("hipMemset", e);
    } else if (!r->pfa72 && !r->g->az.direct && cols_two_step(r->g->az.m)) {
        // three-launch chirp-z along pulses: window and fftshift are folded into the copy-in; sequence element e is
        // source row (e + sh) mod n_p (the roll by n_p/2 of :396-399), so it carries chirp[e] * window[(e + sh) mod n_p]
        const int sh = (n_p - n_p / 2) % n_p;
        std::vector<zd> pre(n_p);
        for (int k = 0; k < n_p; ++k) {
            const long long k2 = ((long long)k * k) % (2LL * n_p);
            pre[k] = std::polar(1.0, -M_PI * (double)k2 / (double)n_p) * win[(k + sh) % n_p];
        }
        if ((e = upload(pre, &r->pre_f)) != hipSuccess) return bail("upload window", e);
    }
    // axes (:363-373, :402-407)
    r->h_fd.resize(n_p); r->h_r.resize(n_r);
    const double t_grp = 2.0 * prm->range_ref_m / 299792458.0;
    for (int i = 0; i < n_p; ++i) {
        const double c0 = (n_p % 2 == 0) ? -(double)n_p / 2 : -((double)n_p - 1) / 2;
        r->h_fd[i] = (c0 + (double)i) * (prm->prf_hz / (double)n_p);
    }
    for (int j = 0; j < n_r; ++j) {
        const double c0 = (n_r % 2 == 0) ? (double)n_r / 2 : ((double)n_r - 1) / 2;
        r->h_r[j] = (((double)j - c0) / fs + t_grp) * 299792458.0 / 2;
    }
    if ((e = hipMalloc(&r->fd, n_p * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&r->r_axis, n_r * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemcpy(r->fd, r->h_fd.data(), n_p * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return bail("upload", e);
    if ((e = hipMemcpy(r->r_axis, r->h_r.data(), n_r * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return bail("upload", e);
    {   // row constants of the RCMC pass (:414-418): alpha_k = fd_k^2 lambda^2 / (8 Vr^2)
        const double k_rcmc = r->lam * r->lam / (8.0 * r->vr * r->vr);
        std::vector<double2> rc(n_p);
        for (int i = 0; i < n_p; ++i) { const double sk = 1.0 - r->h_fd[i] * r->h_fd[i] * k_rcmc; rc[i] = make_double2(sk, 1.0 / sk); }
        if ((e = hipMalloc(&r->rowc, n_p * sizeof(double2))) != hipSuccess) return bail("hipMalloc", e);
        if ((e = hipMemcpy(r->rowc, rc.data(), n_p * sizeof(double2), hipMemcpyHostToDevice)) != hipSuccess) return bail("upload", e);
        r->inv_dr = (n_r > 1) ? (double)(n_r - 1) / (r->h_r[n_r - 1] - r->h_r[0]) : 1.0;
    }
    const size_t img = (size_t)n_p * n_r;
    if ((e = hipMalloc(&r->pc, img * sizeof(cf))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&r->rd, img * sizeof(cf))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&r->rc, img * sizeof(cf))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&r->mag, img * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
    r->bytes = general_csa_bytes(r->g) + img * (3 * sizeof(cf) + sizeof(float));
    return r;
}

// d_in: [n_p x n_r] complex64 (the reference's phist transposed).  Results stay in the object's buffers.
hipError_t rda_focus(Rda* r, const float2* d_in, hipStream_t st, float* mag_out, bool want_rc, bool want_ac) {
    GeneralCsa* g = r->g;
    if (!mag_out) mag_out = r->mag;
    if (want_ac && !r->ac) GCK(hipMalloc(&r->ac, (size_t)r->n_p * r->n_r * sizeof(cf)));
    const int n_p = r->n_p, n_r = r->n_r, m = r->m_c;
    cf* w = g->work_a;
    // 1 range compression
    if (r->m_conv) {
        // one launch: zero-padded FFT . filter spectrum . IFFT of every pulse at the circular length, 'same' window out (:388-392)
        RangeArgs ca{};
        ca.in = d_in; ca.out = r->pc; ca.n_az = n_p; ca.inv_n = 1.0f / (float)r->m_conv; ca.mulvec = r->hhat_conv; ca.mul_period = 1;
        ca.conv_valid = n_r; ca.conv_crop0 = (r->l_mf - 1) / 2; ca.conv_out = n_r; ca.conv_in_ld = (size_t)n_r; ca.conv_out_ld = (size_t)n_r;
        if (is_pow2(r->m_conv)) { ca.tw = g->tw_all + r->m_conv; GCK(launch_range_pass(r->m_conv, RG_CONV, ca, st)); }
        else GCK(launch_range_conv(r->m_conv, ca, r->cus, st));
    } else {
    // split lines (m > 16384): the zero padding is neither written nor read, and only the 'same' window is written back
    const int vin = m > 16384 ? n_r : 0, vout = m > 16384 ? (r->l_mf - 1) / 2 + n_r : 0;
    GCK(scale_copy(d_in, n_p, n_r, n_r, w, n_p, vin ? n_r : m, m, nullptr, nullptr, 1.0f, st));
    GCK(rows_pow2(g, w, n_p, m, false, st, r->hhat, 0, vin));    // * filter spectrum in the epilogue
    GCK(rows_pow2(g, w, n_p, m, true, st, nullptr, 0, vout));
    GCK(scale_copy(w + (r->l_mf - 1) / 2, n_p, n_r, m, r->pc, n_p, n_r, n_r, nullptr, nullptr, 1.0f, st));   // mode='same'
    }
    // 2 window, fftshift . FFT . fftshift over pulses: roll by h = n_p/2 is source row (r - h) mod n
    const int h = n_p / 2, sh = (n_p - h) % n_p;
    const bool z3 = r->pre_f != nullptr;           // chirp-z along pulses in three launches, shifts / window / magnitude in its ends
    if (r->az2) {                                  // two four-step launches: window and fftshift while copying in, fftshift while copying out
        const ColsSrc src{r->pc, (size_t)n_r, n_p, n_r, nullptr, sh};
        const ColsDst dst{r->rd, (size_t)n_r, n_p, n_r, nullptr, 1.0f, (n_p - sh) % n_p, nullptr};
        GCK(cols_pow2_ends(g, g->work_a, n_p, false, st, src, dst, 1.0f / (float)(n_p - 1)));
    } else if (r->pfa72) {                         // two prime-factor launches, window and both fftshifts in their row addresses
        GCK(az_pfa7200_run(false, r->pc, (size_t)n_r, n_r, g->work_a, (size_t)g->ldc, r->rd, nullptr, (size_t)n_r, sh, sh, r->winf, 1.0f, st));
    } else if (z3) {
        const ColsSrc src{r->pc, (size_t)n_r, n_p, n_r, r->pre_f, sh};
        const ColsDst dst{r->rd, (size_t)n_r, n_p, n_r, g->az.chirp_f, 1.0f, (n_p - sh) % n_p, nullptr};
        GCK(cols_bluestein3(g, g->work_a, false, st, src, dst, AZ_EPI_CROPOUT));
    } else {
        GCK(fft_cols(g, r->pc, r->rd, false, st, sh, r->win, sh));
    }
    // 3, 4: RCMC and azimuth compression in one pass (the RCMC map is stored only on request)
    RdaArgsDev a{};
    a.fd = r->fd; a.r_axis = r->r_axis; a.n_p = n_p; a.n_r = n_r;
    a.k_rcmc = r->lam * r->lam / (8.0 * r->vr * r->vr);
    a.k_ac = r->lam / (2.0 * r->vr * r->vr);
    a.in = r->rd; a.out = g->data; a.rowc = r->rowc; a.inv_dr = r->inv_dr;
    a.out_shift = 0; a.out_ld = (size_t)n_r; a.ac_out = want_ac ? r->ac : nullptr;
    if (r->az2) {          // straight into the inverse transform's work array, in its sequence order (the ifftshift of :438)
        a.out = g->work_b; a.out_ld = (size_t)g->ldc; a.out_shift = (n_p - h) % n_p;
    }
    if (n_r > 1 && n_p <= 65535) {
        hipLaunchKernelGGL(rda_rcmc_azcomp_rows_kernel, dim3((n_r + 1023) / 1024, n_p), dim3(256), 0, st, a, want_rc ? r->rc : (cf*)nullptr);
    } else {
        dim3 grid((n_r + 255) / 256, n_p < 16384 ? n_p : 16384);
        hipLaunchKernelGGL(rda_rcmc_azcomp_kernel, grid, dim3(256), 0, st, a, want_rc ? r->rc : (cf*)nullptr);
    }
    GCK(hipGetLastError());
    // 5 ifftshift . IFFT . ifftshift: source row (r + h) mod n both ways; magnitude
    if (r->az2) {
        const ColsSrc src{nullptr, 0, 0, 0, nullptr, 0};
        const ColsDst dst{nullptr, (size_t)n_r, n_p, n_r, nullptr, 1.0f / (float)n_p, (n_p - h) % n_p, mag_out};
        return cols_pow2_ends(g, g->work_b, n_p, true, st, src, dst);
    }
    if (r->pfa72)
        return az_pfa7200_run(true, g->data, (size_t)n_r, n_r, g->work_a, (size_t)g->ldc, nullptr, mag_out, (size_t)n_r, h, h, nullptr,
                              1.0f / (float)n_p, st);
    if (z3) {
        const ColsSrc src{g->data, (size_t)n_r, n_p, n_r, g->az.chirp_i, h};
        const ColsDst dst{nullptr, (size_t)n_r, n_p, n_r, g->az.chirp_i, 1.0f / (float)n_p, (n_p - h) % n_p, mag_out};
        return cols_bluestein3(g, g->work_a, true, st, src, dst, AZ_EPI_CROPOUT_MAG);
    }
    GCK(fft_cols(g, g->data, g->data, true, st, h, nullptr, h));
    hipLaunchKernelGGL(rda_mag_kernel, dim3(4096), dim3(256), 0, st, g->data, mag_out, (size_t)n_p * n_r);
    return hipGetLastError();
}

const float* rda_mag(const Rda* r) { return r->mag; }
const float2* rda_stage(const Rda* r, int which) { return which == 0 ? r->pc : which == 1 ? r->rd : which == 2 ? r->rc : r->ac; }
void rda_axes(const Rda* r, double* range_centered, double* cross_range, double* doppler) {
    double mean = 0;
    for (double v : r->h_r) mean += v;
    mean /= (double)r->n_r;
    if (range_centered) for (int j = 0; j < r->n_r; ++j) range_centered[j] = r->h_r[j] - mean;                 // :443-444
    if (cross_range) for (int i = 0; i < r->n_p; ++i)
        cross_range[i] = r->vr * (((double)i - ((r->n_p % 2 == 0) ? (double)r->n_p / 2 : ((double)r->n_p - 1) / 2)) / r->prf);   // :363-366,442
    if (doppler) for (int i = 0; i < r->n_p; ++i) doppler[i] = r->h_fd[i];
}
uint64_t rda_bytes(const Rda* r) { return r->bytes; }

uint64_t general_csa_bytes(const GeneralCsa* g) { return g->bytes; }
bool general_csa_set_max_slot(GeneralCsa* g, unsigned* slot) {
    if (slot && !(g->rg_mixed && g->pfa)) return false;      // only the 7199 x 13200 route has the reduction in its last launch
    g->max_slot = slot;
    return true;
}

int general_csa_ati_parts(const GeneralCsa* g) { return (g->rg_mixed && g->pfa) ? az_pfa_ati_parts(g->n_rg) : -1; }
int general_csa_set_ati(GeneralCsa* g, const AtiFuse* ati) {
    if (!ati || !ati->s1) { g->ati = AtiFuse{}; return 0; }
    if (!(g->rg_mixed && g->pfa)) return -1;                   // only the 7199 x 13200 route has the epilogue
    g->ati = *ati;
    return az_pfa_ati_parts(g->n_rg);
}

void general_csa_destroy(GeneralCsa* g) {
    if (!g) return;
    az_pfa_destroy(g->pfa);
    axis_free(g->az); axis_free(g->rg);
    hipFree(g->c1); hipFree(g->c2); hipFree(g->c3);
    hipFree(g->data); hipFree(g->work_a); hipFree(g->work_b); hipFree(g->ktab);
    delete g;
}

GeneralCsa* general_csa_create(int n_az, int n_rg, const sarx_radar_params* prm, const float2* tw_all, std::string& err,
                               bool csa_tables, int cus) {
    if (n_az < 2 || n_rg < 2 || n_rg > 32768 || n_az > 32768) { err = "sizes must be in [2, 32768]"; return nullptr; }
    GeneralCsa* g = new GeneralCsa();
    g->n_az = n_az; g->n_rg = n_rg; g->p = *prm; g->tw_all = tw_all;
    g->ldc = (n_rg + 31) / 32 * 32;
    if (cus > 0) g->cus = cus;
    const char* mv = getenv("SARX_RANGE_MIXED");          // SARX_RANGE_MIXED=0 keeps the chirp-z range path (A/B measurements)
    g->rg_mixed = csa_tables && range_mixed_supported(n_rg) && !(mv && atoi(mv) == 0);
    const char* pv = getenv("SARX_AZ_PFA");               // SARX_AZ_PFA=0 keeps the chirp-z azimuth route (A/B measurements)
    if (g->rg_mixed && az_pfa_supported(n_az) && !(pv && atoi(pv) == 0)) {
        hipError_t pe = hipSuccess;
        g->pfa = az_pfa_create((size_t)n_rg, (size_t)g->ldc, (size_t)n_rg, &pe);
        if (!g->pfa) { err = std::string("prime-factor tables: ") + hipGetErrorString(pe); general_csa_destroy(g); return nullptr; }
    }
    auto bail = [&](const char* what, hipError_t e) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        general_csa_destroy(g);
        return (GeneralCsa*)nullptr;
    };
    hipError_t e;
    if (g->pfa) { g->az.n = n_az; g->az.m = n_az; }       // no chirp-z tables along azimuth either
    else if ((e = axis_init(g->az, n_az, 32768, 32768, false)) != hipSuccess) {
        if (e == hipErrorInvalidValue) { err = "a non-power-of-two n_az must be <= 16384 (chirp-z length 32768)"; general_csa_destroy(g); return nullptr; }
        return bail("azimuth tables", e);
    }
    if (g->rg_mixed) { g->rg.n = n_rg; g->rg.m = n_rg; g->rg.direct = true; }     // no chirp-z tables along range
    else if ((e = axis_init(g->rg, n_rg, 65536, 16384, true)) != hipSuccess) return bail("range tables", e);
    // migration factors in natural fftfreq order, any parity (sar_ati_dcpa_sim_csa.py:225,244-249,262)
    const double C0 = 299792458.0, lam = prm->wavelength_m, Kr = prm->chirp_rate_hz_s, Vr = prm->platform_speed_mps,
                 Rref = prm->range_ref_m, fa_step = 1.0 / ((double)n_az * (1.0 / prm->prf_hz));
    std::vector<double2> c1(n_az), c2(n_az), c3(n_az);
    for (int i = 0; i < n_az; ++i) {
        const int ks = (i < (n_az + 1) / 2) ? i : i - n_az;
        const double fa = (double)ks * fa_step, u = lam * fa / (2.0 * Vr);
        double arg = 1.0 - u * u;
        if (arg < 0) arg = 1e-9;
        const double D = sqrt(arg), Cs = 1.0 / D - 1.0;
        c1[i] = make_double2(-0.5 * Kr * Cs, 2.0 * Rref / (C0 * D));
        c2[i] = make_double2(0.5 / (Kr * (1.0 + Cs)), 2.0 * Rref * Cs / C0);
        c3[i] = make_double2(C0 * D / lam, -0.5 * Kr * Cs * (1.0 + Cs));
    }
    const size_t tb = (size_t)n_az * sizeof(double2);
    for (auto pr : {std::make_pair(&g->c1, &c1), std::make_pair(&g->c2, &c2), std::make_pair(&g->c3, &c3)}) {
        if ((e = hipMalloc(pr.first, tb)) != hipSuccess) return bail("hipMalloc tables", e);
        if ((e = hipMemcpy(*pr.first, pr.second->data(), tb, hipMemcpyHostToDevice)) != hipSuccess) return bail("upload tables", e);
    }
    const size_t rows_work = (size_t)n_az * (size_t)g->rg.m;
    const size_t cols_work = (size_t)g->az.m * (size_t)g->ldc;
    g->work_elems = rows_work > cols_work ? rows_work : cols_work;
    if ((e = hipMalloc(&g->data, (size_t)n_az * n_rg * sizeof(cf))) != hipSuccess) return bail("hipMalloc image", e);
    // the prime-factor route needs ONE [n_az x ldc] intermediate; the chirp-z routes two [m_az x ldc] / [n_az x m_rg] work arrays
    const int n_work = g->pfa ? 1 : 2;
    if ((e = hipMalloc(&g->work_a, g->work_elems * sizeof(cf))) != hipSuccess) return bail("hipMalloc work", e);
    if (n_work == 2 && (e = hipMalloc(&g->work_b, g->work_elems * sizeof(cf))) != hipSuccess) return bail("hipMalloc work", e);
    g->bytes = ((size_t)n_az * n_rg + (size_t)n_work * g->work_elems) * sizeof(cf) + 3 * tb;
    // range FFT . Phi_2 . IFFT as one convolution: per-row kernel spectra (SARX_GENERAL_KTAB=0 keeps the two chirp-z transforms)
    const char* ev = getenv("SARX_GENERAL_KTAB");
    if (csa_tables && !g->rg.direct && !(ev && atoi(ev) == 0)) {
        if ((e = hipMalloc(&g->ktab, rows_work * sizeof(cf))) != hipSuccess) return bail("hipMalloc kernel table", e);
        if ((e = build_range_kernel_table(g, nullptr)) != hipSuccess) return bail("range kernel table", e);
        g->bytes += rows_work * sizeof(cf);
    }
    return g;
}

}  // namespace sarx
