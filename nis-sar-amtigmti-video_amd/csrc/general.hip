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
#include <vector>

#include "fft_core.hpp"
#include "phase.hpp"

namespace sarx {

// ---- element-wise kernels --------------------------------------------------------------------------
// out[r][c] = (r < in_rows && c < in_cols ? in[r*in_ld + c] : 0) * rowvec[r] * colvec[c] * scalar
// (in may equal out: each thread reads and writes the same element)
__global__ __launch_bounds__(256) void scale_copy_2d_kernel(const cf* in, int in_rows, int in_cols, size_t in_ld,
                                                            cf* out, int out_rows, int out_cols, size_t out_ld,
                                                            const cf* __restrict__ rowvec, const cf* __restrict__ colvec,
                                                            float scalar) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= out_cols) return;
    const bool cin = c < in_cols;
    cf cv = make_float2(scalar, 0.f);
    if (colvec && cin) { const cf w = colvec[c]; cv = make_float2(w.x * scalar, w.y * scalar); }
    for (int r = blockIdx.y; r < out_rows; r += gridDim.y) {
        cf x = make_float2(0.f, 0.f);
        if (cin && r < in_rows) {
            x = cmul(in[(size_t)r * in_ld + c], cv);
            if (rowvec) x = cmul(x, rowvec[r]);
        }
        out[(size_t)r * out_ld + c] = x;
    }
}

static hipError_t scale_copy(const cf* in, int in_rows, int in_cols, size_t in_ld, cf* out, int out_rows, int out_cols,
                             size_t out_ld, const cf* rowvec, const cf* colvec, float scalar, hipStream_t st) {
    dim3 grid((out_cols + 255) / 256, out_rows < 16384 ? out_rows : 16384);
    hipLaunchKernelGGL(scale_copy_2d_kernel, grid, dim3(256), 0, st, in, in_rows, in_cols, in_ld, out, out_rows, out_cols,
                       out_ld, rowvec, colvec, scalar);
    return hipGetLastError();
}

// Phi_1 / Phi_2 / Phi_3 as a separate element-wise launch, direct fp64 evaluation, any size.
struct PhaseArgs {
    cf* buf;
    const double2* c;      // per azimuth bin
    int n_az, n_rg;
    double dt, t_start, t0, df;
};
template <int WHICH> __global__ __launch_bounds__(256) void phase_mul_kernel(PhaseArgs a) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.n_rg) return;
    const double tau = __dadd_rn(a.t_start, __dmul_rn((double)j, a.dt));
    const int ks = (j < (a.n_rg + 1) / 2) ? j : j - a.n_rg;            // numpy.fft.fftfreq order, any parity
    const double f = (double)ks * a.df;
    for (int i = blockIdx.y; i < a.n_az; i += gridDim.y) {
        const double2 c = a.c[i];
        double p;
        if (WHICH == 1) { const double d = tau - c.y; p = c.x * d * d; }                          // :272
        else if (WHICH == 2) p = f * fma(c.x, f, c.y);                                            // :318-324
        else { const double d = tau - a.t0; p = fma(c.x, tau, c.y * d * d); }                     // :359,375-380
        cf* x = a.buf + (size_t)i * a.n_rg + j;
        *x = cmul(*x, cis_rev(p));
    }
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

struct Axis {
    int n = 0, m = 0;            // length, convolution length (m == n: direct power of two)
    bool direct = false;
    cf *chirp_f = nullptr, *chirp_i = nullptr, *bhat_f = nullptr, *bhat_i = nullptr;   // device
};
static const int SPLIT_A = 128, SPLIT_B = 256;       // 32768 = 128 x 256

static hipError_t upload(const std::vector<zd>& h, cf** d) {
    std::vector<cf> f(h.size());
    for (size_t i = 0; i < h.size(); ++i) f[i] = make_float2((float)h[i].real(), (float)h[i].imag());
    hipError_t e = hipMalloc(d, f.size() * sizeof(cf));
    if (e != hipSuccess) return e;
    return hipMemcpy(*d, f.data(), f.size() * sizeof(cf), hipMemcpyHostToDevice);
}
static hipError_t axis_init(Axis& ax, int n, int m_max) {
    ax.n = n;
    if (is_pow2(n) && n >= 16 && n <= 16384) { ax.direct = true; ax.m = n; return hipSuccess; }
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
    if (m == 32768) {                  // spectrum order of the split line FFT: position k1*256 + k2 holds bin k1 + 128*k2
        std::vector<zd> p(m), pi(m);
        for (int k1 = 0; k1 < SPLIT_A; ++k1)
            for (int k2 = 0; k2 < SPLIT_B; ++k2) {
                p[k1 * SPLIT_B + k2] = b[k1 + SPLIT_A * k2];
                pi[k1 * SPLIT_B + k2] = bi[k1 + SPLIT_A * k2];
            }
        b.swap(p);
        bi.swap(pi);
    }
    hipError_t e;
    if ((e = upload(cf_, &ax.chirp_f)) != hipSuccess) return e;
    if ((e = upload(ci_, &ax.chirp_i)) != hipSuccess) return e;
    if ((e = upload(b, &ax.bhat_f)) != hipSuccess) return e;
    return upload(bi, &ax.bhat_i);
}
static void axis_free(Axis& ax) { hipFree(ax.chirp_f); hipFree(ax.chirp_i); hipFree(ax.bhat_f); hipFree(ax.bhat_i); }

struct GeneralCsa {
    int n_az = 0, n_rg = 0, ldc = 0;
    sarx_radar_params p{};
    const cf* tw_all = nullptr;
    Axis az, rg;
    double2 *c1 = nullptr, *c2 = nullptr, *c3 = nullptr;
    cf *data = nullptr, *work_a = nullptr, *work_b = nullptr;
    size_t work_elems = 0;
    uint64_t bytes = 0;
};

// ---- power-of-two transforms on work arrays ---------------------------------------------------------
static hipError_t rows_pow2(GeneralCsa* g, cf* buf, int rows, int m, bool inv, hipStream_t st) {
    RangeArgs a{};
    hipError_t e;
    if (m <= 16384) {
        a.in = buf; a.out = buf; a.tw = g->tw_all + m; a.inv_n = 1.0f / (float)m; a.n_az = rows;
        return launch_range_pass(m, inv ? RG_IFFT : RG_FFT, a, st);
    }
    // 32768 = 128 x 256 on the [(rows*128) x 256] view of the lines
    AzArgs z{};
    z.in = buf; z.out = buf; z.tw_r = g->tw_all + SPLIT_A; z.n_rg = SPLIT_B; z.tw_scale = 1.0f / 32768.0f;
    z.scale = 1.0f / (float)SPLIT_A;
    z.in_q_stride = SPLIT_A; z.in_m_stride = 1; z.out_q_stride = SPLIT_A; z.out_m_stride = 1;
    a.in = buf; a.out = buf; a.tw = g->tw_all + SPLIT_B; a.inv_n = 1.0f / (float)SPLIT_B; a.n_az = rows * SPLIT_A;
    if (!inv) {
        if ((e = launch_az_tile(SPLIT_A, 32, false, AZ_EPI_TWCOL, z, rows, st)) != hipSuccess) return e;
        return launch_range_pass(SPLIT_B, RG_FFT, a, st);
    }
    if ((e = launch_range_pass(SPLIT_B, RG_IFFT, a, st)) != hipSuccess) return e;
    return launch_az_tile(SPLIT_A, 32, true, AZ_EPI_PROCOL, z, rows, st);
}

// column FFT of length n (power of two, 16..16384) on a [n x ld] array, ld a multiple of 32; in -> out via tmp
static hipError_t cols_pow2(GeneralCsa* g, const cf* in, cf* tmp, cf* out, int n, int ld, bool inv, hipStream_t st) {
    int l2 = 0;
    while ((1 << l2) < n) ++l2;
    const int S = (n <= 128) ? n : (1 << (l2 / 2)), RA = n / S;
    AzArgs a{};
    a.scale = 1.0f / (float)n;
    a.n_rg = ld;
    a.tw_n = g->tw_all + n;
    const int epi_last = inv ? AZ_EPI_SCALE : AZ_EPI_NONE;
    if (S == n) {
        a.in = in; a.out = out; a.tw_r = g->tw_all + n;
        a.in_q_stride = 0; a.in_m_stride = 1; a.out_q_stride = 0; a.out_m_stride = 1;
        return launch_az_tile(n, 32, inv, epi_last, a, 1, st);
    }
    a.in = in; a.out = tmp; a.tw_r = g->tw_all + RA;
    a.in_q_stride = 1; a.in_m_stride = S; a.out_q_stride = 1; a.out_m_stride = S;
    hipError_t e = launch_az_tile(RA, 32, inv, AZ_EPI_TWIDDLE, a, S, st);
    if (e != hipSuccess) return e;
    a.in = tmp; a.out = out; a.tw_r = g->tw_all + S;
    a.in_q_stride = S; a.in_m_stride = 1; a.out_q_stride = 1; a.out_m_stride = RA;
    return launch_az_tile(S, 32, inv, epi_last, a, RA, st);
}

// ---- any-length transforms of the dense [n_az x n_rg] image `d` (in place) ----------------------------------
#define GCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

static hipError_t fft_rows(GeneralCsa* g, cf* d, bool inv, hipStream_t st) {
    const Axis& ax = g->rg;
    const int B = g->n_az, n = g->n_rg, m = ax.m;
    if (ax.direct) return rows_pow2(g, d, B, n, inv, st);
    cf* w = g->work_a;
    GCK(scale_copy(d, B, n, n, w, B, m, m, nullptr, inv ? ax.chirp_i : ax.chirp_f, 1.0f, st));
    GCK(rows_pow2(g, w, B, m, false, st));
    GCK(scale_copy(w, B, m, m, w, B, m, m, nullptr, inv ? ax.bhat_i : ax.bhat_f, 1.0f, st));
    GCK(rows_pow2(g, w, B, m, true, st));
    return scale_copy(w, B, n, m, d, B, n, n, nullptr, inv ? ax.chirp_i : ax.chirp_f, inv ? 1.0f / (float)n : 1.0f, st);
}

static hipError_t fft_cols(GeneralCsa* g, const cf* src, cf* dst, bool inv, hipStream_t st) {
    const Axis& ax = g->az;
    const int n = g->n_az, C = g->n_rg, ld = g->ldc, m = ax.m;
    cf *wa = g->work_a, *wb = g->work_b;
    if (ax.direct) {
        GCK(scale_copy(src, n, C, C, wa, n, ld, ld, nullptr, nullptr, 1.0f, st));
        GCK(cols_pow2(g, wa, wa, wb, n, ld, inv, st));                     // step A in place on wa, result in wb
        return scale_copy(wb, n, C, ld, dst, n, C, C, nullptr, nullptr, 1.0f, st);
    }
    GCK(scale_copy(src, n, C, C, wa, m, ld, ld, inv ? ax.chirp_i : ax.chirp_f, nullptr, 1.0f, st));
    GCK(cols_pow2(g, wa, wa, wb, m, ld, false, st));                       // step A in place on wa, result in wb
    GCK(scale_copy(wb, m, ld, ld, wa, m, ld, ld, inv ? ax.bhat_i : ax.bhat_f, nullptr, 1.0f, st));
    GCK(cols_pow2(g, wa, wa, wb, m, ld, true, st));
    return scale_copy(wb, n, C, ld, dst, n, C, C, inv ? ax.chirp_i : ax.chirp_f, nullptr, inv ? 1.0f / (float)n : 1.0f, st);
}

template <int WHICH> static hipError_t phase(GeneralCsa* g, cf* d, hipStream_t st) {
    PhaseArgs a{};
    a.buf = d; a.c = (WHICH == 1) ? g->c1 : (WHICH == 2) ? g->c2 : g->c3;
    a.n_az = g->n_az; a.n_rg = g->n_rg;
    a.dt = 1.0 / g->p.sample_rate_hz; a.t_start = g->p.t_start_fast_s;
    a.t0 = 2.0 * g->p.range_ref_m / 299792458.0;
    a.df = 1.0 / ((double)g->n_rg * a.dt);
    dim3 grid((g->n_rg + 255) / 256, g->n_az < 16384 ? g->n_az : 16384);
    hipLaunchKernelGGL(phase_mul_kernel<WHICH>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t general_csa_focus(GeneralCsa* g, const float2* d_in, float2* d_out, hipStream_t st) {
    cf* d = g->data;
    GCK(fft_cols(g, d_in, d, false, st));          // :233
    GCK(phase<1>(g, d, st));                       // :272-274
    GCK(fft_rows(g, d, false, st));                // :278
    GCK(phase<2>(g, d, st));                       // :318-326
    GCK(fft_rows(g, d, true, st));                 // :331
    GCK(phase<3>(g, d, st));                       // :359-382
    return fft_cols(g, d, d_out, true, st);        // :385
}

uint64_t general_csa_bytes(const GeneralCsa* g) { return g->bytes; }

void general_csa_destroy(GeneralCsa* g) {
    if (!g) return;
    axis_free(g->az); axis_free(g->rg);
    hipFree(g->c1); hipFree(g->c2); hipFree(g->c3);
    hipFree(g->data); hipFree(g->work_a); hipFree(g->work_b);
    delete g;
}

GeneralCsa* general_csa_create(int n_az, int n_rg, const sarx_radar_params* prm, const float2* tw_all, std::string& err) {
    if (n_az < 2 || n_rg < 2 || n_rg > 16384 || n_az > 16384) { err = "sizes must be in [2, 16384]"; return nullptr; }
    GeneralCsa* g = new GeneralCsa();
    g->n_az = n_az; g->n_rg = n_rg; g->p = *prm; g->tw_all = tw_all;
    g->ldc = (n_rg + 31) / 32 * 32;
    auto bail = [&](const char* what, hipError_t e) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        general_csa_destroy(g);
        return (GeneralCsa*)nullptr;
    };
    hipError_t e;
    if ((e = axis_init(g->az, n_az, 16384)) != hipSuccess) {
        if (e == hipErrorInvalidValue) { err = "a non-power-of-two n_az must be <= 8192 (chirp-z length 16384)"; general_csa_destroy(g); return nullptr; }
        return bail("azimuth tables", e);
    }
    if ((e = axis_init(g->rg, n_rg, 32768)) != hipSuccess) return bail("range tables", e);
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
    if ((e = hipMalloc(&g->work_a, g->work_elems * sizeof(cf))) != hipSuccess) return bail("hipMalloc work", e);
    if ((e = hipMalloc(&g->work_b, g->work_elems * sizeof(cf))) != hipSuccess) return bail("hipMalloc work", e);
    g->bytes = ((size_t)n_az * n_rg + 2 * g->work_elems) * sizeof(cf) + 3 * tb;
    return g;
}

}  // namespace sarx
