// Time-domain back-projection for the VideoSAR batch: tdbp_gpu, sar_batch_sim.py:171-238.
//
//   1 range compression (:180-185): circular correlation of every pulse [num_samples] with the fftshifted
//     reference chirp (int(T_P*FS) taps).  num_samples is 22004 = 4 * 5501 natively, so the correlation
//     runs as overlap-save blocks on the power-of-two line FFT (M <= 32768): block b produces outputs
//     [b*B, b*B + B), B = M - taps + 1, from the wrapped input segment starting at b*B.
//   2 back-projection (:197-236): for every pixel and pulse the two-way delay with the receiver moved by
//     the platform during the flight time (:216-220), the range-Doppler coupling shift (:209-213), a linear
//     interpolation of the compressed pulse and the carrier phase exp(j 2 pi FC tau) (:231), summed over
//     pulses.  Geometry and phase argument in fp64 (FC*tau = 3e7 revolutions); the interpolation
//     coordinate goes through float32 exactly as the reference's F.grid_sample call does (:223-226:
//     normalise, cast, unnormalise with one fused multiply-add), because that quantisation (1e-3 sample)
//     is visible at the 1e-4 parity bar; samples fp32, pulse sum fp64.
//     One thread per pixel (16 x 16 pixel tiles keep a wave's gather inside a few hundred bytes of one
//     pulse), pulses split into chunks across blockIdx.y, partial images reduced in a fixed order.
//     Compute-bound in fp64 (one rsqrt + a short series instead of two square roots and a division):
//     6.6e8 pixel-pulses per 512 x 512 x 2500 frame in 1.66 ms.
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "fft_core.hpp"
#include "general.h"
#include "tdbp.h"

namespace sarx {

typedef std::complex<double> zd;

struct PulseGeo {            // one 64-byte record per pulse, read with scalar loads
    double px, py, pz;       // platform position
    double wx, wy, wz;       // platform velocity - focus velocity  (v_rel, :211)
    double dt, pad;          // t_pulse - mean(t_pulses) (:203-204); pad = |v_rel|^2
};

struct TdbpArgs {
    const cf* rc;            // [n_p][n_s] range-compressed pulses
    const PulseGeo* geo;
    const double *xax, *yax;
    double2* part;           // [chunks][ny*nx]
    double vfx, vfy, vfz;
    double inv_c, fc, fs, t_start, k_shift, inv_ns;
    float half_w;
    int n_p, n_s, nx, ny, per_chunk;
};

__global__ __launch_bounds__(256) void wrap_copy_kernel(const cf* in, int rows, int n, cf* out, int m, int col0) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int s = (col0 + j) % n;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) out[(size_t)r * m + j] = in[(size_t)r * n + s];
}

__global__ __launch_bounds__(256) void tdbp_kernel(TdbpArgs a) {
    const int tiles_x = (a.nx + 15) >> 4;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int ix = tx * 16 + (threadIdx.x & 15), iy = ty * 16 + (threadIdx.x >> 4);
    const bool live = ix < a.nx && iy < a.ny;
    const double gx0 = a.xax[live ? ix : 0], gy0 = a.yax[live ? iy : 0];
    const int p0 = blockIdx.y * a.per_chunk, p1 = min(a.n_p, p0 + a.per_chunk);
    double acc_re = 0.0, acc_im = 0.0;
    for (int p = p0; p < p1; ++p) {
        const PulseGeo g = a.geo[p];
        // pixel moved with the focus velocity (:205), vector to the transmitter (:207-208)
        const double dx = fma(a.vfx, g.dt, gx0) - g.px;
        const double dy = fma(a.vfy, g.dt, gy0) - g.py;
        const double dz = a.vfz * g.dt - g.pz;
        const double s_tx = fma(dx, dx, fma(dy, dy, dz * dz));
        const double inv_d = rsqrt(s_tx);                                                    // one reciprocal square root
        const double d_tx = s_tx * inv_d;                                                    // serves range and unit vector
        const double dw = g.wx * dx + g.wy * dy + g.wz * dz;
        const double v_rad = dw * inv_d;                                                    // :210-212
        const double t_shift = v_rad * a.k_shift;                                           // :213
        const double tau_a = 2.0 * d_tx * a.inv_c;                                          // :215
        // (g + v_f tau_a) - (pos + vel tau_a) = d - v_rel tau_a (:216-218), so
        // d_rx^2 = d_tx^2 (1 - e), e = tau_a (2 d.w - tau_a |w|^2) / d_tx^2 ~ 1e-4: sqrt(1 - e) by its series
        // (e^5 / d_tx < 1e-20 relative; a second fp64 square root costs three times as much)
        const double eps = tau_a * (2.0 * dw - tau_a * g.pad) * (inv_d * inv_d);
        const double sq = fma(eps, fma(eps, fma(eps, fma(eps, -5.0 / 128.0, -1.0 / 16.0), -1.0 / 8.0), -0.5), 1.0);
        const double d_rx = d_tx * sq;
        const double tau = (d_tx + d_rx) * a.inv_c;                                         // :219
        const double idx_f = (tau - a.t_start + t_shift) * a.fs;                            // :221
        const float xn = (float)(2.0 * (idx_f * a.inv_ns) - 1.0);                           // :222, .float() :226
        const float x = __fmaf_rn(xn + 1.0f, a.half_w, -0.5f);                              // grid_sample unnormalise
        const float x0 = floorf(x);
        const float w = x - x0, e = 1.0f - w;
        const int i0 = (int)x0;
        const cf* row = a.rc + (size_t)p * a.n_s;
        cf v0 = make_float2(0.f, 0.f), v1 = v0;
        if (i0 >= 0 && i0 < a.n_s) v0 = row[i0];
        if (i0 + 1 >= 0 && i0 + 1 < a.n_s) v1 = row[i0 + 1];
        const float s_re = fmaf(v1.x, w, v0.x * e), s_im = fmaf(v1.y, w, v0.y * e);
        const cf ph = cis_rev(a.fc * tau);                                                  // :231-232
        acc_re += (double)(s_re * ph.x - s_im * ph.y);
        acc_im += (double)(s_re * ph.y + s_im * ph.x);
    }
    if (live) a.part[(size_t)blockIdx.y * a.nx * a.ny + (size_t)iy * a.nx + ix] = make_double2(acc_re, acc_im);
}

// ------------------------------------------------------------------------------------------------------------
// The same sum with the geometry expanded about the tile's reference pixel (round 5).  A 16 x 16 tile spans ~10 m at
// ranges of hundreds of kilometres, so per (tile, pulse) everything that needs a square root or a division is computed
// ONCE - by one lane per pulse, 256 pulses per batch, handed to the tile through LDS - and a pixel keeps only
//   d_tx = d0 sqrt(1 + u),  u = (2 D.q + |q|^2) / d0^2        as d0 (1 + u/2 - u^2/8 + u^3/16)   (u <= 5e-5: next term 5 u^4 / 128)
//   d_tx - d_rx = E0 + grad E . q                             (E = d_tx (1 - sqrt(1 + 4|w|^2/c^2 - 4 r / c)), r = d.w / d_tx: its
//                                                              leading term (2/c) d.w is exactly linear in q, the rest < 1e-11 m)
//   t_shift fs = ts0 + grad ts . q                            (fp32: it only moves the interpolation coordinate, by < 1e-6 sample)
// with q the pixel's offset from the reference pixel (exact differences of the axis tables).  22 fp64 instructions per
// pixel and pulse instead of 48; the launcher takes this kernel only when 5 u_max^4 / 128 * d_max < 1e-9 m over all
// pulses (tdbp_tile_ok), otherwise the exact kernel above.  Same pulse order per pixel and the same fp64 accumulation.
// ------------------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) TilePulse {
    double a1, a2;           // 2 Dx, 2 Dy
    double inv_d0sq, d0;
    double e0, ex;
    double ey; float ts0, tsx;
    float tsy, pad0; double pad1;
};
static_assert(sizeof(TilePulse) == 80, "five 16-byte LDS reads per pulse");
static constexpr int TP_BATCH = 256;

__global__ __launch_bounds__(256) void tdbp_tile_kernel(TdbpArgs a) {
    __shared__ TilePulse s_tp[TP_BATCH];
    const int tiles_x = (a.nx + 15) >> 4;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int ix = tx * 16 + (threadIdx.x & 15), iy = ty * 16 + (threadIdx.x >> 4);
    const bool live = ix < a.nx && iy < a.ny;
    const double xc = a.xax[min(tx * 16 + 8, a.nx - 1)], yc = a.yax[min(ty * 16 + 8, a.ny - 1)];
    const double qx = a.xax[live ? ix : 0] - xc, qy = a.yax[live ? iy : 0] - yc;
    const double q2 = fma(qx, qx, qy * qy);
    const float qxf = (float)qx, qyf = (float)qy;
    const int p0 = blockIdx.y * a.per_chunk, p1 = min(a.n_p, p0 + a.per_chunk);
    const double two_inv_ns = 2.0 * a.inv_ns;
    double acc_re = 0.0, acc_im = 0.0;
    for (int pb = p0; pb < p1; pb += TP_BATCH) {
        const int nb = min(TP_BATCH, p1 - pb);
        __syncthreads();                                   // the previous batch has been consumed
        if ((int)threadIdx.x < nb) {
            const PulseGeo g = a.geo[pb + threadIdx.x];
            const double dx = fma(a.vfx, g.dt, xc) - g.px, dy = fma(a.vfy, g.dt, yc) - g.py, dz = a.vfz * g.dt - g.pz;   // :205-208 at the reference pixel
            const double s0 = fma(dx, dx, fma(dy, dy, dz * dz));
            const double inv_d = rsqrt(s0), d0 = s0 * inv_d;
            const double dw = g.wx * dx + g.wy * dy + g.wz * dz;
            const double r = dw * inv_d;                                                         // :210-212
            const double tau_a = 2.0 * d0 * a.inv_c;
            const double eps = tau_a * (2.0 * dw - tau_a * g.pad) * (inv_d * inv_d);
            const double sm1 = eps * fma(eps, fma(eps, fma(eps, -5.0 / 128.0, -1.0 / 16.0), -1.0 / 8.0), -0.5);   // sqrt(1 - eps) - 1
            const double ux = dx * inv_d, uy = dy * inv_d;                                      // unit vector, ground components
            const double f = -sm1, fp = 2.0 * a.inv_c / (1.0 + sm1);                            // F(r) and F'(r)
            const double rx = (g.wx - r * ux) * inv_d, ry = (g.wy - r * uy) * inv_d;            // grad r
            TilePulse t;
            t.a1 = 2.0 * dx; t.a2 = 2.0 * dy; t.inv_d0sq = inv_d * inv_d; t.d0 = d0;
            t.e0 = d0 * f;
            t.ex = fma(d0 * fp, rx, f * ux); t.ey = fma(d0 * fp, ry, f * uy);
            const double ks = a.k_shift * a.fs;
            t.ts0 = (float)(r * ks); t.tsx = (float)(rx * ks); t.tsy = (float)(ry * ks);
            t.pad0 = 0.f; t.pad1 = 0.0;
            s_tp[threadIdx.x] = t;
        }
        __syncthreads();
        for (int k = 0; k < nb; ++k) {
            const TilePulse& t = s_tp[k];
            const double u = fma(t.a1, qx, fma(t.a2, qy, q2)) * t.inv_d0sq;
            const double h = u * fma(u, fma(u, 1.0 / 16.0, -1.0 / 8.0), 0.5);
            const double d_tx = fma(t.d0, h, t.d0);
            const double e = fma(t.ex, qx, fma(t.ey, qy, t.e0));                                // d_tx - d_rx
            const double tau = fma(2.0, d_tx, -e) * a.inv_c;                                   // :219
            const float ts = fmaf(t.tsx, qxf, fmaf(t.tsy, qyf, t.ts0));                         // t_shift * fs (:213)
            const double idx_f = fma(tau - a.t_start, a.fs, (double)ts);                        // :221
            const float xn = (float)fma(idx_f, two_inv_ns, -1.0);                               // :222, .float() :226
            const float x = __fmaf_rn(xn + 1.0f, a.half_w, -0.5f);                              // grid_sample unnormalise
            const float x0 = floorf(x);
            const float w = x - x0, ew = 1.0f - w;
            const int i0 = (int)x0;
            const cf* row = a.rc + (size_t)(pb + k) * a.n_s;
            cf v0 = make_float2(0.f, 0.f), v1 = v0;
            if (i0 >= 0 && i0 < a.n_s) v0 = row[i0];
            if (i0 + 1 >= 0 && i0 + 1 < a.n_s) v1 = row[i0 + 1];
            const float s_re = fmaf(v1.x, w, v0.x * ew), s_im = fmaf(v1.y, w, v0.y * ew);
            const cf ph = cis_rev(a.fc * tau);                                                  // :231-232
            acc_re += (double)(s_re * ph.x - s_im * ph.y);
            acc_im += (double)(s_re * ph.y + s_im * ph.x);
        }
    }
    if (live) a.part[(size_t)blockIdx.y * a.nx * a.ny + (size_t)iy * a.nx + ix] = make_double2(acc_re, acc_im);
}

__global__ __launch_bounds__(256) void tdbp_reduce_kernel(const double2* part, int chunks, size_t n_pix, double2* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pix) return;
    double re = 0.0, im = 0.0;
    for (int c = 0; c < chunks; ++c) {
        const double2 v = part[(size_t)c * n_pix + i];
        re += v.x;
        im += v.y;
    }
    out[i] = make_double2(re, im);
}

struct Tdbp {
    int n_p = 0, n_s = 0, nx = 0, ny = 0, taps = 0, m = 0, blk = 0, chunks = 1, per_chunk = 0;
    sarx_tdbp_params k{};
    const cf* tw_all = nullptr;
    cf *work = nullptr, *rc = nullptr;
    std::map<int, cf*> hhat;      // conj spectrum of the reference chirp per FFT length (device order)
    int l_ref = 0;
    bool full_rc = false;         // SARX_TDBP_FULL_RC=1: always compress every sample
    bool three_launch = false;    // SARX_TDBP_RC_FUSED=0: copy-in, two transforms, copy-out per block (the form of rounds 2-4, for A/B)
    int win_lo = 0, win_hi = 0;   // samples compressed by the last call
    // per-pulse table, double-buffered: frame i + 1's table is staged in page-locked memory and copied on the stream while frame i
    // still reads its own - no stream synchronisation per frame (the copy of rounds 2-4 was a blocking one behind a hipStreamSynchronize
    // that made the host wait for the frame's echo, noise and range compression before it could enqueue the back-projection)
    PulseGeo* geo[2] = {nullptr, nullptr};
    PulseGeo* h_geo[2] = {nullptr, nullptr};
    hipEvent_t geo_done[2] = {nullptr, nullptr};
    int geo_slot = 0;
    double *xax = nullptr, *yax = nullptr;
    double axes_scene = -1.0;     // scene_size the pixel axes on the device were built for
    double2 *part = nullptr, *img = nullptr;
    uint64_t bytes = 0;
};

void tdbp_destroy(Tdbp* t) {
    if (!t) return;
    for (auto& kv : t->hhat) hipFree(kv.second);
    hipFree(t->work); hipFree(t->rc); hipFree(t->xax); hipFree(t->yax);
    for (int b = 0; b < 2; ++b) {
        hipFree(t->geo[b]);
        if (t->h_geo[b]) hipHostFree(t->h_geo[b]);
        if (t->geo_done[b]) hipEventDestroy(t->geo_done[b]);
    }
    hipFree(t->part); hipFree(t->img);
    delete t;
}

#define TCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

// conj(FFT_m(fftshift(ref_chirp))) (:177-179) for an m-point transform, cached; linspace endpoints as torch builds
// them (two-sided)
static hipError_t filter_spectrum(Tdbp* t, int m, cf** out) {
    auto it = t->hhat.find(m);
    if (it != t->hhat.end()) { if (out) *out = it->second; return hipSuccess; }
    const int l_ref = t->l_ref;
    const sarx_tdbp_params* k = &t->k;
    std::vector<zd> h(m, zd(0, 0));
    const double step = l_ref > 1 ? k->t_p / (double)(l_ref - 1) : 0.0;
    for (int i = 0; i < t->taps; ++i) {
        const int src = ((i - l_ref / 2) % l_ref + l_ref) % l_ref;     // fftshift: out[i] = ref[(i - L//2) mod L]
        const double tt = src < l_ref / 2 ? -k->t_p / 2 + step * (double)src : k->t_p / 2 - step * (double)(l_ref - 1 - src);
        h[i] = std::polar(1.0, M_PI * k->k_rate * tt * tt);
    }
    host_fft_pow2(h);
    for (auto& v : h) v = std::conj(v);
    if (m == 32768) to_split_order(h);
    std::vector<cf> hf(m);
    for (int i = 0; i < m; ++i) hf[i] = make_float2((float)h[i].real(), (float)h[i].imag());
    cf* d = nullptr;
    TCK(hipMalloc(&d, (size_t)m * sizeof(cf)));
    hipError_t e = hipMemcpy(d, hf.data(), (size_t)m * sizeof(cf), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(d); return e; }
    t->hhat[m] = d;
    if (out) *out = d;
    return hipSuccess;
}

Tdbp* tdbp_create(int n_pulses, int num_samples, int nx, int ny, const sarx_tdbp_params* k, const float2* tw_all,
                  std::string& err) {
    const int l_ref = (int)(k->t_p * k->fs);                           // int(T_P * FS), :177
    if (l_ref < 1) { err = "T_P * FS must be >= 1 sample"; return nullptr; }
    const int taps = l_ref < num_samples ? l_ref : num_samples;       // fft(..., n=num_samples) truncates
    int m = 16;
    while (m < num_samples + taps - 1 && m < 32768) m <<= 1;
    if (m - taps + 1 < m / 2) { err = "reference chirp longer than 16385 samples"; return nullptr; }
    Tdbp* t = new Tdbp();
    t->n_p = n_pulses; t->n_s = num_samples; t->nx = nx; t->ny = ny; t->taps = taps; t->m = m; t->blk = m - taps + 1;
    t->k = *k; t->tw_all = tw_all;
    const size_t n_pix = (size_t)nx * ny;
    int chunks = (int)(((size_t)1 << 20) / (n_pix ? n_pix : 1));
    if (chunks > n_pulses / 32) chunks = n_pulses / 32;
    if (chunks > 64) chunks = 64;
    if (chunks < 1) chunks = 1;
    t->per_chunk = (n_pulses + chunks - 1) / chunks;
    t->chunks = (n_pulses + t->per_chunk - 1) / t->per_chunk;
    auto bail = [&](const char* what, hipError_t e) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        tdbp_destroy(t);
        return (Tdbp*)nullptr;
    };
    hipError_t e;
    t->l_ref = l_ref;
    if (const char* ev = getenv("SARX_TDBP_FULL_RC")) t->full_rc = atoi(ev) != 0;
    if (const char* ev = getenv("SARX_TDBP_RC_FUSED")) t->three_launch = atoi(ev) == 0;
    if ((e = filter_spectrum(t, m, nullptr)) != hipSuccess) return bail("filter spectrum", e);
    if ((e = hipMalloc(&t->work, (size_t)n_pulses * m * sizeof(cf))) != hipSuccess) return bail("hipMalloc work", e);
    if ((e = hipMalloc(&t->rc, (size_t)n_pulses * num_samples * sizeof(cf))) != hipSuccess) return bail("hipMalloc", e);
    for (int b = 0; b < 2; ++b) {
        if ((e = hipMalloc(&t->geo[b], (size_t)n_pulses * sizeof(PulseGeo))) != hipSuccess) return bail("hipMalloc", e);
        if ((e = hipHostMalloc(&t->h_geo[b], (size_t)n_pulses * sizeof(PulseGeo), hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
        if ((e = hipEventCreateWithFlags(&t->geo_done[b], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    }
    if ((e = hipMalloc(&t->xax, (size_t)nx * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&t->yax, (size_t)ny * sizeof(double))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&t->part, (size_t)t->chunks * n_pix * sizeof(double2))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&t->img, n_pix * sizeof(double2))) != hipSuccess) return bail("hipMalloc", e);
    t->bytes = (uint64_t)n_pulses * (m + num_samples) * sizeof(cf) + (uint64_t)(t->chunks + 1) * n_pix * sizeof(double2);
    return t;
}

// outputs [n0, n0 + cnt) of the circular correlation through one m-point transform (cnt <= m - taps + 1)
static hipError_t compress_block(Tdbp* t, const float2* raw, int n0, int cnt, int m, hipStream_t st) {
    cf* hhat = nullptr;
    TCK(filter_spectrum(t, m, &hhat));
    const int n = t->n_s;
    if (m <= 16384 && !t->three_launch) {
        // one launch: the wrapped m-sample segment read straight from the pulse, FFT . conj(reference spectrum) . IFFT in
        // registers / LDS, the cnt wanted outputs written to their place (range_pass_kernel<m, RG_CONV>): 0.42 -> 0.185 ms per
        // 2500 x 22004 frame against copy-in, two transforms and copy-out (profiles/r05_bj_*)
        RangeArgs ca{};
        ca.in = raw; ca.out = t->rc + n0; ca.tw = t->tw_all + m; ca.n_az = t->n_p; ca.inv_n = 1.0f / (float)m;
        ca.mulvec = hhat; ca.mul_period = 1;
        ca.conv_valid = m; ca.conv_crop0 = 0; ca.conv_out = cnt; ca.conv_in_ld = (size_t)n; ca.conv_out_ld = (size_t)n;
        ca.conv_wrap_n = n; ca.conv_wrap_c0 = n0 % n;
        return launch_range_pass(m, RG_CONV, ca, st);
    }
    dim3 grid((m + 255) / 256, t->n_p < 8192 ? t->n_p : 8192);
    hipLaunchKernelGGL(wrap_copy_kernel, grid, dim3(256), 0, st, raw, t->n_p, n, t->work, m, n0);
    TCK(hipGetLastError());
    TCK(line_fft_pow2(t->tw_all, t->work, t->n_p, m, false, st, hhat));      // * conj(reference spectrum) in the epilogue
    TCK(line_fft_pow2(t->tw_all, t->work, t->n_p, m, true, st));
    return scale_copy_cols(t->work, t->n_p, cnt, m, t->rc + n0, t->n_p, cnt, n, nullptr, 1.0f, st);
}

// raw: device [n_p][n_s] complex64.  Leaves the range-compressed samples [lo, hi) of every pulse in t->rc
// (the back-projection of one scene reads a narrow window of each pulse: ~1700 of 22004 samples natively).
hipError_t tdbp_range_compress(Tdbp* t, const float2* raw, int lo, int hi, hipStream_t st) {
    const int n = t->n_s;
    if (lo < 0) lo = 0;
    if (hi > n) hi = n;
    t->win_lo = lo; t->win_hi = hi;
    if (hi <= lo) return hipSuccess;
    int m = 16;
    while (m < (hi - lo) + t->taps - 1) m <<= 1;
    if (m < t->m) return compress_block(t, raw, lo, hi - lo, m, st);        // one transform shorter than the plan's
    for (int n0 = lo; n0 < hi; n0 += t->blk) {
        const int cnt = (hi - n0 < t->blk) ? hi - n0 : t->blk;
        TCK(compress_block(t, raw, n0, cnt, t->m, st));
    }
    return hipSuccess;
}

// Sample window [lo, hi) that the back-projection can touch: a bound, not an estimate.  Per pulse the pixel rectangle
// moved by v_f dt is convex, so the transmit range is largest at a corner and smallest at the projection of the
// platform clamped to the rectangle; |d_rx - d_tx| <= |v_rel| tau_a and |t_shift| <= |v_rel| |k_shift| (:209-219).
static void sample_window(const Tdbp* t, const std::vector<PulseGeo>& geo, const double* vf, double t_start, double scene_size,
                          int* lo, int* hi) {
    const double c = t->k.c, fs = t->k.fs, ks = fabs(t->k.fc * 2.0 / t->k.c / t->k.k_rate), h = scene_size / 2;
    double imin = 1e300, imax = -1e300;
    for (const PulseGeo& g : geo) {
        const double sx = vf[0] * g.dt, sy = vf[1] * g.dt, gz = vf[2] * g.dt - g.pz;
        const double qx = g.px - sx, qy = g.py - sy;                    // platform relative to the moved rectangle
        const double cx = qx < -h ? -h : (qx > h ? h : qx), cy = qy < -h ? -h : (qy > h ? h : qy);
        const double dmin = sqrt((cx - qx) * (cx - qx) + (cy - qy) * (cy - qy) + gz * gz);
        const double ax = fabs(qx) + h, ay = fabs(qy) + h;              // farthest corner
        const double dmax = sqrt(ax * ax + ay * ay + gz * gz);
        const double w = sqrt(g.wx * g.wx + g.wy * g.wy + g.wz * g.wz);
        const double slack = w * (2.0 * dmax / c);
        const double a = ((2.0 * dmin - slack) / c - t_start - w * ks) * fs;
        const double b = ((2.0 * dmax + slack) / c - t_start + w * ks) * fs;
        if (a < imin) imin = a;
        if (b > imax) imax = b;
    }
    // x = idx - 0.5 rounded through float32 (< 1 sample at any n_s <= 2^23), i0 = floor(x), i0 + 1 is read too
    const double l = floor(imin) - 3.0, u = ceil(imax) + 4.0;
    *lo = l < 0 ? 0 : (l > (double)t->n_s ? t->n_s : (int)l);
    *hi = u < 0 ? 0 : (u > (double)t->n_s ? t->n_s : (int)u);
}

// May the tile kernel stand in for the exact one?  Farthest pixel of a 16 x 16 tile from its reference pixel (index 8 of 16) against
// the nearest range of any pulse: the first omitted series term of d_tx, 5 u^4 / 128 * d, must stay below 1e-9 m (4e-7 rad of
// carrier phase at X band).  SARX_TDBP_TILE=0 forces the exact kernel (A/B).
static bool tdbp_tile_ok(const Tdbp* t, const std::vector<PulseGeo>& geo, const double* vf, double scene_size) {
    if (const char* ev = getenv("SARX_TDBP_TILE")) if (atoi(ev) == 0) return false;
    if (t->nx < 2 || t->ny < 2) return false;
    const double h = scene_size / 2;
    const double qm = 8.0 * hypot(scene_size / (double)(t->nx - 1), scene_size / (double)(t->ny - 1));
    for (const PulseGeo& g : geo) {
        const double sx = vf[0] * g.dt, sy = vf[1] * g.dt, gz = vf[2] * g.dt - g.pz;
        const double qx = g.px - sx, qy = g.py - sy;
        const double cx = qx < -h ? -h : (qx > h ? h : qx), cy = qy < -h ? -h : (qy > h ? h : qy);
        const double dmin = sqrt((cx - qx) * (cx - qx) + (cy - qy) * (cy - qy) + gz * gz);
        const double ax = fabs(qx) + h, ay = fabs(qy) + h;
        const double dmax = sqrt(ax * ax + ay * ay + gz * gz);
        if (!(dmin > 0.0)) return false;
        const double u = 2.0 * qm / dmin + (qm / dmin) * (qm / dmin);
        if (!(5.0 / 128.0 * u * u * u * u * dmax < 1e-9)) return false;
    }
    return true;
}

static void linspace(double a, double b, int n, std::vector<double>& out) {     // numpy.linspace (:173-174)
    out.resize(n);
    const double step = n > 1 ? (b - a) / (double)(n - 1) : 0.0;
    for (int i = 0; i < n; ++i) out[i] = (double)i * step + a;
    if (n > 1) out[n - 1] = b;
}

// raw: device [n_p][n_s]; pos, vel: host [n_p][3]; t_pulses: host [n_p].  Result in t->img (device, complex128 [ny][nx]).
// all_samples: compress every sample of every pulse (the caller wants rc_data), else only the window the scene can touch
hipError_t tdbp_focus(Tdbp* t, const float2* raw, const double* pos, const double* vel, const double* t_pulses, double t_start,
                      const double* vel_focus, double scene_size, bool all_samples, hipStream_t st) {
    const int slot = (t->geo_slot ^= 1);
    TCK(hipEventSynchronize(t->geo_done[slot]));           // the focus two calls ago has finished with this slot (returns at once as a rule)
    std::vector<PulseGeo> geo(t->n_p);
    double mean = 0.0;
    for (int p = 0; p < t->n_p; ++p) mean += t_pulses[p];
    mean /= (double)t->n_p;
    for (int p = 0; p < t->n_p; ++p) {
        PulseGeo& g = geo[p];
        g.px = pos[3 * p]; g.py = pos[3 * p + 1]; g.pz = pos[3 * p + 2];
        g.wx = vel[3 * p] - vel_focus[0]; g.wy = vel[3 * p + 1] - vel_focus[1]; g.wz = vel[3 * p + 2] - vel_focus[2];
        g.dt = t_pulses[p] - mean; g.pad = g.wx * g.wx + g.wy * g.wy + g.wz * g.wz;
    }
    int lo = 0, hi = t->n_s;
    if (!all_samples && !t->full_rc) sample_window(t, geo, vel_focus, t_start, scene_size, &lo, &hi);
    TCK(tdbp_range_compress(t, raw, lo, hi, st));
    memcpy(t->h_geo[slot], geo.data(), geo.size() * sizeof(PulseGeo));
    TCK(hipMemcpyAsync(t->geo[slot], t->h_geo[slot], geo.size() * sizeof(PulseGeo), hipMemcpyHostToDevice, st));
    if (t->axes_scene != scene_size) {                    // pixel axes: once per scene size (a frame loop keeps it)
        std::vector<double> xa, ya;
        linspace(-scene_size / 2, scene_size / 2, t->nx, xa);
        linspace(-scene_size / 2, scene_size / 2, t->ny, ya);
        TCK(hipStreamSynchronize(st));                     // an earlier focus may still read the old axes
        TCK(hipMemcpy(t->xax, xa.data(), xa.size() * sizeof(double), hipMemcpyHostToDevice));
        TCK(hipMemcpy(t->yax, ya.data(), ya.size() * sizeof(double), hipMemcpyHostToDevice));
        t->axes_scene = scene_size;
    }
    TdbpArgs a{};
    a.rc = t->rc; a.geo = t->geo[slot]; a.xax = t->xax; a.yax = t->yax; a.part = t->part;
    a.vfx = vel_focus[0]; a.vfy = vel_focus[1]; a.vfz = vel_focus[2];
    a.inv_c = 1.0 / t->k.c; a.fc = t->k.fc; a.fs = t->k.fs; a.t_start = t_start;
    a.k_shift = -t->k.fc * 2.0 / t->k.c / t->k.k_rate;
    a.inv_ns = 1.0 / (double)t->n_s; a.half_w = (float)t->n_s / 2.0f;
    a.n_p = t->n_p; a.n_s = t->n_s; a.nx = t->nx; a.ny = t->ny; a.per_chunk = t->per_chunk;
    const int tiles = ((t->nx + 15) / 16) * ((t->ny + 15) / 16);
    if (tdbp_tile_ok(t, geo, vel_focus, scene_size)) hipLaunchKernelGGL(tdbp_tile_kernel, dim3(tiles, t->chunks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(tdbp_kernel, dim3(tiles, t->chunks), dim3(256), 0, st, a);
    TCK(hipGetLastError());
    const size_t n_pix = (size_t)t->nx * t->ny;
    hipLaunchKernelGGL(tdbp_reduce_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, st, t->part, t->chunks, n_pix, t->img);
    TCK(hipGetLastError());
    return hipEventRecord(t->geo_done[slot], st);
}

const double2* tdbp_image(const Tdbp* t) { return t->img; }
const float2* tdbp_rc(const Tdbp* t) { return t->rc; }
void tdbp_window(const Tdbp* t, int* lo, int* hi) { *lo = t->win_lo; *hi = t->win_hi; }
uint64_t tdbp_bytes(const Tdbp* t) { return t->bytes; }

}  // namespace sarx
