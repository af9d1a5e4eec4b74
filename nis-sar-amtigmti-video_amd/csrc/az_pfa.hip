// Azimuth transforms of the reference's native pulse count without chirp-z: 7199 = 23 * 313 (sar_ati_dcpa_sim_csa.py:47,
// 402-403: 7200 pulses minus the DPCA shift), both factors prime and coprime.
//
//   Good-Thomas (prime-factor) map, no twiddles between the factors:
//       n = (313 n1 + 23 n2) mod 7199,      k = (c1 k1 + c2 k2) mod 7199,  c1 = 313 (313^-1 mod 23), c2 = 23 (23^-1 mod 313)
//       X[k] = sum_n1 W_23^(n1 k1) [ sum_n2 x[n] W_313^(n2 k2) ]
//   launch 1 (pfa_rader313_kernel): the 313-point transforms over n2, one (n1, column tile) per workgroup, by Rader's
//       algorithm: with g a primitive root mod 313,  y[g^-m] = a[0] + (a' (*) w')[m],  a'[q] = a[g^q],  w'[q] = W_313^(g^-q),
//       a cyclic convolution of length 312 = 24 * 13 done as FFT_312 . spectrum of w' . IFFT_312 on a [312 x W] LDS
//       image (two Stockham stages each way, in-register DFT-24 / DFT-13 of fft_mixed.hpp); y[0] = a[0] + DC bin.
//       The permutations g^q / g^-m are folded into the row addresses of the global loads and stores, so the image makes
//       one HBM round trip: rows of the dense source in, rows n1*313 + k2 of the intermediate out.
//   launch 2 (pfa_dft23_kernel): the 23-point transforms over n1 entirely in registers (one thread per (k2, column)),
//       the output row permutation k(k1, k2), and the epilogue (Phi_1, sar_ati_dcpa_sim_csa.py:262-274, or the 1/N of
//       the inverse) on the way out.
// Two HBM round trips of the unpadded [7199 x n_rg] image per transform; the chirp-z route over 16384 rows moved 5.5x
// the bytes in three launches.
#include <cmath>
#include <complex>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "csa_kernels.h"
#include "fft_mixed.hpp"
#include "phase.hpp"
#include "ati_pixel.hpp"

namespace sarx {

typedef float nt_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf ldnt(const char* p, bool nt) {
    if (nt) { const nt_v2f v = __builtin_nontemporal_load(reinterpret_cast<const nt_v2f*>(p)); return make_float2(v.x, v.y); }
    return *reinterpret_cast<const cf*>(p);
}
__device__ __forceinline__ void stnt(char* p, cf x, bool nt) {
    if (nt) __builtin_nontemporal_store(nt_v2f{x.x, x.y}, reinterpret_cast<nt_v2f*>(p));
    else *reinterpret_cast<cf*>(p) = x;
}

namespace pfa {
constexpr int N = 7199, N1 = 23, P = 313, L = 312, RA = 24, RB = 13;
static_assert(N1 * P == N && RA * RB == L, "factorisation");
}  // namespace pfa

// 313-point DFTs along rows n = (313 n1 + 23 n2) mod 7199 of a.in, column tile of W samples; result rows n1*313 + k2 of a.u.
// One (n1, column tile) per workgroup.  Row addresses are 32-bit byte offsets (the image is < 4 GB): n1 * (313 rows) + a
// per-q table entry kept in LDS, wrapped at 7199 rows, added to a uniform base pointer.
// Time budget at 7199 x 13200 (ablation builds, profiles/r02_pfa_ablation.txt): loads + exchanges + stores alone 0.31 ms
// (5.0 TB/s), arithmetic alone 0.20 ms, together 0.46 ms - with one 768-thread workgroup per CU (78 KiB image, 101 VGPRs)
// the two do not overlap.  Tried and dropped: 13 threads per column with two butterflies each so that two 416-thread
// workgroups fit a CU (0.72 ms: the serial work per thread doubles), 16-column tiles (two workgroups per CU, 0.46 ms),
// prefetching the next n1's rows into a second register set (168 VGPRs + 292 B of scratch per lane), and a persistent form
// that stages the next tile in 26 VGPRs and passes it through LDS (tools/rader_prefetch.patch: 0.432-0.437 vs 0.445 ms; with the
// arithmetic alone at 0.268 ms, loads add 0.05, stores 0.06, both 0.17 - tools/overlapbench.hip shows the same for any
// workgroup that runs its waves in lockstep on 128 KiB tiles).
// TWO: two workgroups per CU, so that one's butterflies run during the other's loads and stores (with one 768-thread workgroup the
// waves are in lockstep and the two add up, see above).  The [312 x W] image alone is 78 KiB, so the tables shrink to 16-bit row
// numbers (1.2 KiB) and the spectrum of w' comes through the scalar cache (a wave holds two butterflies j, j + 1: one 16-byte
// scalar load per r, picked per half-wave); six waves per SIMD means 80 VGPRs, which the last 24-point butterfly only fits when it
// is staged by hand (below).
template <int W, bool TWO>
__device__ __forceinline__ void rader313_body(const PfaArgs& a, cf* lds) {
    using namespace pfa;
    typedef typename std::conditional<TWO, unsigned short, unsigned>::type tab_t;
    tab_t* tin = reinterpret_cast<tab_t*>(lds + (size_t)L * W);          // 23 g^q rows of the source: bytes, or row numbers (TWO)
    tab_t* tout = tin + L;                                               // g^-m rows of the intermediate
    cf* bsp = reinterpret_cast<cf*>(tout + L);                           // !TWO: spectrum of w': read mid-tile, and a global load there would
                                                                         // make the in-order vmcnt wait for everything issued before it
    const unsigned c = threadIdx.x % W, j = threadIdx.x / W;     // j in [0, 24)
    const unsigned col = blockIdx.x * W + c, n1 = blockIdx.y;
    const bool live = (int)col < a.in_cols;
    const unsigned colb = col * (unsigned)sizeof(cf);
    const unsigned pin = a.off0in / (unsigned)P, pu = a.off0u / (unsigned)P;    // bytes per row
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 bq[TWO ? RB : 1];
    if constexpr (TWO) {
        static_assert(!TWO || (W == 32 && RB == 13), "a wave holds butterflies j = 2w, 2w + 1");
        for (int i = threadIdx.x; i < L; i += RA * W) { tin[i] = (tab_t)(a.offin[i] / pin); tout[i] = (tab_t)(a.offu[i] / pu); }
        const unsigned long long pv = (unsigned long long)(a.bspec + 2 * (threadIdx.x / 64));
        const unsigned long long ps = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pv);
        // bins j + 24 r, r = 0..12: 192 bytes apart
        asm volatile("s_load_dwordx4 %0, %13, 0x0\n\ts_load_dwordx4 %1, %13, 0xc0\n\ts_load_dwordx4 %2, %13, 0x180\n\t"
                     "s_load_dwordx4 %3, %13, 0x240\n\ts_load_dwordx4 %4, %13, 0x300\n\ts_load_dwordx4 %5, %13, 0x3c0\n\t"
                     "s_load_dwordx4 %6, %13, 0x480\n\ts_load_dwordx4 %7, %13, 0x540\n\ts_load_dwordx4 %8, %13, 0x600\n\t"
                     "s_load_dwordx4 %9, %13, 0x6c0\n\ts_load_dwordx4 %10, %13, 0x780\n\ts_load_dwordx4 %11, %13, 0x840\n\t"
                     "s_load_dwordx4 %12, %13, 0x900\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(bq[0]), "=&s"(bq[1]), "=&s"(bq[2]), "=&s"(bq[3]), "=&s"(bq[4]), "=&s"(bq[5]), "=&s"(bq[6]), "=&s"(bq[7]),
                       "=&s"(bq[8]), "=&s"(bq[9]), "=&s"(bq[10]), "=&s"(bq[11]), "=&s"(bq[12])
                     : "s"(ps) : "memory");
    } else {
        for (int i = threadIdx.x; i < L; i += RA * W) { tin[i] = a.offin[i]; tout[i] = a.offu[i]; bsp[i] = a.bspec[i]; }
    }
    __syncthreads();
    const char* __restrict__ src = reinterpret_cast<const char*>(a.in);
    char* __restrict__ dst = reinterpret_cast<char*>(a.u);
    const unsigned wrap = (unsigned)N * pin;                              // bytes of 7199 source rows

    cf v[RA];
    cf a0 = make_float2(0.f, 0.f), y0 = a0;
    // forward FFT_312, stage 1: radix 24 on a'[j + 13 r], butterflies j < 13, straight from HBM
    if (j < RB) {
        const unsigned b = n1 * a.off0in;
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            unsigned o = b + (TWO ? __umul24((unsigned)tin[j + RB * r], pin) : (unsigned)tin[j + RB * r]);
            if (o >= wrap) o -= wrap;
            v[r] = live ? ldnt(src + (o + colb), TWO || a.nt) : make_float2(0.f, 0.f);
        }
        if (j == 0 && live) a0 = *reinterpret_cast<const cf*>(src + (b + colb));          // n2 = 0: row 313 n1
        mix::dft_any<RA, false>(v);
#pragma unroll
        for (int r = 0; r < RA; ++r) lds[(j * RA + r) * W + c] = v[r];
    }
    __syncthreads();
    // stage 2: radix 13 on y[j + 24 r], twiddle W_312^(j r); thread j ends with spectrum bins j + 24 r
#pragma unroll
    for (int r = 0; r < RB; ++r) v[r] = lds[(j + RA * r) * W + c];
    mix::apply_powers<RB>(v, cis_frac(-(float)j * (1.0f / (float)L)));
    mix::dft_any<RB, false>(v);
    // times the spectrum of w' (carries the 1/312 of the convolution); the DC bin also gives y[0] and takes a[0]
    if (j == 0) y0 = cadd(a0, v[0]);
    if constexpr (TWO) {
        const bool odd = (threadIdx.x & 32) != 0;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const cf w = make_float2(__uint_as_float(odd ? bq[r].z : bq[r].x), __uint_as_float(odd ? bq[r].w : bq[r].y));
            v[r] = cmul(v[r], w);
        }
    } else {
#pragma unroll
        for (int r = 0; r < RB; ++r) v[r] = cmul(v[r], bsp[j + RA * r]);
    }
    if (j == 0) v[0] = cadd(v[0], a0);
    // inverse FFT_312, radices reversed: stage 1 radix 13 on the registers as they are
    mix::dft_any<RB, true>(v);
    __syncthreads();                                             // stage 2's reads of the image are finished
#pragma unroll
    for (int r = 0; r < RB; ++r) lds[(j * RB + r) * W + c] = v[r];
    __syncthreads();
    // stage 2: radix 24 on z[j + 13 r], twiddle W_312^(-j r), butterflies j < 13; outputs m = j + 13 r go to k2 = g^-m
    if (j < RB) {
        if constexpr (!TWO) {
#pragma unroll
            for (int r = 0; r < RA; ++r) v[r] = lds[(j + RB * r) * W + c];
        }
        if constexpr (TWO) {
            // the 24-point butterfly as 8 x 3 by hand (three 8-point butterflies, then eight 3-point ones), its inputs read, twiddled and
            // consumed eight at a time and its outputs stored three at a time: the 24 inputs, 24 partial results and 24 store addresses are never alive
            // together (as one dft_any<24> after 24 LDS reads the compiler held all of them: 80 VGPRs + 36 spilled)
            cf y[RA];
#pragma unroll
            for (int n2 = 0; n2 < 3; ++n2) {
                cf t[8];
#pragma unroll
                for (int q1 = 0; q1 < 8; ++q1) {
                    const int r = 3 * q1 + n2;
                    t[q1] = lds[(j + RB * r) * W + c];
                    if (r > 0) {          // times exp(+2 pi i j r / 312), every power from its own sine / cosine: no table of powers alive
                        unsigned m = j * (unsigned)r;             // < 13 * 23
                        if (m >= (unsigned)L) m -= L;
                        t[q1] = cmul(t[q1], cis_frac((float)m * (1.0f / (float)L)));
                    }
                }
                mix::dft_any<8, true>(t);
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1) y[n2 * 8 + k1] = mix::mul_root<RA, true>(t[k1], n2 * k1);
            }
            const bool st = (int)col < a.u_cols;
            const unsigned b = n1 * a.off0u + colb;
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) {
                cf t[3];
#pragma unroll
                for (int n2 = 0; n2 < 3; ++n2) t[n2] = y[n2 * 8 + k1];
                mix::dft_any<3, true>(t);
                if (st) {
#pragma unroll
                    for (int k2 = 0; k2 < 3; ++k2) stnt(dst + (b + __umul24((unsigned)tout[j + RB * (k1 + 8 * k2)], pu)), t[k2], true);
                }
            }
            if (st && j == 0) *reinterpret_cast<cf*>(dst + b) = y0;
            return;
        }
        mix::apply_powers<RA>(v, cis_frac((float)j * (1.0f / (float)L)));
        mix::dft_any<RA, true>(v);
        if ((int)col < a.u_cols) {
            const unsigned b = n1 * a.off0u + colb;
#pragma unroll
            for (int r = 0; r < RA; ++r) stnt(dst + (b + (unsigned)tout[j + RB * r]), v[r], TWO || a.nt);
            if (j == 0) *reinterpret_cast<cf*>(dst + b) = y0;
        }
    }
}
template <int W>
__global__ __launch_bounds__(pfa::RA * W) void pfa_rader313_kernel(PfaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    rader313_body<W, false>(a, reinterpret_cast<cf*>(smem_raw));
}
template <int W>
__global__ __launch_bounds__(pfa::RA * W) __attribute__((amdgpu_waves_per_eu(6, 6))) void pfa_rader313_two_kernel(PfaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    rader313_body<W, true>(a, reinterpret_cast<cf*>(smem_raw));
}

// 23-point DFTs over n1 of rows n1*313 + k2 of a.u, in registers; output row k = (c2 k2 + c1 k1) mod 7199
// EPI: 0 none, 1 times Phi_1(row, col) (forward), 2 times a.scale (inverse)
template <bool INV, int EPI>
__global__ __launch_bounds__(256) void pfa_dft23_kernel(PfaArgs a) {
    using namespace pfa;
    const unsigned col = blockIdx.x * 64 + (threadIdx.x & 63);
    const unsigned k2 = blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool live = (int)col < a.out_cols && k2 < (unsigned)P;
    float vmax = 0.f;
    float ati_thr = 0.f;
    double ati_re = 0.0, ati_im = 0.0;
    if constexpr (EPI == 3) {                      // mask threshold: max over the shards the first channel's focus left
        __shared__ float s_m[4];
        float m = a.ati.thr[32 * threadIdx.x];     // 256 threads, MAX_SHARDS = 256
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
        __syncthreads();
        ati_thr = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])) * a.ati.frac;
    }
    if (live) {
        const unsigned colb = col * (unsigned)sizeof(cf);
        const char* __restrict__ src = reinterpret_cast<const char*>(a.u);
        char* __restrict__ dst = reinterpret_cast<char*>(a.out);
        cf v[N1];
        unsigned o = k2 * a.pitch_u + colb;                          // row n1*313 + k2: byte offset advances by 313 rows
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) { v[n1] = ldnt(src + o, a.nt); o += a.off0u; }
        unsigned row = (unsigned)(((unsigned long long)a.c2k * k2) % N);
        cf s1v[EPI == 3 ? N1 : 1];                   // the first channel's samples, requested before the transform (az_tile_kernel does the same)
        if constexpr (EPI == 3) {
            unsigned rw = row;
#pragma unroll
            for (int k1 = 0; k1 < N1; ++k1) {
                s1v[k1] = a.ati.s1[(size_t)rw * (a.pitch_out / (unsigned)sizeof(cf)) + col];
                rw += a.c1k;
                if (rw >= (unsigned)N) rw -= N;
            }
        }
        mix::dft_any<N1, INV>(v);
        unsigned oo = row * a.pitch_out + colb;
        const unsigned step = (unsigned)a.c1k * a.pitch_out, wrap = (unsigned)N * a.pitch_out;
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) {
            cf x = v[k1];
            if constexpr (EPI == 1) x = cmul(x, phi1((int)col, a.c1[row], a.dt, a.t_start));
            else if constexpr (EPI == 2) {
                x.x *= a.scale; x.y *= a.scale;
                if (a.max_out) vmax = fmaxf(vmax, hypotf(x.x, x.y));      // max |image| for the ATI mask, as in az_tile_kernel
            } else if constexpr (EPI == 3) {                               // the ATI / DPCA products instead of (or beside) the image
                x.x *= a.scale; x.y *= a.scale;
                const size_t o = (size_t)row * (a.pitch_out / (unsigned)sizeof(cf)) + col;
                Pix px;
                ati_pixel<false>(s1v[k1], x, a.ati.cc, a.ati.cs, px);
                a.ati.phase[o] = px.m1 > ati_thr ? px.phase : 0.f;
                a.ati.m1[o] = px.m1;
                a.ati.dm[o] = px.dm;
                ati_re += px.sre; ati_im += px.sim;
            }
            if (EPI != 3 || a.ati.keep_image) stnt(dst + oo, x, a.nt);
            row += a.c1k; oo += step;
            if (row >= (unsigned)N) { row -= N; oo -= wrap; }
        }
    }
    if constexpr (EPI == 3) {   // every lane arrives here (dead lanes carry 0): one partial per wave, finished in fixed order
        for (int off = 32; off > 0; off >>= 1) { ati_re += __shfl_xor(ati_re, off, 64); ati_im += __shfl_xor(ati_im, off, 64); }
        if ((threadIdx.x & 63) == 0)
            a.ati.part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = make_double2(ati_re, ati_im);
    }
    if constexpr (EPI == 2) {
        if (a.max_out) {        // every lane arrives here (dead lanes carry 0): one sharded atomic per wave
            for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
            if ((threadIdx.x & 63) == 0)
                atomicMax(a.max_out + 32u * ((blockIdx.x * 7u + blockIdx.y * 13u + (threadIdx.x >> 6)) & (MAX_SHARDS - 1u)), __float_as_uint(vmax));
        }
    }
}

bool az_pfa_supported(int n_az) { return n_az == pfa::N; }
int az_pfa_ati_parts(int dst_cols) { return ((dst_cols + 63) / 64) * ((pfa::P + 3) / 4) * 4; }

// ---- host tables ------------------------------------------------------------------------------------------------------
static int pow_mod(long long b, long long e, long long m) {
    long long r = 1;
    b %= m;
    while (e > 0) { if (e & 1) r = r * b % m; b = b * b % m; e >>= 1; }
    return (int)r;
}
static int inv_mod(int a, int m) { return pow_mod(a, m - 2, m); }     // m prime

struct AzPfa {
    unsigned *offin = nullptr, *offu = nullptr;   // [312] byte offsets: 23 g^q rows of the source, g^-m rows of the intermediate
    size_t in_ld = 0, u_ld = 0, out_ld = 0;       // leading dimensions (elements) the tables were built for
    cf *bspec_f = nullptr, *bspec_i = nullptr;
    int c1k = 0, c2k = 0;
};
void az_pfa_destroy(AzPfa* z) {
    if (!z) return;
    hipFree(z->offin); hipFree(z->offu); hipFree(z->bspec_f); hipFree(z->bspec_i);
    delete z;
}
AzPfa* az_pfa_create(size_t in_ld, size_t u_ld, size_t out_ld, hipError_t* err) {
    using namespace pfa;
    AzPfa* z = new AzPfa();
    int g = 2;                                                   // smallest primitive root mod 313: order exactly 312 = 2^3 * 3 * 13
    for (;; ++g)
        if (pow_mod(g, L / 2, P) != 1 && pow_mod(g, L / 3, P) != 1 && pow_mod(g, L / 13, P) != 1) break;
    std::vector<int> gpow(L), ginv(L);
    for (int q = 0; q < L; ++q) gpow[q] = pow_mod(g, q, P);
    for (int m = 0; m < L; ++m) ginv[m] = gpow[(L - m) % L];
    // spectrum of w'[q] = exp(s 2 pi i g^-q / 313), forward FFT_312, divided by 312, for both signs s
    std::vector<cf> bf(L), bi(L);
    for (int dir = 0; dir < 2; ++dir) {
        const double s = dir == 0 ? -1.0 : 1.0;
        for (int k = 0; k < L; ++k) {
            std::complex<double> acc(0, 0);
            for (int q = 0; q < L; ++q) {
                const double aw = s * 2.0 * M_PI * (double)ginv[q] / (double)P;
                const double af = -2.0 * M_PI * (double)((long long)q * k % L) / (double)L;
                acc += std::polar(1.0, aw + af);
            }
            acc /= (double)L;
            (dir == 0 ? bf : bi)[k] = make_float2((float)acc.real(), (float)acc.imag());
        }
    }
    z->c1k = P * inv_mod(P % N1, N1);
    z->c2k = N1 * inv_mod(N1 % P, P);
    hipError_t e = hipSuccess;
    auto up = [&](const void* h, size_t bytes, void** d) {
        if (e != hipSuccess) return;
        e = hipMalloc(d, bytes);
        if (e == hipSuccess) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
    };
    // byte offsets fit 32 bits: the images are [7199 x ld] complex64 with ld <= 32768 + 31
    if (2ull * N * (in_ld > u_ld ? (in_ld > out_ld ? in_ld : out_ld) : (u_ld > out_ld ? u_ld : out_ld)) * sizeof(cf) >= (1ull << 32)) {
        if (err) *err = hipErrorInvalidValue;
        delete z;
        return nullptr;
    }
    z->in_ld = in_ld; z->u_ld = u_ld; z->out_ld = out_ld;
    std::vector<unsigned> offin(L), offu(L);                   // n1 enters in the kernel as n1 * (313 rows), wrapped at 7199 rows
    for (int q = 0; q < L; ++q) {
        offin[q] = (unsigned)((size_t)(N1 * gpow[q]) * in_ld * sizeof(cf));
        offu[q] = (unsigned)((size_t)ginv[q] * u_ld * sizeof(cf));
    }
    up(offin.data(), offin.size() * sizeof(unsigned), (void**)&z->offin);
    up(offu.data(), offu.size() * sizeof(unsigned), (void**)&z->offu);
    up(bf.data(), L * sizeof(cf), (void**)&z->bspec_f);
    up(bi.data(), L * sizeof(cf), (void**)&z->bspec_i);
    if (err) *err = e;
    if (e != hipSuccess) { az_pfa_destroy(z); return nullptr; }
    return z;
}

// src: dense [7199 x src_cols] (leading dimension src_ld); u: work array [7199 x u_ld], u_ld >= dst_cols;
// dst: dense [7199 x dst_cols]; epi: 0 none, 1 Phi_1 (a.c1 / dt / t_start must be set), 2 scale
hipError_t az_pfa_run(const AzPfa* z, bool inv, const cf* src, size_t src_ld, int src_cols, cf* u, size_t u_ld, cf* dst,
                      size_t dst_ld, int dst_cols, int epi, const double2* c1, double dt, double t_start, float scale,
                      hipStream_t st, unsigned* max_out, const AtiFuse* ati) {
    using namespace pfa;
    constexpr int W = 32;
    PfaArgs a{};
    a.in = src; a.in_ld = src_ld; a.in_cols = src_cols;
    a.u = u; a.u_ld = u_ld; a.u_cols = (int)(u_ld < (size_t)dst_cols ? u_ld : (size_t)dst_cols);
    a.out = dst; a.out_ld = dst_ld; a.out_cols = dst_cols;
    a.offin = z->offin; a.offu = z->offu; a.bspec = inv ? z->bspec_i : z->bspec_f;
    a.c1 = c1; a.dt = dt; a.t_start = t_start; a.scale = scale; a.c1k = z->c1k; a.c2k = z->c2k;
    a.max_out = (inv && epi == 2) ? max_out : nullptr;
    const bool fuse_ati = inv && epi == 2 && ati && ati->s1;
    if (fuse_ati) a.ati = *ati;
    if (src_ld != z->in_ld || u_ld != z->u_ld || dst_ld != z->out_ld) return hipErrorInvalidValue;   // tables are per pitch
    a.off0in = (unsigned)(P * src_ld * sizeof(cf)); a.off0u = (unsigned)(P * u_ld * sizeof(cf));
    a.pitch_u = (unsigned)(u_ld * sizeof(cf)); a.pitch_out = (unsigned)(dst_ld * sizeof(cf));
    // every image here is 0.76 GB and is next read a whole launch later: nontemporal accesses (2.19 -> 2.13 ms per native frame);
    // SARX_PFA_NT=0 for A/B
    { static const int nt = [] { const char* e = getenv("SARX_PFA_NT"); return e ? atoi(e) : 1; }(); a.nt = nt != 0; }
    // two workgroups per CU: 0.45 -> 0.40 ms per launch at 7199 x 13200, the native frame 2.00 -> 1.87 ms (profiles/r03_s_rader_two_ab.log;
    // SARX_RADER_TWO=0 for A/B)
    static const int two = [] { const char* e = getenv("SARX_RADER_TWO"); return e ? atoi(e) : 1; }();
    hipError_t e;
    if (two && a.nt) {      // the two-workgroup form has the nontemporal accesses compiled in
        const size_t lds = (size_t)L * W * sizeof(cf) + 2 * L * sizeof(unsigned short);          // image + the two row-number tables: two fit a CU
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(pfa_rader313_two_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(pfa_rader313_two_kernel<W>, dim3((dst_cols + W - 1) / W, N1), dim3(RA * W), lds, st, a);
    } else {
        const size_t lds = (size_t)L * W * sizeof(cf) + 2 * L * sizeof(unsigned) + L * sizeof(cf);     // image + the two offset tables + the spectrum of w'
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(pfa_rader313_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(pfa_rader313_kernel<W>, dim3((dst_cols + W - 1) / W, N1), dim3(RA * W), lds, st, a);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    dim3 g2((dst_cols + 63) / 64, (P + 3) / 4);
    if (!inv) {
        if (epi == 1) hipLaunchKernelGGL((pfa_dft23_kernel<false, 1>), g2, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((pfa_dft23_kernel<false, 0>), g2, dim3(256), 0, st, a);
    } else {
        if (fuse_ati) hipLaunchKernelGGL((pfa_dft23_kernel<true, 3>), g2, dim3(256), 0, st, a);
        else if (epi == 2) hipLaunchKernelGGL((pfa_dft23_kernel<true, 2>), g2, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((pfa_dft23_kernel<true, 0>), g2, dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

}  // namespace sarx
