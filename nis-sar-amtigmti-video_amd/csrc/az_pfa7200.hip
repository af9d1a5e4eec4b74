// Pulse-axis transforms of the Range-Doppler focuser at the satellite scripts' 7200 pulses (sar_satellite_sim.py:83-85,
// 396-399, 438) without chirp-z: 7200 = 32 * 225, coprime, so the Good-Thomas (prime-factor) map needs no twiddles
// between the factors:
//       n = (225 n1 + 32 n2) mod 7200,      k = (c1 k1 + c2 k2) mod 7200,  c1 = 225 (225^-1 mod 32) = 225,  c2 = 32 (32^-1 mod 225) = 6976
//       X[k] = sum_n1 W_32^(n1 k1) [ sum_n2 x[n] W_225^(n2 k2) ]
//   launch 1 (pfa_dft225_kernel): the 225-point transforms over n2 as 15 x 15 - two Stockham stages of in-register DFT-15
//       (fft_mixed.hpp: 3 x 5) around one exchange through a [225 x 32] LDS image - one (n1, 32-column tile) per
//       workgroup of 480 threads, two workgroups per CU.  Rows (n + shift) mod 7200 of the dense source in (the fftshift of
//       :398 / :438 is part of the row address; the Hamming window of :396-397 is a per-row factor), rows n1*225 + k2 of
//       the intermediate out.
//   launch 2 (pfa_dft32_kernel): the 32-point transforms over n1 entirely in registers, one thread per (k2, column),
//       output rows (k - shift) mod 7200 (the second fftshift), complex or as the magnitude (:439) on the way out.
// Two HBM round trips of the unpadded [7200 x n_ranges] image per transform; the chirp-z route over 16384 rows took three
// launches over 2.3x the rows.
#include <cstdlib>

#include "csa_kernels.h"
#include "fft_mixed.hpp"

namespace sarx {
namespace pfa72 {
constexpr int N = 7200, N1 = 32, N2 = 225, R = 15, W = 32, THREADS = R * W;
constexpr int C1K = 225, C2K = 6976;
static_assert(N1 * N2 == N && R * R == N2, "factorisation");
static_assert((C1K % N1) == 1 && (C1K % N2) == 0 && (C2K % N2) == 1 && (C2K % N1) == 0, "Good-Thomas output map");

typedef float nt_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf ldg(const cf* p, bool nt) {
    if (nt) { const nt_v2f v = __builtin_nontemporal_load(reinterpret_cast<const nt_v2f*>(p)); return make_float2(v.x, v.y); }
    return *p;
}
__device__ __forceinline__ void stg(cf* p, cf x, bool nt) {
    if (nt) __builtin_nontemporal_store(nt_v2f{x.x, x.y}, reinterpret_cast<nt_v2f*>(p));
    else *p = x;
}
}  // namespace pfa72

struct Pfa72Args {
    const float2* in; size_t in_ld;      // dense source [7200 x cols]
    float2* u; size_t u_ld;              // intermediate, rows n1*225 + k2
    float2* out; float* out_mag; size_t out_ld;   // dense destination: complex, or magnitudes when out_mag is set
    const float* pre;                    // optional factor per SOURCE row (azimuth window)
    int cols, shift_in, shift_out;
    float scale;
    bool nt;
};

template <bool INV>
__global__ __launch_bounds__(pfa72::THREADS, 4) void pfa_dft225_kernel(Pfa72Args a) {
    using namespace pfa72;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);                   // [225][W]
    float* wrow = reinterpret_cast<float*>(lds + (size_t)N2 * W);   // [225] row factors of this n1 (the per-row loads cost 0.17 ms as 15 vector loads per thread)
    const int c = threadIdx.x % W, j = threadIdx.x / W;          // j in [0, 15)
    const int col = blockIdx.x * W + c, n1 = blockIdx.y;
    const bool live = col < a.cols;
    const int base = (N2 * n1 + a.shift_in) % N;
    if (a.pre) {
        if (threadIdx.x < N2) {
            int row = base + N1 * (int)threadIdx.x;
            if (row >= N) row -= N;
            wrow[threadIdx.x] = a.pre[row];
        }
        __syncthreads();
    }
    cf v[R];
    // stage 1: radix 15 on a[j + 15 r], straight from HBM; sequence element n = (225 n1 + 32 n2) mod 7200 is source row (n + shift_in) mod 7200
    {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int row = base + N1 * (j + R * r);                   // < 2 * 7200
            if (row >= N) row -= N;
            v[r] = live ? ldg(a.in + (size_t)row * a.in_ld + col, a.nt) : make_float2(0.f, 0.f);
        }
        if (a.pre) {
#pragma unroll
            for (int r = 0; r < R; ++r) { const float p = wrow[j + R * r]; v[r].x *= p; v[r].y *= p; }
        }
        mix::dft_any<R, INV>(v);
#pragma unroll
        for (int r = 0; r < R; ++r) lds[(R * j + r) * W + c] = v[r];
    }
    __syncthreads();
    // stage 2: radix 15 on y[j + 15 r], twiddle W_225^(-+ j r); outputs k2 = j + 15 s
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = lds[(j + R * r) * W + c];
    {
        const float x = (float)j * (1.0f / (float)N2);
        mix::apply_powers<R>(v, cis_frac(INV ? x : -x));
    }
    mix::dft_any<R, INV>(v);
    if (live) {
        cf* dst = a.u + (size_t)(n1 * N2 + j) * a.u_ld + col;
#pragma unroll
        for (int s = 0; s < R; ++s) stg(dst + (size_t)(R * s) * a.u_ld, v[s], a.nt);
    }
}

// MAG: |x| * scale as fp32 instead of the complex value
template <bool INV, bool MAG>
__global__ __launch_bounds__(256) void pfa_dft32_kernel(Pfa72Args a) {
    using namespace pfa72;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int k2 = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (col >= a.cols || k2 >= N2) return;
    cf v[N1];
    const cf* src = a.u + (size_t)k2 * a.u_ld + col;
#pragma unroll
    for (int n1 = 0; n1 < N1; ++n1) v[n1] = ldg(src + (size_t)(n1 * N2) * a.u_ld, a.nt);
    dft32<INV>(v);
    // bin k = (c2 k2 + c1 k1) mod 7200 goes to row (k - shift_out) mod 7200
    int row = (int)(((long long)C2K * k2 + (N - a.shift_out)) % N);
#pragma unroll
    for (int k1 = 0; k1 < N1; ++k1) {
        const size_t o = (size_t)row * a.out_ld + col;
        if constexpr (MAG) a.out_mag[o] = hypotf(v[k1].x, v[k1].y) * a.scale;
        else stg(a.out + o, make_float2(v[k1].x * a.scale, v[k1].y * a.scale), a.nt);
        row += C1K;
        if (row >= N) row -= N;
    }
}

bool az_pfa7200_supported(int n) { return n == pfa72::N; }

// src: dense [7200 x cols] (leading dimension src_ld); u: work array [7200 x u_ld], u_ld >= cols; dst / dst_mag: dense
// [7200 x cols] (leading dimension dst_ld), exactly one of them set.  Sequence element n is source row (n + shift_in) mod
// 7200 times pre[that row]; destination row r receives bin (r + shift_out) mod 7200, times scale.
hipError_t az_pfa7200_run(bool inv, const float2* src, size_t src_ld, int cols, float2* u, size_t u_ld, float2* dst, float* dst_mag,
                          size_t dst_ld, int shift_in, int shift_out, const float* pre, float scale, hipStream_t st) {
    using namespace pfa72;
    if (!src || !u || (!dst) == (!dst_mag) || cols <= 0 || u_ld < (size_t)cols) return hipErrorInvalidValue;
    Pfa72Args a{};
    a.in = src; a.in_ld = src_ld; a.u = u; a.u_ld = u_ld; a.out = dst; a.out_mag = dst_mag; a.out_ld = dst_ld;
    a.pre = pre; a.cols = cols; a.shift_in = ((shift_in % N) + N) % N; a.shift_out = ((shift_out % N) + N) % N; a.scale = scale;
    { static const int nt = [] { const char* e = getenv("SARX_PFA_NT"); return e ? atoi(e) : 1; }(); a.nt = nt != 0; }
    const size_t lds = (size_t)N2 * W * sizeof(cf) + N2 * sizeof(float);     // 58500: two workgroups per CU
    hipError_t e;
    const dim3 g1((cols + W - 1) / W, N1), g2((cols + 63) / 64, (N2 + 3) / 4);
    if (inv) hipLaunchKernelGGL((pfa_dft225_kernel<true>), g1, dim3(THREADS), lds, st, a);
    else hipLaunchKernelGGL((pfa_dft225_kernel<false>), g1, dim3(THREADS), lds, st, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (inv) {
        if (dst_mag) hipLaunchKernelGGL((pfa_dft32_kernel<true, true>), g2, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((pfa_dft32_kernel<true, false>), g2, dim3(256), 0, st, a);
    } else {
        if (dst_mag) hipLaunchKernelGGL((pfa_dft32_kernel<false, true>), g2, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((pfa_dft32_kernel<false, false>), g2, dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

}  // namespace sarx
