// Fused range pass for n_rg = 16384:  FFT . Phi_2 . IFFT . Phi_3  in one launch
// (sar_ati_dcpa_sim_csa.py:278-382), organised so that only two of its twelve
// LDS exchanges need workgroup barriers.
//
// The spectrum never leaves registers, so its ordering is free.  The forward
// transform is decimation-in-frequency over 16 x 1024, the inverse mirrors it:
//
//   load   x[n1*1024 + n2]            thread t owns n2 = 2t, 2t+1 (16 B per lane), n1 = 0..15
//   radix-16 over n1, twiddle W_N^(n2*q)                      -> y_q[n2], q = 0..15
//   CROSS exchange (2 barriers): wave w takes the lines q = 2w, 2w+1
//   per wave: two 1024-point transforms over n2, 64 lanes x 16 points each, exchanged through
//             wave-private LDS regions (no barriers: LDS ops of one wave execute in order) and
//             advanced stage by stage so one's LDS traffic flies while the other's butterflies issue
//                                                              -> X[q + 16*k2]
//   Phi_2 at bin k = q + 16*k2 (natural-order fftfreq value, any storage order)
//   per wave: two inverse 1024-point transforms               -> z_q[n2]
//   CROSS exchange back (1 barrier), twiddle conj(W_N^(n2*q)), inverse radix-16 over q
//   Phi_3 / N, store x[n1*1024 + n2]  (16 B per lane)
//
// Residency: one persistent 512-thread workgroup per CU (complex cross image [16][1088] = 136 KiB),
// two waves per SIMD, 141 VGPRs.  The register file, not LDS, rules out two lines per CU: the same
// body capped at 128 VGPRs (real/imaginary parts exchanged separately through a 68 KiB image, two
// workgroups per CU) spills 308 B/lane and runs 2.0 ms against 1.41 ms (measured).
#include <cstdlib>
#include "csa_kernels.h"
#include "fft_core.hpp"
#include "phase.hpp"

namespace sarx {

namespace wl {
constexpr int N = 16384, M = 1024, THREADS = 512;
constexpr int ROWSTR = LdsSize<M, 1>::value;                     // 1088: the padded 1024-point image
constexpr size_t LDS_BYTES = (size_t)16 * ROWSTR * sizeof(cf);   // 139264

// thread-major (v[b*16+q] = y_q[2t+b]) -> wave-major (v[s*16+r] = y_{2w+s}[l + 64 r])
__device__ __forceinline__ void cross_fwd(cf* v, int t, int w, int l, cf* lds, bool lead_barrier) {
    if (lead_barrier) __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q)
        *reinterpret_cast<float4*>(&lds[q * ROWSTR + 2 * t]) = make_float4(v[q].x, v[q].y, v[16 + q].x, v[16 + q].y);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[s * 16 + r] = lds[(2 * w + s) * ROWSTR + l + 64 * r];
}
// wave-major -> thread-major.  Each wave writes only its own two lines, which nobody else has read
// since the forward cross exchange, so no leading barrier.
__device__ __forceinline__ void cross_inv(cf* v, int t, int w, int l, cf* lds) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[(2 * w + s) * ROWSTR + l + 64 * r] = v[s * 16 + r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float4 p = *reinterpret_cast<const float4*>(&lds[q * ROWSTR + 2 * t]);
        v[q] = make_float2(p.x, p.y);
        v[16 + q] = make_float2(p.z, p.w);
    }
}
}  // namespace wl

__global__ __launch_bounds__(wl::THREADS, 2) void range_fused_wl_kernel(RangeArgs a) {
    using namespace wl;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    const cf* __restrict__ tw = a.tw;                 // exp(-2 pi i m / 16384)
    const cf* __restrict__ tw_m = a.tw - N + M;       // the 1024 table sits at offset 1024 of the same array

    for (int row = blockIdx.x; row < a.n_az; row += gridDim.x) {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));                   // keep addresses per-line (no hoisting out of the loop + spilling)
        const int w = t >> 6, l = t & 63;
        const cf* __restrict__ src = a.in + (size_t)row * N;
        cf* __restrict__ dst = a.out + (size_t)row * N;
        cf* priv0 = lds + (2 * w) * ROWSTR;           // wave-private: lines 2w and 2w+1 of the cross image

        cf v[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 q4 = *reinterpret_cast<const float4*>(src + 2 * t + r * M);
            v[r] = make_float2(q4.x, q4.y);
            v[16 + r] = make_float2(q4.z, q4.w);
        }
        // forward radix-16 over n1, then twiddle W_N^(n2 q)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            dft16<false>(v + 16 * b);
            apply_twiddle_powers<16>(v + 16 * b, stage_twiddle<N, N, false>(2 * t + b, tw));
        }
        cross_fwd(v, t, w, l, lds, row != (int)blockIdx.x);
        stockham_run2_wave<M, false, false>(v, v + 16, l, priv0, priv0 + ROWSTR, tw_m);
        // Phi_2: register (s; b, r) of the radix-4 last stage holds k2 = l + 64 m, m = b + 4 r, i.e. bin
        // k = (2w+s) + 16 l + 1024 m; m >= 8 are the negative frequencies
        {
            const double2 c2 = a.c2[row];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ks = 2 * w + 16 * l + half * (8 * 1024 - N);     // signed fftfreq index at m = 8*half
                FixPhase q0 = phi2_seed(ks, 1024, c2, a.df), q1 = phi2_seed(ks + 1, 1024, c2, a.df);
#pragma unroll
                for (int mm = 0; mm < 8; ++mm) {
                    const int m = half * 8 + mm;
                    const int reg = (m % 4) * 4 + m / 4;
                    v[reg] = cmul(v[reg], q0.next());
                    v[16 + reg] = cmul(v[16 + reg], q1.next());
                }
            }
        }
        // two wave-private inverse transforms; the reversed plan starts on the radix-4 layout just produced
        stockham_run2_wave<M, true, true>(v, v + 16, l, priv0, priv0 + ROWSTR, tw_m);
        exchange_sync<true>();
        cross_inv(v, t, w, l, lds);
        // conj twiddle, inverse radix-16 over q
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            apply_twiddle_powers<16>(v + 16 * b, stage_twiddle<N, N, true>(2 * t + b, tw));
            dft16<true>(v + 16 * b);
        }
        // Phi_3 / N and store: v[b*16 + n1] = N * x[n1*1024 + 2t + b]
        {
            const double2 c3 = a.c3[row];
            const float sc = a.inv_n;
            FixPhase q0 = phi3_seed(2 * t, M, c3, a.dt, a.t_start, a.t0);
            FixPhase q1 = phi3_seed(2 * t + 1, M, c3, a.dt, a.t_start, a.t0);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                cf p0 = q0.next(), p1 = q1.next();
                p0.x *= sc; p0.y *= sc; p1.x *= sc; p1.y *= sc;
                const cf y0 = cmul(v[n1], p0), y1 = cmul(v[16 + n1], p1);
                *reinterpret_cast<float4*>(dst + 2 * t + n1 * M) = make_float4(y0.x, y0.y, y1.x, y1.y);
            }
        }
    }
}

bool range_fused_wl_supported(int n_rg) { return n_rg == wl::N; }

hipError_t launch_range_fused_wl(const RangeArgs& a, hipStream_t st) {
    static bool attr_set = false;
    static int cus = 0;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(range_fused_wl_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl::LDS_BYTES);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
        attr_set = true;
    }
    int grid = cus;                                    // one resident workgroup per CU, persistent over lines
    if (grid > a.n_az) grid = a.n_az;
    hipLaunchKernelGGL(range_fused_wl_kernel, dim3(grid), dim3(wl::THREADS), wl::LDS_BYTES, st, a);
    return hipGetLastError();
}

}  // namespace sarx
