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
//   per half wave: one 1024-point transform over n2 as 32 x 32 (32 lanes x 32 points), its single
//             exchange through a private LDS region (no barriers: LDS ops of one wave execute in order)
//                                                              -> X[q + 16*k2]
//   Phi_2 at bin k = q + 16*k2 (natural-order fftfreq value, any storage order)
//   per wave: two inverse 1024-point transforms               -> z_q[n2]
//   CROSS exchange back (1 barrier), twiddle conj(W_N^(n2*q)), inverse radix-16 over q
//   Phi_3 / N, store x[n1*1024 + n2]  (16 B per lane)
//
// Residency: one persistent 512-thread workgroup per CU (complex cross image [16][1088] = 136 KiB),
// two waves per SIMD, 233 VGPRs (122 of them the thread-constant twiddles hoisted out of the line loop).
// The register file, not LDS, rules out two lines per CU: the same body capped at 128 VGPRs spills
// (real/imaginary split image, round 1: 308 B/lane, 2.0 vs 1.4 ms; the sixteen-wave structure of
// range_wp.hip, round 3: 28-156 B/lane, 1.18-1.25 vs 1.09-1.12 ms).  Bound by instruction
// issue at two waves per SIMD (3 760 vector instructions per 32 samples), not by its exchanges or by HBM.
#include <cstdlib>
#include "csa_kernels.h"
#include "fft_core.hpp"
#include "phase.hpp"

namespace sarx {

namespace wl {
constexpr int N = 16384, M = 1024, THREADS = 512;
constexpr int ROWSTR = LdsSize<M, 1>::value;                     // 1088: the padded 1024-point image
constexpr size_t LDS_BYTES = (size_t)16 * ROWSTR * sizeof(cf);   // 139264

// Wave-private 1024-point transforms as 32 x 32: half-wave h = l >> 5 owns line 2w+h, lane i = l & 31
// holds its points i + 32 r, r = 0..31.  One LDS exchange per transform (radix-32 twice); the 16.16.4
// form with two exchanges measured 4 % slower (1.345 vs 1.29 ms).
__device__ __forceinline__ int pad32(int idx) { return idx + (idx >> 5); }   // stride-32 writes hit distinct banks

__device__ __forceinline__ void cross_fwd32(cf* v, int t, int w, int l, cf* lds, bool lead_barrier) {
    if (lead_barrier) __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q)
        *reinterpret_cast<float4*>(&lds[q * ROWSTR + 2 * t]) = make_float4(v[q].x, v[q].y, v[16 + q].x, v[16 + q].y);
    __syncthreads();
    const cf* line = lds + (2 * w + (l >> 5)) * ROWSTR + (l & 31);
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = line[32 * r];
}
__device__ __forceinline__ void cross_inv32(cf* v, int t, int w, int l, cf* lds) {
    cf* line = lds + (2 * w + (l >> 5)) * ROWSTR + (l & 31);
#pragma unroll
    for (int r = 0; r < 32; ++r) line[32 * r] = v[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float4 p = *reinterpret_cast<const float4*>(&lds[q * ROWSTR + 2 * t]);
        v[q] = make_float2(p.x, p.y);
        v[16 + q] = make_float2(p.z, p.w);
    }
}
// one 1024-point transform of the half wave, in place in v: inputs v[r] = x[i + 32 r], outputs v[r] = X[i + 32 r]
// wp[r] = W_1024^(i r) depends on the lane only: computed once per kernel, kept in registers (62 VGPRs)
template <bool INV> __device__ __forceinline__ void sub1024_r32(cf* v, int i, cf* img, const cf* wp) {
    dft32<INV>(v);
    exchange_sync<true>();
#pragma unroll
    for (int r = 0; r < 32; ++r) img[pad32(32 * i + r)] = v[r];
    exchange_sync<true>();
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = img[pad32(i + 32 * r)];
#pragma unroll
    for (int r = 1; r < 32; ++r)
        v[r] = INV ? make_float2(fmaf(v[r].x, wp[r].x, v[r].y * wp[r].y), fmaf(v[r].y, wp[r].x, -v[r].x * wp[r].y))   // * conj
                   : cmul(v[r], wp[r]);
    dft32<INV>(v);
}
}  // namespace wl

// PF = 1 (round 5): every thread touches two 128-byte lines of the NEXT line of its workgroup once this line's own loads have landed
// (one dword each into a register nobody reads: the data is wanted in L2 / the Infinity Cache, where the loads at the top of the next
// iteration then find it).  No register prefetch fits this kernel; this costs ten VGPRs.  Alone on the GPU and in place the launch runs
// 1.040 against 1.089 ms (0.516 against 0.493 of the HBM peak, profiles/r05_f_wl_touch_prefetch.log); with two frames in flight the
// extra requests compete with the other lane's azimuth tiles and the frame gets 1 % SLOWER, so the launcher uses it only when the
// launch has the chip to itself (no CU share set).  Touching right behind the line's own loads, and nontemporal stores with or without
// the touch, measured worse or equal (same log).
template <int PF>
__device__ __forceinline__ void range_fused_wl_body(const RangeArgs& a) {
    using namespace wl;
    constexpr bool TOUCH = (PF == 1);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    if (a.stamp && threadIdx.x == 0) atomicMin(a.stamp, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    const cf* __restrict__ tw = a.tw;                 // exp(-2 pi i m / 16384)
    const cf* __restrict__ tw_m = a.tw - N + M;       // the 1024 table sits at offset 1024 of the same array

    cf wp[32];
    {
        const cf w1 = stage_twiddle<M, M, false>((int)(threadIdx.x & 31), tw_m);
        wp[0] = make_float2(1.f, 0.f);
        wp[1] = w1;
#pragma unroll
        for (int r = 2; r < 32; ++r) wp[r] = cmul(wp[r / 2], wp[r - r / 2]);
    }
    // cross twiddles W_N^((2t+b) q), q = 1..15: thread constants as well (60 VGPRs)
    cf cw[2][16];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const cf w1 = stage_twiddle<N, N, false>(2 * (int)threadIdx.x + b, tw);
        cw[b][0] = make_float2(1.f, 0.f);
        cw[b][1] = w1;
#pragma unroll
        for (int q = 2; q < 16; ++q) cw[b][q] = cmul(cw[b][q / 2], cw[b][q - q / 2]);
    }
    unsigned pf0 = 0, pf1 = 0;                        // destinations of the touch loads: reserved until the loads have landed
    for (int line = blockIdx.x; line < a.n_az; line += gridDim.x) {
        const int row = range_row(a, line);
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));                   // keep addresses per-line (no hoisting out of the loop + spilling)
        const int w = t >> 6, l = t & 63;
        cf* __restrict__ dst = a.out + (size_t)row * N;

        cf v[32];
        const cf* __restrict__ src = a.in + (size_t)row * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 q4 = ld16<false>(src + 2 * t + r * M);
            v[r] = make_float2(q4.x, q4.y);
            v[16 + r] = make_float2(q4.z, q4.w);
        }
        auto touch_next = [&] {
            const int nl = line + (int)gridDim.x;
            if (nl < a.n_az) {                        // 1024 lines of 128 bytes per 16384-sample row: two per thread
                const cf* np = a.in + (size_t)range_row(a, nl) * N + 16 * t;
                asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off"
                             : "=&v"(pf0), "=&v"(pf1) : "v"(np), "v"(np + 16 * THREADS) : "memory");
            }
        };
        // the row's phase constants through the scalar cache, behind the line's own loads (their latency covers it): as vector
        // loads next to their use their latency was exposed twice per line
        const double2 c2 = sload_double2(a.c2 + row), c3 = sload_double2(a.c3 + row);
        // forward radix-16 over n1, then twiddle W_N^(n2 q)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            dft16<false>(v + 16 * b);
#pragma unroll
            for (int q = 1; q < 16; ++q) v[16 * b + q] = cmul(v[16 * b + q], cw[b][q]);
        }
        cf* img = lds + (2 * w + (l >> 5)) * ROWSTR;  // the half wave's private image
        const int li = l & 31;
        if constexpr (TOUCH) {
            // vmcnt retires in order: this line's loads are younger than the previous iteration's touch loads, so a value computed from
            // v[] (the butterflies above waited for it) proves pf0 / pf1 have been written: their registers may be reused from here on
            asm volatile("" :: "v"(pf0), "v"(pf1), "v"(v[0].x), "v"(v[31].y));
        }
        cross_fwd32(v, t, w, l, lds, line != (int)blockIdx.x);
        if constexpr (TOUCH) touch_next();
        sub1024_r32<false>(v, li, img, wp);
        // Phi_2: v[r] is bin k = (2w+h) + 16 i + 512 r; r >= 16 are the negative frequencies
        {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                FixPhase q0 = phi2_seed(2 * w + (l >> 5) + 16 * li + half * (8192 - N), 512, c2, a.df);
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) v[half * 16 + rr] = cmul(v[half * 16 + rr], q0.next());
            }
        }
        sub1024_r32<true>(v, li, img, wp);
        exchange_sync<true>();
        cross_inv32(v, t, w, l, lds);
        // conj twiddle, inverse radix-16 over q
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int q = 1; q < 16; ++q) {       // * conj(cw)
                const cf x = v[16 * b + q], c = cw[b][q];
                v[16 * b + q] = make_float2(fmaf(x.x, c.x, x.y * c.y), fmaf(x.y, c.x, -x.x * c.y));
            }
            dft16<true>(v + 16 * b);
        }
        // Phi_3 / N and store: v[b*16 + n1] = N * x[n1*1024 + 2t + b]
        {
            const float sc = a.inv_n;
            FixPhase q0 = phi3_seed(2 * t, M, c3, a.dt, a.t_start, a.t0);
            FixPhase q1 = phi3_seed(2 * t + 1, M, c3, a.dt, a.t_start, a.t0);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                cf p0 = q0.next(), p1 = q1.next();
                p0.x *= sc; p0.y *= sc; p1.x *= sc; p1.y *= sc;
                const cf y0 = cmul(v[n1], p0), y1 = cmul(v[16 + n1], p1);
                st16<false>(dst + 2 * t + n1 * M, make_float4(y0.x, y0.y, y1.x, y1.y));
            }
        }
    }
    if constexpr (TOUCH) {
        __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the last touch loads have landed before their registers die
        asm volatile("" :: "v"(pf0), "v"(pf1));
    }
    if (a.stamp && threadIdx.x == 0) {                // after this wave's last stores have been acknowledged
        __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0)
        atomicMax(a.stamp + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}
__global__ __launch_bounds__(wl::THREADS, 2) void range_fused_wl_kernel(RangeArgs a) { range_fused_wl_body<0>(a); }
__global__ __launch_bounds__(wl::THREADS, 2) void range_fused_wl_touch_kernel(RangeArgs a) { range_fused_wl_body<1>(a); }

bool range_fused_wl_supported(int n_rg) { return n_rg == wl::N; }

// cus: compute units of the device the stream belongs to (from the ctx).  The 136 KiB dynamic-LDS opt-in is a
// per-device attribute of the function, so it is set on every launch like the other launchers do (a process may own
// contexts on several GPUs; a once-per-process flag would leave the second device without it).
hipError_t launch_range_fused_wl(const RangeArgs& a, int cus, hipStream_t st, bool alone) {
    static const int pf = [] { const char* e = getenv("SARX_WL_PREFETCH"); return e ? atoi(e) : -1; }();      // 0 / 1 force it off / on (A/B)
    const bool touch = pf < 0 ? (alone && a.in == a.out) : pf == 1;
    auto k = touch ? range_fused_wl_touch_kernel : range_fused_wl_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl::LDS_BYTES);
    if (e != hipSuccess) return e;
    const int grid = persistent_grid(1, cus, a.n_az);   // one resident workgroup per CU, persistent over lines
    hipLaunchKernelGGL(k, dim3(grid), dim3(wl::THREADS), wl::LDS_BYTES, st, a);
    return hipGetLastError();
}

}  // namespace sarx
