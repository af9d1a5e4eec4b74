// Range pass, second form: 32 points per thread, T = N/32 threads per line.
//
// Why: at n_rg = 16384 a full complex line image (128 KiB) leaves room for one
// workgroup per CU, so load, butterflies, LDS exchange and store of a line
// serialise (measured 17.5 us per line per CU).  Here the inter-stage exchange
// moves the real parts and then the imaginary parts through one float image
// (N*17/16 floats = 68 KiB at 16384), so two 512-thread workgroups are resident
// per CU and one line's HBM traffic overlaps the other's arithmetic.
//
// Thread t owns butterflies j = 2t, 2t+1 (+ multiples of 2T): neighbouring
// butterflies make every global access 16 B per lane (1 KiB per wave
// instruction) and every LDS access after the first exchange 8 B per lane.
#include <cstdlib>
#include "csa_kernels.h"
#include "fft_core.hpp"
#include "phase.hpp"

#ifndef SARX_NT_V2
#define SARX_NT_V2 3      // bit 0: nontemporal line loads, bit 1: nontemporal line stores (16384-sample lines only)
#endif

namespace sarx {

template <int N> struct V2 {
    using PL = Plan<N>;
    static constexpr int P = 32;
    static constexpr int T = N / P;
    static constexpr int ROWS = (T >= 256) ? 1 : 256 / T;
    static constexpr int THREADS = T * ROWS;
#ifndef SARX_V2_PAD_SHIFT
#define SARX_V2_PAD_SHIFT 5
#endif
    static constexpr int PAD_SHIFT = SARX_V2_PAD_SHIFT;            // two pad floats per 2^PAD_SHIFT floats
    static constexpr int LDS_PER_ROW = N + 2 * (N >> PAD_SHIFT);   // floats
    static constexpr size_t LDS_BYTES = (size_t)ROWS * LDS_PER_ROW * sizeof(float);
    // butterfly index of (thread t, slot b): pairs (b even, b odd) are adjacent
    __device__ static __forceinline__ int j(int t, int b) { return 2 * t + (b & 1) + (b >> 1) * (2 * T); }
    // float image index with two pad floats per block (keeps pairs 8-byte aligned, and keeps every
    // access of a thread at base + compile-time offset: an XOR swizzle needs one computed address per
    // access, which the compiler hoists out of the line loop and spills - measured 400 B/lane)
    __device__ static __forceinline__ int pad(int o) { return o + 2 * (o >> PAD_SHIFT); }
};

template <int C> __device__ __forceinline__ float& comp(cf& z) { if constexpr (C == 0) return z.x; else return z.y; }
template <int C> __device__ __forceinline__ float comp(const cf& z) { if constexpr (C == 0) return z.x; else return z.y; }

template <int N, int R, int NS, bool INV>
__device__ __forceinline__ void v2_compute(cf* v, int t, const cf* __restrict__ tw) {
    constexpr int B = 32 / R;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        if constexpr (NS > 1) {
            const int j = V2<N>::j(t, b);
            apply_twiddle_powers<R>(v + b * R, stage_twiddle<N, NS * R, INV>(j % NS, tw));
        }
        dft<R, INV>(v + b * R);
    }
}

template <int N, int R, int NS, int C>
__device__ __forceinline__ void v2_scatter(const cf* v, int t, float* lds) {
    constexpr int B = 32 / R;
    if constexpr (NS == 1) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int base = V2<N>::j(t, b) * R;
#pragma unroll
            for (int r = 0; r < R; ++r) lds[V2<N>::pad(base + r)] = comp<C>(v[b * R + r]);
        }
    } else {
#pragma unroll
        for (int b = 0; b < B; b += 2) {
            const int j = V2<N>::j(t, b);                      // even; j+1 shares j/NS
            const int base = (j / NS) * (NS * R) + (j % NS);
#pragma unroll
            for (int r = 0; r < R; ++r)
                *reinterpret_cast<float2*>(&lds[V2<N>::pad(base + r * NS)]) =
                    make_float2(comp<C>(v[b * R + r]), comp<C>(v[(b + 1) * R + r]));
        }
    }
}

template <int N, int R, int C>
__device__ __forceinline__ void v2_gather(cf* v, int t, const float* lds) {
    constexpr int B = 32 / R;
#pragma unroll
    for (int b = 0; b < B; b += 2) {
        const int j = V2<N>::j(t, b);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float2 p = *reinterpret_cast<const float2*>(&lds[V2<N>::pad(j + r * (N / R))]);
            comp<C>(v[b * R + r]) = p.x;
            comp<C>(v[(b + 1) * R + r]) = p.y;
        }
    }
}

template <int N, bool INV, bool REV, int S = 0>
__device__ __forceinline__ void v2_run(cf* v, int t, float* lds, const cf* __restrict__ tw, bool lds_busy) {
    using PL = Plan<N>;
    constexpr int R = PL::template radix<REV>(S);
    constexpr int NS = PL::template ns_before<REV>(S);
    v2_compute<N, R, NS, INV>(v, t, tw);
    if constexpr (S + 1 < PL::nstages) {
        constexpr int R2 = PL::template radix<REV>(S + 1);
        if (S > 0 || lds_busy) __syncthreads();       // earlier gather finished before the image is overwritten
        v2_scatter<N, R, NS, 0>(v, t, lds);
        __syncthreads();
        v2_gather<N, R2, 0>(v, t, lds);
        __syncthreads();
        v2_scatter<N, R, NS, 1>(v, t, lds);
        __syncthreads();
        v2_gather<N, R2, 1>(v, t, lds);
        v2_run<N, INV, REV, S + 1>(v, t, lds, tw, true);
    }
}

// first-stage load / last-stage store: butterfly pair (b, b+1), point r <-> 16 B at j(t,b) + r*N/R
template <int N, int R> __device__ __forceinline__ void v2_load(cf* v, int t, const cf* __restrict__ src) {
    constexpr int B = 32 / R;
#pragma unroll
    for (int b = 0; b < B; b += 2)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 q = ld16<((SARX_NT_V2 & 1) && N == 16384)>(src + V2<N>::j(t, b) + r * (N / R));
            v[b * R + r] = make_float2(q.x, q.y);
            v[(b + 1) * R + r] = make_float2(q.z, q.w);
        }
}
template <int N, int R> __device__ __forceinline__ void v2_store(const cf* v, int t, cf* __restrict__ dst) {
    constexpr int B = 32 / R;
#pragma unroll
    for (int b = 0; b < B; b += 2)
#pragma unroll
        for (int r = 0; r < R; ++r)
            st16<((SARX_NT_V2 & 2) && N == 16384)>(dst + V2<N>::j(t, b) + r * (N / R),
                make_float4(v[b * R + r].x, v[b * R + r].y, v[(b + 1) * R + r].x, v[(b + 1) * R + r].y));
}

// After the last stage (radix RL) register (b, r) holds output index
//   k = 2t + e + 2T*m,  e = b&1,  m = (b>>1) + (B/2)*r  in [0,16),  B = 32/RL.
template <int N, int RL> __device__ __forceinline__ int v2_reg(int m, int e) {
    constexpr int B = 32 / RL;
    return (2 * (m % (B / 2)) + e) * RL + m / (B / 2);
}

template <int N, int MODE>
__device__ __forceinline__ void v2_row(const RangeArgs& a, int row, bool live, int t, float* my_lds, bool lds_busy) {
    using CFG = V2<N>;
    using PL = Plan<N>;
    constexpr int T = CFG::T;
    const cf* __restrict__ src = a.in + (size_t)row * N;
    cf* __restrict__ dst = a.out + (size_t)row * N;

    constexpr bool FWD_FIRST = (MODE == RG_FFT || MODE == RG_FFT_PHI2 || MODE == RG_FUSED);
    cf v[32];
    if constexpr (FWD_FIRST) {
        constexpr int R0 = PL::template radix<false>(0);
        constexpr int RL = PL::template radix<false>(PL::nstages - 1);
        v2_load<N, R0>(v, t, src);
        v2_run<N, false, false>(v, t, my_lds, a.tw, lds_busy);
        if constexpr (MODE == RG_FFT) {
            if (live) v2_store<N, RL>(v, t, dst);
            return;
        } else {
            // output bins k = 2t + e + 2T*m; m >= 8 are the negative frequencies (fftfreq order)
            const double2 c2 = a.c2[row];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ks = 2 * t - half * (N / 2);     // bin 2t + 16T*half, minus N for the upper half
                FixPhase q0 = phi2_seed(ks, 2 * T, c2, a.df), q1 = phi2_seed(ks + 1, 2 * T, c2, a.df);
#pragma unroll
                for (int mm = 0; mm < 8; ++mm) {
                    const int m = half * 8 + mm;
                    const int r0 = v2_reg<N, RL>(m, 0), r1 = v2_reg<N, RL>(m, 1);
                    v[r0] = cmul(v[r0], q0.next());
                    v[r1] = cmul(v[r1], q1.next());
                    if constexpr (MODE == RG_FFT_PHI2) {
                        if (live) st16<((SARX_NT_V2 & 2) && N == 16384)>(dst + 2 * t + 2 * T * m, make_float4(v[r0].x, v[r0].y, v[r1].x, v[r1].y));
                    }
                }
            }
            if constexpr (MODE == RG_FFT_PHI2) return;
        }
    }
    constexpr bool REV = (MODE == RG_FUSED);
    constexpr int R0 = PL::template radix<REV>(0);
    constexpr int RL = PL::template radix<REV>(PL::nstages - 1);
    if constexpr (!FWD_FIRST) v2_load<N, R0>(v, t, src);
    v2_run<N, true, REV>(v, t, my_lds, a.tw, FWD_FIRST || lds_busy);
    const float s = a.inv_n;
    if constexpr (MODE == RG_IFFT) {
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = make_float2(v[i].x * s, v[i].y * s);
        if (live) v2_store<N, RL>(v, t, dst);
    } else {
        const double2 c3 = a.c3[row];
        FixPhase q0 = phi3_seed(2 * t, 2 * T, c3, a.dt, a.t_start, a.t0);
        FixPhase q1 = phi3_seed(2 * t + 1, 2 * T, c3, a.dt, a.t_start, a.t0);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int r0 = v2_reg<N, RL>(m, 0), r1 = v2_reg<N, RL>(m, 1);
            cf p0 = q0.next(), p1 = q1.next();
            p0.x *= s; p0.y *= s; p1.x *= s; p1.y *= s;
            const cf y0 = cmul(v[r0], p0), y1 = cmul(v[r1], p1);
            if (live) st16<((SARX_NT_V2 & 2) && N == 16384)>(dst + 2 * t + 2 * T * m, make_float4(y0.x, y0.y, y1.x, y1.y));
        }
    }
}

// Persistent workgroups: the grid is sized to what is resident (launcher), each workgroup walks
// line groups g, g + gridDim.x, ...  Re-dispatching a 512-thread / 64 KiB workgroup per line left
// a CU with 1.4 workgroups resident on average instead of 2 (measured, rocprofv3 SQ_WAVE_CYCLES).
template <int N, int MODE>
__global__ __launch_bounds__(V2<N>::THREADS, 4) void range_pass_v2_kernel(RangeArgs a) {
    using CFG = V2<N>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    const int r_in_wg = threadIdx.x / CFG::T;
    const int t = threadIdx.x % CFG::T;
    float* my_lds = lds + r_in_wg * CFG::LDS_PER_ROW;
    const int groups = (a.n_az + CFG::ROWS - 1) / CFG::ROWS;
    for (int g = blockIdx.x; g < groups; g += gridDim.x) {
        int line = g * CFG::ROWS + r_in_wg;
        const bool live = line < a.n_az;
        if (!live) line = a.n_az - 1;              // keep barriers uniform
        const int row = range_row(a, line);
        // make the lane's index opaque per line: otherwise every LDS/global address of the body is
        // loop-invariant, gets hoisted out of this loop and spills (192 B/lane of scratch measured)
        int tt = t;
        asm volatile("" : "+v"(tt));
        v2_row<N, MODE>(a, row, live, tt, my_lds, g != (int)blockIdx.x);
    }
}

template <int N, int MODE> static hipError_t launch_v2(const RangeArgs& a, int cus, hipStream_t st) {
    using CFG = V2<N>;
    auto k = range_pass_v2_kernel<N, MODE>;
    if (CFG::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)CFG::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    const int groups = (a.n_az + CFG::ROWS - 1) / CFG::ROWS;
    int per_cu = (int)((160 * 1024) / CFG::LDS_BYTES);
    if (per_cu > 1024 / CFG::THREADS) per_cu = 1024 / CFG::THREADS;     // 128 VGPRs: 16 waves per CU
    if (per_cu < 1) per_cu = 1;
    int grid = persistent_grid(per_cu, cus, groups);
    if (const char* e = getenv("SARX_V2_WGS_PER_CU")) { const int w = atoi(e); grid = (w <= 0) ? groups : persistent_grid(w, cus, groups); }
    hipLaunchKernelGGL(k, dim3(grid), dim3(CFG::THREADS), CFG::LDS_BYTES, st, a);
    return hipGetLastError();
}
template <int N> static hipError_t launch_v2_mode(int mode, const RangeArgs& a, int cus, hipStream_t st) {
    switch (mode) {
        case RG_FFT: return launch_v2<N, RG_FFT>(a, cus, st);
        case RG_IFFT: return launch_v2<N, RG_IFFT>(a, cus, st);
        case RG_FFT_PHI2: return launch_v2<N, RG_FFT_PHI2>(a, cus, st);
        case RG_IFFT_PHI3: return launch_v2<N, RG_IFFT_PHI3>(a, cus, st);
    }
    return hipErrorInvalidValue;
}

// Only the instantiations that are a default somewhere and fit their register budget are built: one transform per launch
// at 16384 samples.  (The fused mode and the 4096 / 8192 forms spilled 20-140 B/lane at the 128-VGPR cap and lost to the
// 16-point kernel and to range_fused_wl.hip; they were reachable through SARX_RANGE_IMPL=2 only and are gone.)
bool range_v2_supported(int n_rg, int mode) { return n_rg == 16384 && mode != RG_FUSED; }

hipError_t launch_range_pass_v2(int n_rg, int mode, const RangeArgs& a, int cus, hipStream_t st) {
    if (!range_v2_supported(n_rg, mode)) return hipErrorInvalidValue;
    return launch_v2_mode<16384>(mode, a, cus, st);
}

}  // namespace sarx
