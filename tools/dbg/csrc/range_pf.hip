// Range pass, persistent form with register prefetch, for long lines (n_rg >= 8192).
//
// At n_rg = 16384 one line image fills the LDS, so one workgroup (16 waves) is
// resident per CU and the register file (128 VGPRs per lane at that occupancy)
// is what limits keeping more than one line in flight: 32 points/thread kernels
// that hold two lines per CU spill (measured: 188-360 B/lane of scratch, 2x
// slower).  This form keeps 16 points per thread (32 VGPRs) and spends 32 more on
// the NEXT line's samples: a workgroup walks lines row, row+grid, ..., issues the
// next line's global loads before it starts the butterflies of the current one,
// and leaves its stores in flight behind them.  HBM latency and the workgroup
// re-dispatch gap (4 us of 17.5 us per line, measured) disappear behind arithmetic.
#include <cstdlib>
#include "csa_kernels.h"
#include "fft_core.hpp"
#include "phase.hpp"

namespace sarx {

template <int N> struct PfCfg {
    using PL = Plan<N>;
    static constexpr int P = PL::P, T = PL::T;
    static constexpr int LDS_ELEMS = LdsSize<N, 1>::value;
    static constexpr size_t LDS_BYTES = (size_t)LDS_ELEMS * sizeof(cf);
    static_assert(T >= 256, "persistent form is for one line per workgroup");
};

template <int N, bool REV> __device__ __forceinline__ void pf_load(cf* v, int t, const cf* __restrict__ src) {
    using E = Edge<N, REV>;
    constexpr int R0 = E::R_first, P = Plan<N>::P;
#pragma unroll
    for (int b = 0; b < P / R0; ++b)
#pragma unroll
        for (int r = 0; r < R0; ++r) v[b * R0 + r] = src[E::in_index(t, b, r)];
}

// one line: registers v hold the first stage's inputs; results go to dst
template <int N, int MODE>
__device__ __forceinline__ void pf_line(const RangeArgs& a, cf* v, int row, int t, cf* lds, cf* __restrict__ dst) {
    using PL = Plan<N>;
    constexpr int P = PL::P, T = PL::T;
    constexpr bool FWD_FIRST = (MODE == RG_FFT || MODE == RG_FFT_PHI2 || MODE == RG_FUSED);
    if constexpr (FWD_FIRST) {
        using E = Edge<N, false>;
        constexpr int RL = E::R_last, B = P / RL;
        stockham_run<N, 1, false, false>(v, t, 0, lds, a.tw);
        if constexpr (MODE == RG_FFT) {
#pragma unroll
            for (int m = 0; m < P; ++m) dst[t + T * m] = v[(m % B) * RL + m / B];
            return;
        } else {
            const double2 c2 = a.c2[row];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                FixPhase q = phi2_seed(t + half * (P / 2) * T - half * N, T, c2, a.df);
#pragma unroll
                for (int mm = 0; mm < P / 2; ++mm) {
                    const int m = half * (P / 2) + mm;
                    const int reg = (m % B) * RL + m / B;
                    v[reg] = cmul(v[reg], q.next());
                    if constexpr (MODE == RG_FFT_PHI2) dst[t + T * m] = v[reg];
                }
            }
            if constexpr (MODE == RG_FFT_PHI2) return;
            __syncthreads();          // forward half's last gather finished before the image is reused
        }
    }
    constexpr bool REV = (MODE == RG_FUSED);
    using EI = Edge<N, REV>;
    constexpr int RL = EI::R_last, B = P / RL;
    stockham_run<N, 1, true, REV>(v, t, 0, lds, a.tw);
    const float s = a.inv_n;
    if constexpr (MODE == RG_IFFT) {
#pragma unroll
        for (int m = 0; m < P; ++m) {
            const cf y = v[(m % B) * RL + m / B];
            dst[t + T * m] = make_float2(y.x * s, y.y * s);
        }
    } else {
        const double2 c3 = a.c3[row];
        FixPhase q = phi3_seed(t, T, c3, a.dt, a.t_start, a.t0);
#pragma unroll
        for (int m = 0; m < P; ++m) {
            cf ph = q.next();
            ph.x *= s; ph.y *= s;
            dst[t + T * m] = cmul(v[(m % B) * RL + m / B], ph);
        }
    }
}

template <int N, int MODE>
__global__ __launch_bounds__(PfCfg<N>::T, 4) void range_pf_kernel(RangeArgs a) {
    using CFG = PfCfg<N>;
    constexpr int P = CFG::P;
    constexpr bool FWD_FIRST = (MODE == RG_FFT || MODE == RG_FFT_PHI2 || MODE == RG_FUSED);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);

    (void)FWD_FIRST;
    int row = blockIdx.x;
    if (row >= a.n_az) return;
    const int step = gridDim.x;
    // two register sets, used alternately (no copies: a copy would make the compiler wait for the
    // prefetch right where it is issued)
    cf va[P], vb[P];
    pf_load<N, false>(va, threadIdx.x, a.in + (size_t)row * N);
    bool first = true;
    while (true) {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));              // per-line addresses: no hoisting out of the loop
        int next = row + step;
        if (next < a.n_az) pf_load<N, false>(vb, t, a.in + (size_t)next * N);
        if (!first) __syncthreads();             // previous line's last gather done before the image is rewritten
        first = false;
        pf_line<N, MODE>(a, va, row, t, lds, a.out + (size_t)row * N);
        row = next;
        if (row >= a.n_az) break;
        asm volatile("" : "+v"(t));
        next = row + step;
        if (next < a.n_az) pf_load<N, false>(va, t, a.in + (size_t)next * N);
        __syncthreads();
        pf_line<N, MODE>(a, vb, row, t, lds, a.out + (size_t)row * N);
        row = next;
        if (row >= a.n_az) break;
    }
}

static int pf_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int N, int MODE> static hipError_t launch_pf(const RangeArgs& a, hipStream_t st) {
    using CFG = PfCfg<N>;
    auto k = range_pf_kernel<N, MODE>;
    if (CFG::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)CFG::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    int per_cu = (int)((160 * 1024) / CFG::LDS_BYTES);
    if (per_cu > 1024 / CFG::T) per_cu = 1024 / CFG::T;
    if (per_cu < 1) per_cu = 1;
    int grid = per_cu * pf_num_cus();
    if (const char* e = getenv("SARX_PF_WGS_PER_CU")) { const int w = atoi(e); if (w > 0) grid = w * pf_num_cus(); }
    if (grid > a.n_az) grid = a.n_az;
    hipLaunchKernelGGL(k, dim3(grid), dim3(CFG::T), CFG::LDS_BYTES, st, a);
    return hipGetLastError();
}
template <int N> static hipError_t launch_pf_mode(int mode, const RangeArgs& a, hipStream_t st) {
    switch (mode) {
        case RG_FFT: return launch_pf<N, RG_FFT>(a, st);
        case RG_IFFT: return launch_pf<N, RG_IFFT>(a, st);
        case RG_FFT_PHI2: return launch_pf<N, RG_FFT_PHI2>(a, st);
        case RG_IFFT_PHI3: return launch_pf<N, RG_IFFT_PHI3>(a, st);
        case RG_FUSED: return launch_pf<N, RG_FUSED>(a, st);
    }
    return hipErrorInvalidValue;
}

bool range_pf_supported(int n_rg) { return n_rg == 8192 || n_rg == 16384; }

hipError_t launch_range_pass_pf(int n_rg, int mode, const RangeArgs& a, hipStream_t st) {
    switch (n_rg) {
        case 8192: return launch_pf_mode<8192>(mode, a, st);
        case 16384: return launch_pf_mode<16384>(mode, a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace sarx
