// Range passes for n_rg = 16384 on sixteen waves per line, the range spectrum kept in a PERMUTED order between the
// two passes (sar_ati_dcpa_sim_csa.py:278-382).
//
// The spectrum of a range line exists only between the forward transform (+ Phi_2, :278-326) and the inverse
// (+ Phi_3, :331-382); nothing else ever reads it, so its storage order is free.  With the order
//
//        P[kf * 1024 + q * 64 + lane] = X[q + 16 (lane + 64 kf)]        (q = bin mod 16; bin div 16 = lane + 64 kf)
//
// each direction needs ONE workgroup-wide exchange instead of two: the forward transform is decimation in frequency
// over 16 x 1024 and stops where its wave-private 1024-point transforms end (wave q holds X[q + 16 k2] with k2 = lane + 64 kf
// in register kf: a 512-byte contiguous store per wave instruction, the sixteen waves of the workgroup filling one 8 KiB
// chunk per kf together, exactly like the natural-order side), the inverse starts from that arrangement and mirrors it.
// (WP_LAYOUT=0 keeps a wave's 8 KiB contiguous instead, P[q*1024 + k2]: same speed out of place, 2-4 % slower in place.)
//
//   forward (FFT + Phi_2):   load x[n1*1024 + t]  (thread t of 1024, n1 = 0..15: 8 B per lane, 512 B per wave instruction)
//                            radix-16 over n1, twiddle W_N^(t q)                               -> y_q[t]
//                            CROSS exchange (2 barriers): wave w takes line q = w, lane l holds y_w[l + 64 r]
//                            wave-private 1024-point transform as 16 . 4 . 16:
//                              radix-16 over r, twiddle W_1024^(l ka)
//                              lane bits 4,5 <-> register bits 0,1 by v_permlane16_swap / v_permlane32_swap (no LDS),
//                              radix-4 over them, twiddle W_64^(l_lo ke)
//                              one exchange through the wave's own row of the LDS image (no barrier), radix-16
//                                                                                              -> X[w + 16 (lane + 64 kf)]
//                            Phi_2 at the natural-order bin, store P[w*1024 + lane + 64 kf]
//   inverse (IFFT + Phi_3):  the mirror image: load P, inverse radix-16, private exchange, inverse radix-4, swaps back,
//                            inverse radix-16, CROSS exchange back, conj twiddle, inverse radix-16 over q, Phi_3 / N,
//                            store x[n1*1024 + t]
//   fused:                   forward . Phi_2 . inverse . Phi_3 in one launch (the spectrum never leaves the registers)
//
// One persistent 1024-thread workgroup per CU (complex cross image [16][1088] = 136 KiB), four waves per SIMD at
// <= 128 VGPRs.  The NEXT line's samples are kept in flight in 32 further VGPRs while the current line is transformed
// (two bursts of eight loads inside the wave-private phase): with one workgroup per CU and barriers the waves run in
// lockstep, so without it the CU's memory pipe idles during the arithmetic.  Three things decide whether that prefetch
// pays (each measured, profiles/r03_rgbench_*.log): (1) vmcnt retires in order, so the per-row phase constants must come
// through the scalar cache - as a vector load issued after the prefetch they made every wave wait for the whole prefetch
// before Phi_2; (2) the first line is landed before the loop, otherwise the loop header's merged wait state is vmcnt(0) and
// every line waits for the PREVIOUS line's stores (now vmcnt(16): the stores stay in flight); (3) one burst of sixteen
// loads right after the cross exchange is slower than no prefetch at all, two bursts of eight are the fastest placement.
#include <cstdlib>
#include "csa_kernels.h"
#include "fft_core.hpp"
#include "phase.hpp"

// Tuning switches.  The defaults are what tools/rgbench.hip measured best on MI355X (profiles/r03_rgbench_*.log: FFT+Phi2 0.73 ms,
// IFFT+Phi3 0.74 ms against 0.755 / 0.775 without the prefetch and 0.78-0.80 for range_v2.hip on the same box).
#ifndef WP_PREFETCH
#define WP_PREFETCH 6        // next line's loads: 0 none, 1 one burst before the wave-private phase, 2 one burst at the top of the line,
#endif                       // 3 four groups spread over the wave-private phase, 4 two groups in the head + two in it, 5 two bursts of eight (start / middle of it),
                             // 6 the same two bursts taking every other access each
#ifndef WP_LAYOUT
#define WP_LAYOUT 1          // spectrum order between the two passes: 0 = P[q*1024 + k2] (a wave's 8 KiB contiguous), 1 = P'[kf*1024 + q*64 + lane],
#endif                       // k2 = lane + 64 kf (the sixteen waves of a workgroup fill each 8 KiB chunk together, like the natural-order side)
#ifndef WP_HOIST
#define WP_HOIST 4           // bit 0: cross twiddles W_N^(t q) kept in registers (30), bit 1: W_1024^(l ka) (30), bit 2: W_64^(l_lo ke) (6)
#endif
#ifndef WP_NT
#define WP_NT 3              // bit 0: nontemporal line loads, bit 1: nontemporal line stores
#endif

namespace sarx {
namespace wp {

constexpr int N = 16384, M = 1024, THREADS = 1024;
constexpr int ROW = 1088;                                         // complex elements per row of the cross image (1024 + room for the padded private image)
constexpr size_t LDS_BYTES = (size_t)16 * ROW * sizeof(cf);       // 139264
constexpr int PF = 65, PI = 65;                                   // row pitch of the private [16 x 64] exchange image: the compiler pairs the accesses into ds_read2_b64 /
                                                                  // ds_write2_b64 (16-lane groups over 32 banks), for which an odd pitch is conflict-free both ways (66 on the inverse:
                                                                  // SQ_LDS_BANK_CONFLICT = 22 % of its LDS cycles, profiles/r03_pmc_range_kernels.json)

__device__ __forceinline__ cf cmulc(cf a, cf b) {                 // a * conj(b)
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
// w[k] = w1^k, k = 1..15 (w[0] unused): depth-4 multiplication tree
__device__ __forceinline__ void powers16(cf w1, cf* w) {
    w[1] = w1;
#pragma unroll
    for (int k = 2; k < 16; ++k) w[k] = cmul(w[k / 2], w[k - k / 2]);
}
__device__ __forceinline__ cf cis_neg(int num, float inv_den) { return cis_frac(-(float)num * inv_den); }   // exp(-2 pi i num/den), exact fp32 argument

// lane bit 4 <-> register bit 0, lane bit 5 <-> register bit 1 of a 16-register complex array (an involution):
// before: lane l = l_lo + 16 l_hi holds element (l_hi, ka) in v[ka];  after: lane l_lo + 16 (ka & 3) holds it in v[l_hi + 4 (ka >> 2)]
__device__ __forceinline__ void swap_lane45_reg01(cf* v) {
#pragma unroll
    for (int g = 0; g < 16; g += 2) {          // pairs (ka even, ka odd): v_permlane16_swap exchanges odd rows of the first with even rows of the second
        {
            auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[g].x), __float_as_uint(v[g + 1].x), false, false);
            v[g].x = __uint_as_float(r[0]); v[g + 1].x = __uint_as_float(r[1]);
        }
        {
            auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[g].y), __float_as_uint(v[g + 1].y), false, false);
            v[g].y = __uint_as_float(r[0]); v[g + 1].y = __uint_as_float(r[1]);
        }
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (g & 2) continue;                   // pairs (g, g + 2): v_permlane32_swap exchanges the upper half of the first with the lower half of the second
        {
            auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[g].x), __float_as_uint(v[g + 2].x), false, false);
            v[g].x = __uint_as_float(r[0]); v[g + 2].x = __uint_as_float(r[1]);
        }
        {
            auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[g].y), __float_as_uint(v[g + 2].y), false, false);
            v[g].y = __uint_as_float(r[0]); v[g + 2].y = __uint_as_float(r[1]);
        }
    }
}

struct Tw {                 // thread constants (hoisted out of the line loop when WP_HOIST says so)
    cf cw[16];              // W_N^(t q)
    cf t1[16];              // W_1024^(l ka)
    cf t2[4];               // W_64^(l_lo ke)
};
template <int WHICH> __device__ __forceinline__ void make_tw(Tw& tw, int t) {
    const int l = t & 63;
    if constexpr (WHICH & 1) powers16(cis_neg(t, 1.0f / N), tw.cw);
    if constexpr (WHICH & 2) powers16(cis_neg(l, 1.0f / M), tw.t1);
    if constexpr (WHICH & 4) {
        tw.t2[1] = cis_neg(l & 15, 1.0f / 64);
        tw.t2[2] = cmul(tw.t2[1], tw.t2[1]);
        tw.t2[3] = cmul(tw.t2[2], tw.t2[1]);
    }
}

// wave-private forward 1024-point transform: in v[r] = y[l + 64 r], out v[kf] = Y[lane + 64 kf]
template <class HOOK>
__device__ __forceinline__ void fwd1024(cf* v, int l, cf* row, const Tw& tw, HOOK hook) {
    hook(0);
    dft16<false>(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tw.t1[k]);
    hook(1);
    swap_lane45_reg01(v);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dft4<false>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
#pragma unroll
        for (int ke = 1; ke < 4; ++ke) v[4 * c + ke] = cmul(v[4 * c + ke], tw.t2[ke]);
    }
    hook(2);
    // element (ka = (l >> 4) + 4 c, ke, l_lo) -> register l_lo of lane ka + 16 ke
    cf* wr = row + (l & 15) * PF + (l >> 4);
    exchange_sync<true>();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ke = 0; ke < 4; ++ke) wr[4 * c + 16 * ke] = v[4 * c + ke];
    exchange_sync<true>();
    const cf* rd = row + l;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = rd[r * PF];
    hook(3);
    dft16<false>(v);
}
// its mirror: in v[kf] = Y[lane + 64 kf], out v[r] = 1024 * y[l + 64 r]
template <class HOOK>
__device__ __forceinline__ void inv1024(cf* v, int l, cf* row, const Tw& tw, HOOK hook) {
    hook(0);
    dft16<true>(v);
    hook(1);
    cf* wr = row + l;
    exchange_sync<true>();
#pragma unroll
    for (int r = 0; r < 16; ++r) wr[r * PI] = v[r];
    exchange_sync<true>();
    const cf* rd = row + (l & 15) * PI + (l >> 4);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ke = 0; ke < 4; ++ke) v[4 * c + ke] = rd[4 * c + 16 * ke];
    hook(2);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int ke = 1; ke < 4; ++ke) v[4 * c + ke] = cmulc(v[4 * c + ke], tw.t2[ke]);
        dft4<true>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    }
    swap_lane45_reg01(v);
    hook(3);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmulc(v[k], tw.t1[k]);
    dft16<true>(v);
}

// forward head: v[n1] = x[n1*1024 + t] -> wave w holds y_w[l + 64 r]
__device__ __forceinline__ void fwd_head(cf* v, int t, cf* lds, const Tw& tw, bool lead_barrier) {
    dft16<false>(v);
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = cmul(v[q], tw.cw[q]);
    if (lead_barrier) __syncthreads();          // every wave has finished with its row of the previous line
#pragma unroll
    for (int q = 0; q < 16; ++q) lds[q * ROW + t] = v[q];
    __syncthreads();
    const cf* rd = lds + (t >> 6) * ROW + (t & 63);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = rd[64 * r];
}
// inverse tail: wave w holds 1024 * y_w[l + 64 r] -> v[n1] = N * x[n1*1024 + t]
__device__ __forceinline__ void inv_tail(cf* v, int t, cf* lds, const Tw& tw) {
    cf* wr = lds + (t >> 6) * ROW + (t & 63);
    exchange_sync<true>();
#pragma unroll
    for (int r = 0; r < 16; ++r) wr[64 * r] = v[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = lds[q * ROW + t];
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = cmulc(v[q], tw.cw[q]);
    dft16<true>(v);
}

// registers [4 g0, 4 g1) of a line: 16 accesses STRIDE samples apart (1024: natural order, 64: permuted spectrum)
template <bool NT, int STRIDE> __device__ __forceinline__ void load_regs(cf* v, const cf* __restrict__ p, int g0 = 0, int g1 = 4) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i >= 4 * g0 && i < 4 * g1) v[i] = ld8<NT>(p + i * STRIDE);
}

// Phi_2 on the permuted spectrum: register kf of lane `lane` of wave w is bin k = w + 16 lane + 1024 kf (kf >= 8: k - N)
__device__ __forceinline__ void apply_phi2(cf* v, int w, int lane, double2 c2, double df) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        FixPhase q = phi2_seed(w + 16 * lane - half * (N / 2), M, c2, df);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[8 * half + i] = cmul(v[8 * half + i], q.next());
    }
}

}  // namespace wp

// MODE: RG_FFT_PHI2 (natural in, permuted out), RG_IFFT_PHI3 (permuted in, natural out), RG_FUSED (natural in and out),
//       RG_FFT / RG_IFFT: the same without the phase (tests)
template <int MODE>
__global__ __launch_bounds__(wp::THREADS, 4) void range_wp_kernel(RangeArgs a) {
    using namespace wp;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    constexpr bool FWD = (MODE == RG_FFT || MODE == RG_FFT_PHI2 || MODE == RG_FUSED);
    constexpr bool INV = (MODE == RG_IFFT || MODE == RG_IFFT_PHI3 || MODE == RG_FUSED);
    constexpr bool PRE = WP_PREFETCH != 0 && MODE != RG_FUSED;
    constexpr bool NTL = WP_NT & 1, NTS = (WP_NT & 2) != 0;

    Tw tw;
    make_tw<WP_HOIST & 7>(tw, (int)threadIdx.x);

    constexpr int STRIDE = (FWD || WP_LAYOUT == 1) ? M : 64;
    const cf* nsrc = nullptr;                         // this thread's first sample of the line being prefetched
    auto line_ptr = [&](int ln, int t) -> const cf* {
        const cf* p = a.in + (size_t)range_row(a, ln) * N;
        return (FWD || WP_LAYOUT == 1) ? p + t : p + (t >> 6) * M + (t & 63);
    };
    cf nxt[16];
    int line = blockIdx.x;
    if constexpr (PRE) {
        if (line < a.n_az) load_regs<NTL, STRIDE>(nxt, line_ptr(line, threadIdx.x));
        // land the first line here: the compiler merges the loop header's wait state over both predecessors, and with these
        // loads still pending on entry it would wait vmcnt(0) at the top of EVERY line - i.e. for the previous line's stores,
        // which are younger than that line's prefetch - instead of vmcnt(16)
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
    }
    for (; line < a.n_az; line += gridDim.x) {
        const int row = range_row(a, line);
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));                   // keep addresses per-line (no hoisting out of the loop + spilling)
        const int w = t >> 6, l = t & 63;
        make_tw<(~WP_HOIST) & 7>(tw, t);              // whatever is not kept across lines
        cf* __restrict__ dst = a.out + (size_t)row * N;
        cf* myrow = lds + w * ROW;

        cf v[16];
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = nxt[i];
        } else {
            load_regs<NTL, STRIDE>(v, line_ptr(line, t));
        }
        const int next_line = line + gridDim.x;
        const bool have_next = PRE && next_line < a.n_az;
        if constexpr (PRE) { if (have_next) nsrc = line_ptr(next_line, t); }
        // the next line's loads: where[] says after which point of the line they are issued
        auto prefetch_all = [&]() { if (have_next) load_regs<NTL, STRIDE>(nxt, nsrc); };
        auto group = [&](int gidx) { if (have_next) load_regs<NTL, STRIDE>(nxt, nsrc, gidx, gidx + 1); };
        auto parity = [&](int odd) {                  // every other access: each burst spans the whole line
            if (have_next) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((i & 1) == odd) nxt[i] = ld8<NTL>(nsrc + i * STRIDE);
            }
        };
        auto hook = [&](int point) {
            if constexpr (WP_PREFETCH == 3) group(point);
            if constexpr (WP_PREFETCH == 4) { if (point == 1) group(2); if (point == 3) group(3); }
            if constexpr (WP_PREFETCH == 5) { if (point == 0) { group(0); group(1); } if (point == 2) { group(2); group(3); } }
            if constexpr (WP_PREFETCH == 6) { if (point == 0) parity(0); if (point == 2) parity(1); }
        };
        if constexpr (WP_PREFETCH == 2) prefetch_all();
        if constexpr (WP_PREFETCH == 4) group(0);

        if constexpr (FWD) {
            fwd_head(v, t, lds, tw, line != (int)blockIdx.x);
            if constexpr (!INV && WP_PREFETCH == 1) prefetch_all();
            if constexpr (!INV && WP_PREFETCH == 4) group(1);
            fwd1024(v, l, myrow, tw, hook);
            if constexpr (MODE != RG_FFT) apply_phi2(v, w, l, sload_double2(a.c2 + row), a.df);
            if constexpr (!INV) {
#pragma unroll
                for (int kf = 0; kf < 16; ++kf)
#if WP_LAYOUT == 1
                    st8<NTS>(dst + t + kf * M, v[kf]);
#else
                    st8<NTS>(dst + w * M + l + 64 * kf, v[kf]);
#endif
            }
        }
        if constexpr (INV) {
            if constexpr (!FWD) { if (line != (int)blockIdx.x) __syncthreads(); }    // the previous line's column reads of this wave's row are done
            inv1024(v, l, myrow, tw, hook);
            inv_tail(v, t, lds, tw);
            if constexpr (!FWD && WP_PREFETCH == 1) prefetch_all();
            if constexpr (!FWD && WP_PREFETCH == 4) group(1);
            const float sc = a.inv_n;
            if constexpr (MODE == RG_IFFT) {
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) st8<NTS>(dst + t + n1 * M, make_float2(v[n1].x * sc, v[n1].y * sc));
            } else {
                FixPhase q = phi3_seed(t, M, sload_double2(a.c3 + row), a.dt, a.t_start, a.t0);
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    cf p = q.next();
                    p.x *= sc; p.y *= sc;
                    st8<NTS>(dst + t + n1 * M, cmul(v[n1], p));
                }
            }
        }
    }
}


// (The same structure as a fused FFT . Phi_2 . IFFT . Phi_3 launch for 8192-sample lines - 8 x 1024, eight waves per line, two workgroups
// per CU, no scratch once each half rebuilds its twiddles - passed the parity tests and measured 2 % SLOWER than the Stockham kernel of
// csa_kernels.hip, 0.345 vs 0.339 ms per launch, 2.42 vs 2.38 ms per two-channel frame: profiles/r03_wp8_fused_ab.log.  Like the
// 16384-sample fused launch it is bound by instruction issue, not by its exchanges.  Removed; it is in the history at the commit named there.)

bool range_wp_supported(int n_rg) { return n_rg == wp::N; }

template <int MODE> static hipError_t launch_wp(const RangeArgs& a, int cus, hipStream_t st) {
    auto k = range_wp_kernel<MODE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wp::LDS_BYTES);
    if (e != hipSuccess) return e;
    const int grid = persistent_grid(1, cus, a.n_az);
    hipLaunchKernelGGL(k, dim3(grid), dim3(wp::THREADS), wp::LDS_BYTES, st, a);
    return hipGetLastError();
}
hipError_t launch_range_wp(int mode, const RangeArgs& a, int cus, hipStream_t st) {
    switch (mode) {
        case RG_FFT: return launch_wp<RG_FFT>(a, cus, st);
        case RG_IFFT: return launch_wp<RG_IFFT>(a, cus, st);
        case RG_FFT_PHI2: return launch_wp<RG_FFT_PHI2>(a, cus, st);
        case RG_IFFT_PHI3: return launch_wp<RG_IFFT_PHI3>(a, cus, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace sarx
