// Range pass for line lengths that are not powers of two: direct mixed-radix transforms, no chirp-z.
//
// The reference's native scene has 13200 samples per pulse (sar_ati_dcpa_sim_csa.py:111: int(22e-6 * 600e6)).
// 13200 = 24 * 22 * 25 = (3*8) * (2*11) * (5*5): three Stockham stages whose butterflies are in-register DFTs of length
// 24, 22 and 25 (fft_mixed.hpp), the line exchanged through one LDS image between stages.  One workgroup per line,
// persistent over lines; a stage of radix R has N/R butterflies (550, 600, 528), one per thread of a 640-thread
// workgroup.  The fused mode runs  FFT . Phi_2 . IFFT . Phi_3  (:278-382) in one launch: the inverse uses the radices
// in reverse order, so its first butterfly consumes exactly the registers the forward's last butterfly produced
// (bins t + r N/25) and the spectrum never leaves them.  One HBM round trip of the unpadded line instead of the
// five padded ones of the chirp-z route (32768-point convolution).
//
// Where the fused launch's 0.63 ms at 7199 x 13200 go (ablation builds, tools/ablation_switches.patch; profiles/r03_q_range_mixed_ablation.log):
// without any global load or store 0.54 ms; without the butterflies and twiddles 0.38 ms; with neither - the four LDS crossings,
// eight barriers and the two phase generators alone - 0.25 ms.  The launch is bound by its own instruction stream and LDS
// exchanges running in lockstep (ten waves, one workgroup per CU: the 103 KiB image leaves no room for a second line), not by HBM:
// the copy of the same bytes takes 0.27 ms, and only 0.09 ms of memory time is still exposed beside the arithmetic.
//
// Phases: fp64-seeded fixed-point accumulators along each thread's arithmetic progression of bins / samples
// (phase.hpp); the one bin per thread where the fftfreq sign change falls inside a progression is evaluated directly.
#include "csa_kernels.h"
#include "fft_mixed.hpp"
#include "phase.hpp"
#include <cstdlib>
#include <type_traits>

#ifndef MIX_PREFETCH
#define MIX_PREFETCH 1       // FFT . x . IFFT modes: the next line's first-stage inputs are loaded during this line's inverse half (two bursts)
#endif

namespace sarx {

// Exchange layouts.  A direction runs radices (RA, RB, RC); between its stages the line crosses LDS twice, and each crossing
// has its own 2-D layout chosen so that BOTH sides address it as  per-thread base + compile-time constant * r  (no
// address arithmetic per access; LDS instructions carry the constants as immediate offsets):
//   crossing 1 (after the NS = 1 stage): element w = RA j + r is stored at  r * PITCH + j          (RA rows of N/RA)
//       the next stage reads index j' + r' RA RC, i.e. row j' mod RA, column j' div RA + RC r'
//   crossing 2 (after the NS = RA stage): element w = (j div RA) RA RB + (j mod RA) + RA r lives in block j div RA of
//       N/RC = RA RB elements:  (j div RA) * PITCH + (j mod RA) + RA r;   the last stage reads  j' + r' * PITCH
// The pitches carry a few pad elements (picked with a bank-conflict count of every access: worst case 1.9x the
// conflict-free LDS cycles on the strided reads of crossing 1, about 3 us of LDS time per 13200-sample line in all).
// NPF_: how many of the R1 first-stage samples per thread are prefetched for the next line (the rest is loaded at the top of the line)
// PLANES_: the crossings move re and im separately through a float image (half the bytes: two workgroups per CU), no prefetch
template <int N_, int R1_, int R2_, int R3_, int T_, int P1_, int P2_, int P3_, int P4_, int NPF_ = R1_, bool PLANES_ = false> struct MixCfg {
    static constexpr int N = N_, R1 = R1_, R2 = R2_, R3 = R3_, T = T_, NPF = NPF_;
    static constexpr bool PLANES = PLANES_;
    static_assert(R1 * R2 * R3 == N, "radices must multiply to the line length");
    static_assert(N / R1 <= T && N / R2 <= T && N / R3 <= T, "one butterfly per thread and stage");
    static constexpr int RMAX = (R1 > R2 ? (R1 > R3 ? R1 : R3) : (R2 > R3 ? R2 : R3));
    static constexpr int G1 = N / R1, G2 = N / R2, G3 = N / R3;
    // forward crossings 1, 2 and inverse crossings 1, 2
    static constexpr int PITCH_F1 = G1 + P1_, PITCH_F2 = G3 + P2_, PITCH_I1 = G3 + P3_, PITCH_I2 = G1 + P4_;
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int LDS_ELEMS = cmax(cmax(R1 * PITCH_F1, R3 * PITCH_F2), cmax(R3 * PITCH_I1, R1 * PITCH_I2));
    static constexpr size_t LDS_BYTES = (size_t)LDS_ELEMS * (PLANES_ ? sizeof(float) : sizeof(cf));
};

template <int RA, int PITCH> __device__ __forceinline__ void cross1_write(const cf* v, int j, cf* lds) {
    cf* p = lds + j;
#pragma unroll
    for (int r = 0; r < RA; ++r) p[r * PITCH] = v[r];
}
template <int RA, int RB, int RC, int PITCH> __device__ __forceinline__ void cross1_read(cf* v, int j, const cf* lds) {
    const cf* p = lds + (j % RA) * PITCH + (j / RA);
#pragma unroll
    for (int r = 0; r < RB; ++r) v[r] = p[RC * r];
}
template <int RA, int RB, int PITCH> __device__ __forceinline__ void cross2_write(const cf* v, int j, cf* lds) {
    cf* p = lds + (j / RA) * PITCH + (j % RA);
#pragma unroll
    for (int r = 0; r < RB; ++r) p[RA * r] = v[r];
}
template <int RC, int PITCH> __device__ __forceinline__ void cross2_read(cf* v, int j, const cf* lds) {
    const cf* p = lds + j;
#pragma unroll
    for (int r = 0; r < RC; ++r) v[r] = p[r * PITCH];
}
// The same four crossings one component (re, then im) at a time through a float image of half the bytes, so that two lines - two
// workgroups - are resident on a CU (MixCfg::PLANES): the other workgroup's butterflies run during this one's exchanges and loads.
// What it buys is the memory time (0.09 ms of 0.63 were exposed beside the arithmetic, none is now: the build without loads and
// stores runs as long as the real one, profiles/r03_u_range_mixed_planes_ablation.log) - not more arithmetic per second: ten waves
// sit 3/3/2/2 on the four SIMDs and a second workgroup lands the same way (6/6/4/4), so the two SIMDs with the extra waves bound
// both forms (3 600 vector instructions per thread-line; starting the second workgroup half a line late, or numbering its threads
// from another wave so that its part-filled waves sit elsewhere, changed nothing: profiles/r03_v_*).
template <int COMP> __device__ __forceinline__ float part(const cf& x) { return COMP ? x.y : x.x; }
template <int COMP> __device__ __forceinline__ void set_part(cf& x, float f) { if (COMP) x.y = f; else x.x = f; }
template <int RA, int PITCH, int COMP> __device__ __forceinline__ void cross1_write_p(const cf* v, int j, float* lds) {
    float* p = lds + j;
#pragma unroll
    for (int r = 0; r < RA; ++r) p[r * PITCH] = part<COMP>(v[r]);
}
template <int RA, int RB, int RC, int PITCH, int COMP> __device__ __forceinline__ void cross1_read_p(cf* v, int j, const float* lds) {
    const float* p = lds + (j % RA) * PITCH + (j / RA);
#pragma unroll
    for (int r = 0; r < RB; ++r) set_part<COMP>(v[r], p[RC * r]);
}
template <int RA, int RB, int PITCH, int COMP> __device__ __forceinline__ void cross2_write_p(const cf* v, int j, float* lds) {
    float* p = lds + (j / RA) * PITCH + (j % RA);
#pragma unroll
    for (int r = 0; r < RB; ++r) p[RA * r] = part<COMP>(v[r]);
}
template <int RC, int PITCH, int COMP> __device__ __forceinline__ void cross2_read_p(cf* v, int j, const float* lds) {
    const float* p = lds + j;
#pragma unroll
    for (int r = 0; r < RC; ++r) set_part<COMP>(v[r], p[r * PITCH]);
}
// butterfly j of a stage with radix R after NS = product of earlier radices: twiddle exp(-+ 2 pi i (j mod NS) r / (NS R))
template <int R, int NS, bool INV> __device__ __forceinline__ void mix_twiddle(cf* v, int j) {
    if constexpr (NS > 1) {
        // the argument (< 1/R revolutions) carries 1.2e-7 relative rounding, so even the highest power w^(R-1) is off by
        // less than 1.2e-7 revolutions
        const float x = (float)(j % NS) * (1.0f / (float)(NS * R));
        mix::apply_powers<R>(v, cis_frac(INV ? x : -x));
    }
}

template <class C, int MODE>
__device__ __forceinline__ void range_mixed_body(const RangeArgs& a, char* smem_raw) {
    constexpr int N = C::N, R1 = C::R1, R2 = C::R2, R3 = C::R3, G1 = C::G1, G2 = C::G2, G3 = C::G3;
    constexpr bool PL = C::PLANES;
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    float* ldsf = reinterpret_cast<float*>(smem_raw);
    constexpr bool FWD = (MODE == RG_FFT || MODE == RG_FFT_PHI2 || MODE == RG_FUSED || MODE == RG_CONV);
    constexpr bool BWD = (MODE == RG_IFFT || MODE == RG_IFFT_PHI3 || MODE == RG_FUSED || MODE == RG_CONV);
    // Phi_2 along a thread's bins k = t + r G3: non-negative frequencies for r <= R_LO and negative ones for r >= R_HI
    // whatever t is; in between (at most one r) it depends on the thread.  Each run is one fp64-seeded fixed-point phase
    // accumulator (phase.hpp), the in-between bins are evaluated directly.
    constexpr int HALF = (N + 1) / 2;
    constexpr int R_LO = (HALF - G3) / G3, R_HI = (HALF + G3 - 1) / G3;
    // With one workgroup per CU and barriers between the stages the waves run in lockstep: without a prefetch the CU's memory
    // pipe idles through the whole transform.  The next line's R1 first-stage samples per thread wait in registers instead,
    // requested in two bursts during the inverse half (after the mid-line vector loads of RG_CONV's filter spectrum: vmcnt
    // retires in order, a load issued behind the prefetch would wait for all of it).
    constexpr bool PRE = MIX_PREFETCH && FWD && BWD && !PL;
    const size_t in_ld = (MODE == RG_CONV) ? a.conv_in_ld : (size_t)N;
    auto load_first_stage = [&](cf* dstv, const cf* p, int t, int r0, int r1) {
#pragma unroll
        for (int r = 0; r < R1; ++r)
            if (r >= r0 && r < r1) {
                if constexpr (MODE == RG_CONV) dstv[r] = (t + r * G1 < a.conv_valid) ? ld8<false>(p + t + r * G1) : make_float2(0.f, 0.f);
                else dstv[r] = ld8<false>(p + t + r * G1);
            }
    };
    constexpr int NPF = C::NPF;
    cf nxt[PRE ? NPF : 1];
    if constexpr (PRE) {
        if ((int)blockIdx.x < a.n_az && (int)threadIdx.x < G1)
            load_first_stage(nxt, a.in + (size_t)range_row(a, blockIdx.x) * in_ld, threadIdx.x, 0, NPF);
        __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the loop header's merged wait state is then the back edge's (stores stay in flight)
    }
    // one crossing: write, barrier, read - or, component by component, write re, barrier, read re, barrier, write im, barrier, read im
    // (the reader's registers take the new re while the old im still waits to be written: no more registers than one line share)
    auto cross = [&](int t, int gw, int gr, auto wr, auto rd, auto between) {
        if constexpr (!PL) {
            if (t < gw) wr(std::integral_constant<int, -1>{});
            between();
            __syncthreads();
            if (t < gr) rd(std::integral_constant<int, -1>{});
        } else {
            if (t < gw) wr(std::integral_constant<int, 0>{});
            __syncthreads();
            if (t < gr) rd(std::integral_constant<int, 0>{});
            __syncthreads();
            if (t < gw) wr(std::integral_constant<int, 1>{});
            __syncthreads();
            if (t < gr) rd(std::integral_constant<int, 1>{});
        }
    };
    auto nothing = [] {};
    for (int line = blockIdx.x; line < a.n_az; line += gridDim.x) {
        const int row = range_row(a, line);
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));                     // per-line addresses: nothing hoisted out of the line loop and spilled
        const cf* __restrict__ src = a.in + (size_t)row * in_ld;
        cf* __restrict__ dst = a.out + (size_t)row * (MODE == RG_CONV ? a.conv_out_ld : (size_t)N);
        const int next_line = line + gridDim.x;
        const cf* nsrc = a.in + (size_t)range_row(a, next_line < a.n_az ? next_line : line) * in_ld;   // the last line re-reads itself (never used)
        cf v[C::RMAX];
        // under the 80-VGPR cap of the two-workgroup form the compiler hoisted the phase seeds' line-invariant fp64 products out of the
        // line loop and then spilled them (+7.5 % HBM traffic from scratch): opaque per-line copies of df / dt keep them inside
        double df = a.df, dt = a.dt;
        if constexpr (PL) asm volatile("" : "+s"(df), "+s"(dt));
        if (line != (int)blockIdx.x) __syncthreads();   // the previous line's last reads of the image are finished
        // the row's phase constants through the scalar cache (as vector loads next to their use their latency was exposed per line)
        double2 c2 = make_double2(0, 0), c3 = c2;
        if constexpr (MODE == RG_FFT_PHI2 || MODE == RG_FUSED) c2 = sload_double2(a.c2 + row);
        if constexpr (MODE == RG_IFFT_PHI3 || MODE == RG_FUSED) c3 = sload_double2(a.c3 + row);
        if constexpr (FWD) {
            // stage 1: radix R1, NS = 1, straight from HBM (8 bytes per lane, consecutive lanes consecutive samples)
            if (t < G1) {
                if constexpr (PRE) {
#pragma unroll
                    for (int r = 0; r < NPF; ++r) v[r] = nxt[r];
                    load_first_stage(v, src, t, NPF, R1);
                } else {
                    load_first_stage(v, src, t, 0, R1);      // RG_CONV: the line is shorter than the transform, zeros beyond it are never read
                }
                mix::dft_any<R1, false>(v);
            }
            cross(t, G1, G2,
                  [&](auto cc) { constexpr int CC = decltype(cc)::value;
                                 if constexpr (CC < 0) cross1_write<R1, C::PITCH_F1>(v, t, lds); else cross1_write_p<R1, C::PITCH_F1, CC>(v, t, ldsf); },
                  [&](auto cc) { constexpr int CC = decltype(cc)::value;
                                 if constexpr (CC < 0) cross1_read<R1, R2, R3, C::PITCH_F1>(v, t, lds); else cross1_read_p<R1, R2, R3, C::PITCH_F1, CC>(v, t, ldsf); },
                  nothing);
            // stage 2: radix R2, NS = R1
            if (t < G2) {
                mix_twiddle<R2, R1, false>(v, t);
                mix::dft_any<R2, false>(v);
            }
            __syncthreads();                            // every read of the image is finished
            cross(t, G2, G3,
                  [&](auto cc) { constexpr int CC = decltype(cc)::value;
                                 if constexpr (CC < 0) cross2_write<R1, R2, C::PITCH_F2>(v, t, lds); else cross2_write_p<R1, R2, C::PITCH_F2, CC>(v, t, ldsf); },
                  [&](auto cc) { constexpr int CC = decltype(cc)::value;
                                 if constexpr (CC < 0) cross2_read<R3, C::PITCH_F2>(v, t, lds); else cross2_read_p<R3, C::PITCH_F2, CC>(v, t, ldsf); },
                  nothing);
            // stage 3: radix R3, NS = R1 R2; thread t ends with bins k = t + r G3
            if (t < G3) {
                mix_twiddle<R3, R1 * R2, false>(v, t);
                mix::dft_any<R3, false>(v);
                if constexpr (MODE == RG_CONV) {      // times the filter spectrum, bins k = t + r G3 (the inverse starts from these registers)
                    const cf* __restrict__ mv = a.mulvec;
#pragma unroll
                    for (int r = 0; r < R3; ++r) v[r] = cmul(v[r], mv[t + r * G3]);
                } else if constexpr (MODE == RG_FFT) {
                    if (a.mulvec) {
                        const cf* __restrict__ mv = a.mulvec + (size_t)(row % a.mul_period) * N;
#pragma unroll
                        for (int r = 0; r < R3; ++r) dst[t + r * G3] = cmul(v[r], mv[t + r * G3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < R3; ++r) dst[t + r * G3] = v[r];
                    }
                } else {
                    FixPhase lo = phi2_seed(t, G3, c2, df);                        // numpy.fft.fftfreq order: k, then k - N
                    FixPhase hi = phi2_seed(t + R_HI * G3 - N, G3, c2, df);
#pragma unroll
                    for (int r = 0; r < R3; ++r) {
                        const int k = t + r * G3;
                        cf ph;
                        if (r <= R_LO) ph = lo.next();
                        else if (r >= R_HI) ph = hi.next();
                        else ph = phi2_at(k < HALF ? k : k - N, c2, df);
                        v[r] = cmul(v[r], ph);
                        if constexpr (MODE == RG_FFT_PHI2) dst[k] = v[r];
                    }
                }
            }
        }
        if constexpr (BWD) {
            // inverse, radices reversed: stage 1 radix R3 on samples t + r G3 - in the fused mode the registers as they are
            if (t < G3) {
                if constexpr (!FWD) {
#pragma unroll
                    for (int r = 0; r < R3; ++r) v[r] = src[t + r * G3];
                }
                mix::dft_any<R3, true>(v);
            }
            if constexpr (FWD) __syncthreads();         // forward stage 3's reads of the image are finished
            cross(t, G3, G2,
                  [&](auto cc) { constexpr int CC = decltype(cc)::value;
                                 if constexpr (CC < 0) cross1_write<R3, C::PITCH_I1>(v, t, lds); else cross1_write_p<R3, C::PITCH_I1, CC>(v, t, ldsf); },
                  [&](auto cc) { constexpr int CC = decltype(cc)::value;
                                 if constexpr (CC < 0) cross1_read<R3, R2, R1, C::PITCH_I1>(v, t, lds); else cross1_read_p<R3, R2, R1, C::PITCH_I1, CC>(v, t, ldsf); },
                  [&] { if constexpr (PRE) { if (t < G1) load_first_stage(nxt, nsrc, t, 0, NPF / 2); } });
            // stage 2: radix R2, NS = R3
            if (t < G2) {
                mix_twiddle<R2, R3, true>(v, t);
                mix::dft_any<R2, true>(v);
            }
            __syncthreads();
            cross(t, G2, G1,
                  [&](auto cc) { constexpr int CC = decltype(cc)::value;
                                 if constexpr (CC < 0) cross2_write<R3, R2, C::PITCH_I2>(v, t, lds); else cross2_write_p<R3, R2, C::PITCH_I2, CC>(v, t, ldsf); },
                  [&](auto cc) { constexpr int CC = decltype(cc)::value;
                                 if constexpr (CC < 0) cross2_read<R1, C::PITCH_I2>(v, t, lds); else cross2_read_p<R1, C::PITCH_I2, CC>(v, t, ldsf); },
                  [&] { if constexpr (PRE) { if (t < G1) load_first_stage(nxt, nsrc, t, NPF / 2, NPF); } });
            // stage 3: radix R1, NS = R3 R2; thread t ends with samples n = t + r G1
            if (t < G1) {
                mix_twiddle<R1, R3 * R2, true>(v, t);
                mix::dft_any<R1, true>(v);
                const float s = a.inv_n;
                if constexpr (MODE == RG_CONV) {      // only the cropped window is written
#pragma unroll
                    for (int r = 0; r < R1; ++r) {
                        const int n = t + r * G1 - a.conv_crop0;
                        if (n >= 0 && n < a.conv_out) st8<false>(dst + n, make_float2(v[r].x * s, v[r].y * s));
                    }
                } else if constexpr (MODE == RG_IFFT) {
#pragma unroll
                    for (int r = 0; r < R1; ++r) dst[t + r * G1] = make_float2(v[r].x * s, v[r].y * s);
                } else {
                    FixPhase q = phi3_seed(t, G1, c3, dt, a.t_start, a.t0);
#pragma unroll
                    for (int r = 0; r < R1; ++r) {
                        cf ph = q.next();
                        ph.x *= s; ph.y *= s;
                        const cf y = cmul(v[r], ph);
                        st8<false>(dst + t + r * G1, y);
                    }
                }
            }
        }
    }
}

template <class C, int MODE>
__global__ __launch_bounds__(C::T) void range_mixed_kernel(RangeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    range_mixed_body<C, MODE>(a, smem_raw);
}
// two workgroups per CU (MixCfg::PLANES): ten or twelve waves each, so a SIMD can be asked for six: 80 VGPRs
template <class C, int MODE>
// (five per SIMD = 96 VGPRs, no scratch, would do for the 20 waves of two workgroups only if they spread 5/5/5/5; they sit 6/6/4/4, one
//  workgroup is left per CU and the launch takes 0.73 instead of 0.58 ms: profiles/r04_q_mixed_planes_waves5.log)
__global__ __launch_bounds__(C::T) __attribute__((amdgpu_waves_per_eu(6, 6))) void range_mixed_planes_kernel(RangeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    range_mixed_body<C, MODE>(a, smem_raw);
}

// ---- supported lengths ----------------------------------------------------------------------------------------------
using Mix13200 = MixCfg<13200, 24, 22, 25, 640, 0, 8, 1, 3>;   // pads from the bank-conflict count of tools/lds_layout_sim.py
// the same radices, re / im planes through a 54 KiB float image: two workgroups per CU (pads: tools/lds_layout_sim.py 13200 24 22 25 640 f32)
using Mix13200P = MixCfg<13200, 24, 22, 25, 640, 0, 8, 25, 19, 24, true>;
// 19683 = 27^3: the circular length for the 'same' convolution of a 13200-sample line with the reference's 12001-tap
// matched filter (sar_satellite_sim.py:377-392).  Any length >= 19200 holds it - the wrapped ends of the 25200-sample full
// convolution then fall on the 6000 samples either side of the window that 'same' discards - and of the lengths whose
// line fits LDS (<= 20480) 27^3 is the one whose three stages all have <= 768 butterflies (729): twelve waves, three per
// SIMD, 168 VGPRs.  (19200 = 32 * 24 * 25 needs 800 butterflies in one stage: thirteen waves, four on one SIMD, 128 VGPRs,
// and spilled 32 of them.  Round 4: the re / im-plane form of 27^3 for two workgroups per CU - MixCfg<19683, 27, 27, 27, 768, 0, 2, 0, 2, 27, true>,
// 77 KiB image, 80 VGPRs - spills 224 B per lane and takes the RDA focus from 2.13 to 3.02 ms: profiles/r04_n_rda_planes_and_lanes.log.)
using Mix19683 = MixCfg<19683, 27, 27, 27, 768, 0, 2, 0, 2, 19>;     // 19 of the 27 rows prefetched: every row a 13200-sample line has samples in (27 rows spill at 168 VGPRs)

template <class C, int MODE> static hipError_t launch_mixed(const RangeArgs& a, int cus, hipStream_t st) {
    void (*k)(RangeArgs);
    if constexpr (C::PLANES) k = range_mixed_planes_kernel<C, MODE>; else k = range_mixed_kernel<C, MODE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    if (e != hipSuccess) return e;
    int per_cu = (int)((160 * 1024) / C::LDS_BYTES) < 1 ? 1 : (int)((160 * 1024) / C::LDS_BYTES);
    if (C::PLANES && per_cu > 2) per_cu = 2;            // 80 VGPRs: six waves per SIMD
    const int grid = persistent_grid(per_cu, cus, a.n_az);
    hipLaunchKernelGGL(k, dim3(grid), dim3(C::T), C::LDS_BYTES, st, a);
    return hipGetLastError();
}
template <class C> static hipError_t launch_mixed_mode(int mode, const RangeArgs& a, int cus, hipStream_t st) {
    switch (mode) {
        case RG_FFT: return launch_mixed<C, RG_FFT>(a, cus, st);
        case RG_IFFT: return launch_mixed<C, RG_IFFT>(a, cus, st);
        case RG_FFT_PHI2: return launch_mixed<C, RG_FFT_PHI2>(a, cus, st);
        case RG_IFFT_PHI3: return launch_mixed<C, RG_IFFT_PHI3>(a, cus, st);
        case RG_FUSED: return launch_mixed<C, RG_FUSED>(a, cus, st);
    }
    return hipErrorInvalidValue;
}

bool range_mixed_supported(int n_rg) { return n_rg == 13200; }

// circular convolution of every line with the filter whose m-point spectrum (natural bin order, NOT divided by m) is
// a.mulvec: a.conv_* describe the zero padding and the crop.  m = 19683 only.
bool range_conv_supported(int m) { return m == 19683; }
hipError_t launch_range_conv(int m, const RangeArgs& a, int cus, hipStream_t st) {
    if (m != 19683 || !a.mulvec || a.conv_valid <= 0 || a.conv_valid > m || a.conv_crop0 < 0 || a.conv_out <= 0 ||
        a.conv_crop0 + a.conv_out > m)
        return hipErrorInvalidValue;
    return launch_mixed<Mix19683, RG_CONV>(a, cus, st);
}

hipError_t launch_range_mixed(int n_rg, int mode, const RangeArgs& a, int cus, hipStream_t st) {
    switch (n_rg) {
        case 13200: {
            // fused launch at 7199 x 13200: 0.635 -> 0.595 ms with two workgroups per CU (profiles/r03_v_range_mixed_planes.log; SARX_MIXED_PLANES=0 for A/B)
            static const int planes = [] { const char* e = getenv("SARX_MIXED_PLANES"); return e ? atoi(e) : 1; }();
            if (planes && mode == RG_FUSED) return launch_mixed<Mix13200P, RG_FUSED>(a, cus, st);
            return launch_mixed_mode<Mix13200>(mode, a, cus, st);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace sarx
