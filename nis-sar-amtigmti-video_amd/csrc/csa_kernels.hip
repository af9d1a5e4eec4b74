// CSA focusing kernels for gfx950: range (row) passes and azimuth (column-tile)
// passes, each an FFT fused with the phase multiply that follows it in the
// reference (sar_ati_dcpa_sim_csa.py:233-385).  fftshift/ifftshift pairs of the
// reference cancel: phases are evaluated at natural-order bins.
#include <type_traits>
#include "csa_kernels.h"
#include "fft_core.hpp"
#include "phase.hpp"
#include "ati_pixel.hpp"

namespace sarx {

// ------------------------------------------------------------------------------
// range pass: one line (row) per T = N/16 threads, several short lines per workgroup
// ------------------------------------------------------------------------------
template <int N> struct RangeCfg {
    using PL = Plan<N>;
    static constexpr int T = PL::T;
    static constexpr int ROWS = (T >= 256) ? 1 : 256 / T;
    static constexpr int THREADS = T * ROWS;
    static constexpr int LDS_PER_ROW = LdsSize<N, 1>::value;   // cf elements
    static constexpr size_t LDS_BYTES = (size_t)ROWS * LDS_PER_ROW * sizeof(cf);
};

// (the 4096-point convolution needs 129 VGPRs left alone: one register over the fourth wave per SIMD that its four 34 KiB workgroups per CU could use)
template <int N, int MODE>
__global__ __launch_bounds__(RangeCfg<N>::THREADS, (N == 4096 && MODE == RG_CONV) ? 4 : 1) void range_pass_kernel(RangeArgs a) {
    using CFG = RangeCfg<N>;
    using PL = Plan<N>;
    constexpr int P = PL::P, T = PL::T;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);

    const int r_in_wg = threadIdx.x / T;
    const int t = threadIdx.x % T;
    int line = blockIdx.x * CFG::ROWS + r_in_wg;
    const bool live = line < a.n_az;
    if (!live) line = a.n_az - 1;                   // keep barriers uniform
    const int row = range_row(a, line);
    cf* my_lds = lds + r_in_wg * CFG::LDS_PER_ROW;
    // RG_CONV: dense lines of conv_valid samples in, conv_out samples out (leading dimensions of their own); the transform
    // length N only exists in registers and LDS
    const cf* __restrict__ src = a.in + (size_t)row * (MODE == RG_CONV ? a.conv_in_ld : (size_t)N);
    cf* __restrict__ dst = a.out + (size_t)row * (MODE == RG_CONV ? a.conv_out_ld : (size_t)N);

    constexpr bool FWD_FIRST = (MODE == RG_FFT || MODE == RG_FFT_PHI2 || MODE == RG_FUSED || MODE == RG_CONV);
    // the row's phase constants are requested FIRST: vmcnt retires in order, so they arrive with the line itself; next to their
    // use (mid-line) their latency was exposed once or twice per line
    double2 c2 = make_double2(0, 0), c3 = c2;
    if constexpr (MODE == RG_FFT_PHI2 || MODE == RG_FUSED) c2 = a.c2[row];
    if constexpr (MODE == RG_IFFT_PHI3 || MODE == RG_FUSED) c3 = a.c3[row];
    cf v[P];
    if constexpr (FWD_FIRST) {
        using E = Edge<N, false>;
        constexpr int R0 = E::R_first;
#pragma unroll
        for (int b = 0; b < P / R0; ++b)
#pragma unroll
            for (int r = 0; r < R0; ++r) {
                if constexpr (MODE == RG_CONV) {      // zero padding up to N: neither stored nor read
                    const int j = E::in_index(t, b, r);
                    int col = j;
                    if (a.conv_wrap_n > 0) col = (int)((unsigned)(col + a.conv_wrap_c0) % (unsigned)a.conv_wrap_n);
                    v[b * R0 + r] = j < a.conv_valid ? src[col] : make_float2(0.f, 0.f);
                } else {
                    v[b * R0 + r] = src[E::in_index(t, b, r)];
                }
            }
        stockham_run<N, 1, false, false>(v, t, 0, my_lds, a.tw);
        constexpr int RL = E::R_last;
        if constexpr (MODE == RG_FFT) {
            if (live) {
                if (a.mulvec) {       // spectrum of the Bluestein filter fused into the forward transform (general.hip)
                    // all P factors requested before the first product is stored: load, multiply, store per element made every
                    // load wait for the store in front of it (tools/isa_load_waits.py: 17 of 32 loads waited for alone)
                    const cf* __restrict__ mv = a.mulvec + (size_t)(row % a.mul_period) * N;
                    cf f[P];
#pragma unroll
                    for (int b = 0; b < P / RL; ++b)
#pragma unroll
                        for (int r = 0; r < RL; ++r) f[b * RL + r] = mv[E::out_index(t, b, r)];
#pragma unroll
                    for (int b = 0; b < P / RL; ++b)
#pragma unroll
                        for (int r = 0; r < RL; ++r) dst[E::out_index(t, b, r)] = cmul(v[b * RL + r], f[b * RL + r]);
                    return;
                }
#pragma unroll
                for (int b = 0; b < P / RL; ++b)
#pragma unroll
                    for (int r = 0; r < RL; ++r) dst[E::out_index(t, b, r)] = v[b * RL + r];
            }
            return;
        } else if constexpr (MODE == RG_CONV) {
            // spectrum of the matched filter at N points (natural order), one table for every line
            const cf* __restrict__ mv = a.mulvec;
#pragma unroll
            for (int b = 0; b < P / RL; ++b)
#pragma unroll
                for (int r = 0; r < RL; ++r) v[b * RL + r] = cmul(v[b * RL + r], mv[E::out_index(t, b, r)]);
        } else {
            // output bin of register (b, r) is t + T*m with m = b + (P/RL)*r; bins >= N/2 are negative frequencies
            constexpr int B = P / RL;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                FixPhase q = phi2_seed(t + half * (P / 2) * T - half * N, T, c2, a.df);
#pragma unroll
                for (int mm = 0; mm < P / 2; ++mm) {
                    const int m = half * (P / 2) + mm;
                    const int reg = (m % B) * RL + m / B;
                    v[reg] = cmul(v[reg], q.next());
                    if constexpr (MODE == RG_FFT_PHI2) {
                        if (live) dst[t + T * m] = v[reg];
                    }
                }
            }
            if constexpr (MODE == RG_FFT_PHI2) return;
        }
    }
    // inverse half.  In the fused pass the registers already hold the first
    // inverse stage's inputs (plan reversed: remainder radix first).
    constexpr bool REV = (MODE == RG_FUSED || MODE == RG_CONV);
    using EI = Edge<N, REV>;
    if constexpr (!FWD_FIRST) {
        constexpr int R0 = EI::R_first;
#pragma unroll
        for (int b = 0; b < P / R0; ++b)
#pragma unroll
            for (int r = 0; r < R0; ++r) v[b * R0 + r] = src[EI::in_index(t, b, r)];
    } else {
        __syncthreads();      // forward half's last gather finished before the image is reused
    }
    stockham_run<N, 1, true, REV>(v, t, 0, my_lds, a.tw);
    constexpr int RL = EI::R_last;
    const float s = a.inv_n;
    if constexpr (MODE == RG_IFFT) {
#pragma unroll
        for (int i = 0; i < P; ++i) v[i] = make_float2(v[i].x * s, v[i].y * s);
    } else if constexpr (MODE == RG_CONV) {
        // of the circular convolution only the 'same' window [conv_crop0, conv_crop0 + conv_out) is wanted (scipy convolve mode='same')
        constexpr int B = P / RL;
#pragma unroll
        for (int m = 0; m < P; ++m) {
            const int reg = (m % B) * RL + m / B;
            const int j = t + T * m - a.conv_crop0;
            if (live && j >= 0 && j < a.conv_out) dst[j] = make_float2(v[reg].x * s, v[reg].y * s);
        }
        return;
    } else {
        FixPhase q = phi3_seed(t, T, c3, a.dt, a.t_start, a.t0);
        constexpr int B = P / RL;
#pragma unroll
        for (int m = 0; m < P; ++m) {
            const int reg = (m % B) * RL + m / B;
            cf ph = q.next();
            ph.x *= s; ph.y *= s;
            if (live) dst[t + T * m] = cmul(v[reg], ph);
        }
        return;
    }
    if (live) {
#pragma unroll
        for (int b = 0; b < P / RL; ++b)
#pragma unroll
            for (int r = 0; r < RL; ++r) dst[EI::out_index(t, b, r)] = v[b * RL + r];
    }
}

// One workgroup per line group, not persistent: a persistent form of this kernel (2 workgroups per CU walking the lines, <= 128
// VGPRs, no scratch) ran 8192-sample lines 3-12 % SLOWER in every mode (fused 0.362-0.376 vs 0.348-0.351 ms, FFT+Phi2 0.244-0.250
// vs 0.216-0.220 ms at 8192 x 8192; profiles/r03_p_range_persistent_ab.log, ABBA order) - with a few waves per line a fresh
// dispatch staggers the workgroups of a CU, the persistent pair falls into step.
template <int N, int MODE> static hipError_t launch_range(const RangeArgs& a, hipStream_t st) {
    using CFG = RangeCfg<N>;
    auto k = range_pass_kernel<N, MODE>;
    if (CFG::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)CFG::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    const int grid = (a.n_az + CFG::ROWS - 1) / CFG::ROWS;
    hipLaunchKernelGGL(k, dim3(grid), dim3(CFG::THREADS), CFG::LDS_BYTES, st, a);
    return hipGetLastError();
}

template <int N> static hipError_t launch_range_mode(int mode, const RangeArgs& a, hipStream_t st) {
    switch (mode) {
        case RG_FFT: return launch_range<N, RG_FFT>(a, st);
        case RG_IFFT: return launch_range<N, RG_IFFT>(a, st);
        case RG_FFT_PHI2: return launch_range<N, RG_FFT_PHI2>(a, st);
        case RG_IFFT_PHI3: return launch_range<N, RG_IFFT_PHI3>(a, st);
        case RG_FUSED: return launch_range<N, RG_FUSED>(a, st);
        case RG_CONV: return launch_range<N, RG_CONV>(a, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_range_pass(int n_rg, int mode, const RangeArgs& a, hipStream_t st) {
    switch (n_rg) {
        case 16: return launch_range_mode<16>(mode, a, st);
        case 32: return launch_range_mode<32>(mode, a, st);
        case 64: return launch_range_mode<64>(mode, a, st);
        case 128: return launch_range_mode<128>(mode, a, st);
        case 256: return launch_range_mode<256>(mode, a, st);
        case 512: return launch_range_mode<512>(mode, a, st);
        case 1024: return launch_range_mode<1024>(mode, a, st);
        case 2048: return launch_range_mode<2048>(mode, a, st);
        case 4096: return launch_range_mode<4096>(mode, a, st);
        case 8192: return launch_range_mode<8192>(mode, a, st);
        case 16384: return launch_range_mode<16384>(mode, a, st);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------
// azimuth tile pass: [R rows x W cols] tile, FFT of length R along the rows of
// each column.  Rows of the tile are  in_base + m*in_stride  of the image, so a
// long azimuth FFT (n_az = R_A * R_B) is two such launches (four-step) with the
// inter-step twiddle and the row permutation folded into the addressing:
//   step A: q in [0,S):   rows q + m*S (m < R_A = n/S)  -> same rows, * W_n^(+-q*m_out)
//   step B: q in [0,n/S): rows q*S + m (m < S)          -> rows q + m_out*(n/S), * Phi_1 | * 1/n
// Every global access is a W*8-byte contiguous row segment.
// ------------------------------------------------------------------------------
template <int R, int W> struct AzCfg {
    using PL = Plan<R>;
    static constexpr int TPC = PL::T;                 // threads per column
    static constexpr int THREADS = TPC * W;
    static constexpr size_t LDS_BYTES = (PL::nstages > 1) ? (size_t)LdsSize<R, W>::value * sizeof(cf) : 0;
};

// Image loads / stores of the azimuth tiles, nontemporal on request: an image of a gigabyte or more is not re-read before
// 2+ GiB of other traffic has passed, so keeping its lines in L2 / the Infinity Cache only evicts what could be reused;
// on the 16384^2 tile copies of tools/membench.hip nontemporal accesses are 4.5-7.6 % faster (0.758 / 0.778 vs 0.820 / 0.815 ms).
typedef float nt_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf ld_stream(const cf* p, bool nt) {
    if (nt) { const nt_v2f v = __builtin_nontemporal_load(reinterpret_cast<const nt_v2f*>(p)); return make_float2(v.x, v.y); }
    return *p;
}
__device__ __forceinline__ void st_stream(cf* p, cf x, bool nt) {
    if (nt) __builtin_nontemporal_store(nt_v2f{x.x, x.y}, reinterpret_cast<nt_v2f*>(p));
    else *p = x;
}

template <int R, int W, bool INV, int EPI>
__global__ __launch_bounds__((AzCfg<R, W>::THREADS)) void az_tile_kernel(AzArgs a) {
    using PL = Plan<R>;
    using E = Edge<R, false>;
    constexpr int P = PL::P;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);

    const int c = threadIdx.x % W;
    const int t = threadIdx.x / W;
    const int col = blockIdx.x * W + c;
    const int q = blockIdx.y + a.q0;
    const size_t in_base = (size_t)q * a.in_q_stride;
    const size_t out_base = (size_t)q * a.out_q_stride;

    cf v[P];
    constexpr int R0 = E::R_first;
#pragma unroll
    for (int b = 0; b < P / R0; ++b)
#pragma unroll
        for (int r = 0; r < R0; ++r) {
            const int mi = E::in_index(t, b, r);
            const size_t rowi = in_base + (size_t)mi * a.in_m_stride;
            cf x;
            if constexpr (EPI == AZ_EPI_TWIDDLE_PADIN) {   // copy-in, zero padding and pre-chirp of general.hip fused into the first step
                // Only the load sits under the bounds test; the weights are applied below, once every load of the tile has been issued.  With the multiply inside this branch each of a thread's 16 loads was waited for before the next was
                // issued (0.306 against 0.211 ms for the same tiles without the branch, 32768 x 2048: profiles/r05_bc_*).
                x = make_float2(0.f, 0.f);
                if (rowi < (size_t)a.io_rows && col < a.io_cols) {
                    size_t srow = rowi + (size_t)a.io_shift_in;             // circular row shift of the source (0 = none)
                    if (srow >= (size_t)a.io_rows) srow -= (size_t)a.io_rows;
                    x = a.in[srow * a.io_ld + col];
                }
            } else if constexpr (EPI == AZ_EPI_TWIDDLE_ROWSIN) {
                x = rowi < (size_t)a.io_rows ? a.in[rowi * a.n_rg + col] : make_float2(0.f, 0.f);
            } else if constexpr (EPI == AZ_EPI_TWCOL) {   // zero-padded line: the padding is not stored, let alone read
                x = (a.valid_len && mi * a.n_rg + col >= a.valid_len) ? make_float2(0.f, 0.f) : a.in[rowi * a.n_rg + col];
            } else {
                x = ld_stream(a.in + rowi * a.n_rg + col, a.nt);
            }
            if constexpr (EPI == AZ_EPI_PROCOL) {      // inverse of the 32768-point line split: W_M^(+-col*m_in) first
                const float rev = (float)(col * mi) * a.tw_scale;
                x = cmul(x, cis_frac(INV ? rev : -rev));
            }
            v[b * R0 + r] = x;
        }
    if constexpr (EPI == AZ_EPI_TWIDDLE_PADIN) {       // the copy-in's weights (see above)
        if (a.hamming_inv > 0.f || a.rowvec) {
#pragma unroll
            for (int b = 0; b < P / R0; ++b)
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    const size_t rowi = in_base + (size_t)E::in_index(t, b, r) * a.in_m_stride;
                    if (rowi >= (size_t)a.io_rows) continue;                // padding rows: zeros, and rowvec has io_rows entries
                    cf& x = v[b * R0 + r];
                    if (a.hamming_inv > 0.f) {                              // weight of the SOURCE row
                        size_t srow = rowi + (size_t)a.io_shift_in;
                        if (srow >= (size_t)a.io_rows) srow -= (size_t)a.io_rows;
                        const float w = fmaf(-0.46f, __builtin_amdgcn_cosf((float)srow * a.hamming_inv), 0.54f);
                        x.x *= w; x.y *= w;
                    } else {
                        x = cmul(x, a.rowvec[rowi]);
                    }
                }
        }
    }
    float ati_thr = 0.f;
    double ati_re = 0.0, ati_im = 0.0;
    if constexpr (EPI == AZ_EPI_SCALE_ATI) {       // mask threshold: max over the shards channel 1's focus left
        __shared__ float s_m[AzCfg<R, W>::THREADS / 64 > 0 ? AzCfg<R, W>::THREADS / 64 : 1];
        float m = 0.f;
        for (unsigned k = threadIdx.x; k < MAX_SHARDS; k += AzCfg<R, W>::THREADS) m = fmaxf(m, a.ati_thr[32 * k]);
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
        __syncthreads();
        m = s_m[0];
        for (int k = 1; k < AzCfg<R, W>::THREADS / 64; ++k) m = fmaxf(m, s_m[k]);
        ati_thr = m * a.ati_frac;
    }
    // SCALE_ATI: the first channel's samples this thread will need, requested before the transform so that they arrive behind it
    // (0.39 -> 0.35 ms for the launch at 8192^2: 135 VGPRs instead of 113, three waves per SIMD instead of four, and still faster)
    cf s1v[EPI == AZ_EPI_SCALE_ATI ? P : 1];
    if constexpr (EPI == AZ_EPI_SCALE_ATI) {
        constexpr int RLp = E::R_last;
#pragma unroll
        for (int b = 0; b < P / RLp; ++b)
#pragma unroll
            for (int r = 0; r < RLp; ++r)
                s1v[b * RLp + r] = a.ati_s1[(out_base + (size_t)E::out_index(t, b, r) * a.out_m_stride) * a.n_rg + col];
    }
    stockham_run<R, W, INV, false>(v, t, c, lds, a.tw_r);
    constexpr int RL = E::R_last;
    float vmax = 0.f;
#pragma unroll
    for (int b = 0; b < P / RL; ++b)
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            const int m = E::out_index(t, b, r);
            const size_t rowo = out_base + (size_t)m * a.out_m_stride;
            cf x = v[b * RL + r];
            if constexpr (EPI == AZ_EPI_TWIDDLE || EPI == AZ_EPI_TWIDDLE_PADIN || EPI == AZ_EPI_TWIDDLE_ROWSIN) {
                // four-step twiddle W_n^(q*m): q*m < n_az <= 2^14 and 1/n_az is a power of two, so the
                // argument is exact in fp32 (HW sine/cosine take revolutions; no table load in this pass)
#if SARX_HW_TWIDDLE
                const float rev = (float)(q * m) * a.scale;
                x = cmul(x, cis_frac(INV ? rev : -rev));
#else
                cf w = a.tw_n[(size_t)q * m];
                if (INV) w = cconj(w);
                x = cmul(x, w);
#endif
            } else if constexpr (EPI == AZ_EPI_PHI1) {
                x = cmul(x, phi1(col, a.c1[rowo], a.dt, a.t_start));
            } else if constexpr (EPI == AZ_EPI_SCALE_ATI) {
                x.x *= a.scale; x.y *= a.scale;
                const size_t o = rowo * a.n_rg + col;
                Pix px;
                ati_pixel<false>(s1v[b * RL + r], x, a.ati_cc, a.ati_cs, px);
                a.ati_phase[o] = px.m1 > ati_thr ? px.phase : 0.f;         // (:447-449)
                a.ati_m1[o] = px.m1;
                a.ati_dm[o] = px.dm;
                ati_re += px.sre; ati_im += px.sim;
                if (!a.ati_keep_image) continue;
            } else if constexpr (EPI == AZ_EPI_SCALE_LOOK) {
                x.x *= a.scale; x.y *= a.scale;
                if (a.max_out) vmax = fmaxf(vmax, hypotf(x.x, x.y));
                // |x|^2 summed over the `look` consecutive columns of this row (consecutive lanes): fixed xor tree, bitwise
                // reproducible; the lane of the group's first column stores the row-wise partial
                float pw = fmaf(x.x, x.x, x.y * x.y);
                // the xor tree over lanes 1, 2, 4, 8 as DPP moves (quad permutes, then row_half_mirror / row_mirror: after the first
                // two steps a quad's lanes hold equal sums, so the mirrors pair the same partial sums the xor would): the same
                // additions in the same order as __shfl_xor - bit-identical - without 4 ds_bpermute per sample
                auto dpp_add = [&](auto ctrl) {
                    pw += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pw), decltype(ctrl)::value, 0xF, 0xF, false));
                };
                if (a.look > 1) dpp_add(std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
                if (a.look > 2) dpp_add(std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
                if (a.look > 4) dpp_add(std::integral_constant<int, 0x141>{});      // row_half_mirror
                if (a.look > 8) dpp_add(std::integral_constant<int, 0x140>{});      // row_mirror
                for (int off = 16; off < a.look; off <<= 1) pw += __shfl_xor(pw, off, 64);
                if ((c & (a.look - 1)) == 0) a.look_part[rowo * (size_t)(a.n_rg / a.look) + col / a.look] = pw;
            } else if constexpr (EPI == AZ_EPI_SCALE || EPI == AZ_EPI_PROCOL) {
                if constexpr (EPI == AZ_EPI_PROCOL) {      // only the cropped part of the line is wanted
                    if (a.valid_len && m * a.n_rg + col >= a.valid_len) continue;
                }
                x.x *= a.scale; x.y *= a.scale;
                if constexpr (EPI == AZ_EPI_SCALE) { if (a.max_out) vmax = fmaxf(vmax, hypotf(x.x, x.y)); }
            } else if constexpr (EPI == AZ_EPI_SCALE_ROWSOUT) {
                if (rowo >= (size_t)a.io_rows) continue;
                x.x *= a.scale; x.y *= a.scale;
            } else if constexpr (EPI == AZ_EPI_ROWVEC) {
                x = cmul(x, a.rowvec[rowo]);
            } else if constexpr (EPI == AZ_EPI_CROPOUT || EPI == AZ_EPI_CROPOUT_PHI1 || EPI == AZ_EPI_CROPOUT_MAG) {
                // post-chirp, scale and crop of general.hip fused into the last step
                if (rowo < (size_t)a.io_rows && col < a.io_cols) {
                    x.x *= a.scale; x.y *= a.scale;
                    if (a.rowvec) x = cmul(x, a.rowvec[rowo]);
                    if constexpr (EPI == AZ_EPI_CROPOUT_PHI1) x = cmul(x, phi1(col, a.c1[rowo], a.dt, a.t_start));
                    size_t drow = rowo + (size_t)a.io_shift_out;            // circular row shift of the destination (0 = none)
                    if (drow >= (size_t)a.io_rows) drow -= (size_t)a.io_rows;
                    if constexpr (EPI == AZ_EPI_CROPOUT_MAG) a.out_mag[drow * a.io_ld + col] = hypotf(x.x, x.y);
                    else a.out[drow * a.io_ld + col] = x;
                }
                continue;
            } else if constexpr (EPI == AZ_EPI_TWCOL) {    // 32768-point line as 128 x 256: twiddle W_M^(+-col*m)
                const float rev = (float)(col * m) * a.tw_scale;
                x = cmul(x, cis_frac(INV ? rev : -rev));
            }
            st_stream(a.out + rowo * a.n_rg + col, x, a.nt);
        }
    if constexpr (EPI == AZ_EPI_SCALE_ATI) {
        for (int off = 32; off > 0; off >>= 1) { ati_re += __shfl_xor(ati_re, off, 64); ati_im += __shfl_xor(ati_im, off, 64); }
        if ((threadIdx.x & 63) == 0)
            a.ati_part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (AzCfg<R, W>::THREADS / 64) + (threadIdx.x >> 6)] = make_double2(ati_re, ati_im);
    }
    if constexpr (EPI == AZ_EPI_SCALE || EPI == AZ_EPI_SCALE_LOOK) {
        // max |image| for the 5 % mask of the ATI products (sar_ati_dcpa_sim_csa.py:447), taken while the image is written so
        // that the ATI launch can mask in the same pass: the same hypotf of the same floats that launch computes
        if (a.max_out) {
            for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
            // one device-scope atomic per wave on ONE address costs ~11 ns each, serialised (65 536 of them at 8192^2 = 0.7 ms):
            // 256 shards, one 128-byte line each, maxed again by the consumer
            if ((threadIdx.x & 63) == 0)
                atomicMax(a.max_out + 32u * ((blockIdx.x * 7u + blockIdx.y * 13u + (threadIdx.x >> 6)) & (MAX_SHARDS - 1u)), __float_as_uint(vmax));
        }
    }
}

template <int R, int W, bool INV, int EPI> static hipError_t launch_az_one(const AzArgs& a, int nq, hipStream_t st) {
    using CFG = AzCfg<R, W>;
    dim3 grid(a.n_rg / W, nq);
    hipLaunchKernelGGL((az_tile_kernel<R, W, INV, EPI>), grid, dim3(CFG::THREADS), CFG::LDS_BYTES, st, a);
    return hipGetLastError();
}
template <int R, int W> static hipError_t launch_az_rw(bool inv, int epi, const AzArgs& a, int nq, hipStream_t st) {
    if (!inv) {
        switch (epi) {
            case AZ_EPI_NONE: return launch_az_one<R, W, false, AZ_EPI_NONE>(a, nq, st);
            case AZ_EPI_TWIDDLE: return launch_az_one<R, W, false, AZ_EPI_TWIDDLE>(a, nq, st);
            case AZ_EPI_PHI1: return launch_az_one<R, W, false, AZ_EPI_PHI1>(a, nq, st);
            case AZ_EPI_TWCOL: return launch_az_one<R, W, false, AZ_EPI_TWCOL>(a, nq, st);
            case AZ_EPI_ROWVEC: return launch_az_one<R, W, false, AZ_EPI_ROWVEC>(a, nq, st);
            case AZ_EPI_TWIDDLE_PADIN: return launch_az_one<R, W, false, AZ_EPI_TWIDDLE_PADIN>(a, nq, st);
            case AZ_EPI_TWIDDLE_ROWSIN: return launch_az_one<R, W, false, AZ_EPI_TWIDDLE_ROWSIN>(a, nq, st);
            case AZ_EPI_CROPOUT:      // last step of a forward two-step transform (R >= 16) into a dense array, rows rotated (RDA)
                if constexpr (R >= 16 && W == 32) return launch_az_one<R, W, false, AZ_EPI_CROPOUT>(a, nq, st); else return hipErrorInvalidValue;
        }
    } else {
        switch (epi) {
            case AZ_EPI_NONE: return launch_az_one<R, W, true, AZ_EPI_NONE>(a, nq, st);
            case AZ_EPI_TWIDDLE: return launch_az_one<R, W, true, AZ_EPI_TWIDDLE>(a, nq, st);
            case AZ_EPI_SCALE: return launch_az_one<R, W, true, AZ_EPI_SCALE>(a, nq, st);
            case AZ_EPI_SCALE_LOOK: return launch_az_one<R, W, true, AZ_EPI_SCALE_LOOK>(a, nq, st);
            case AZ_EPI_SCALE_ATI: if constexpr (AzCfg<R, W>::THREADS % 64 == 0) return launch_az_one<R, W, true, AZ_EPI_SCALE_ATI>(a, nq, st); else return hipErrorInvalidValue;
            case AZ_EPI_PROCOL: return launch_az_one<R, W, true, AZ_EPI_PROCOL>(a, nq, st);
            case AZ_EPI_CROPOUT: return launch_az_one<R, W, true, AZ_EPI_CROPOUT>(a, nq, st);
            case AZ_EPI_SCALE_ROWSOUT: return launch_az_one<R, W, true, AZ_EPI_SCALE_ROWSOUT>(a, nq, st);
            case AZ_EPI_CROPOUT_PHI1: return launch_az_one<R, W, true, AZ_EPI_CROPOUT_PHI1>(a, nq, st);
            case AZ_EPI_CROPOUT_MAG: return launch_az_one<R, W, true, AZ_EPI_CROPOUT_MAG>(a, nq, st);
        }
    }
    return hipErrorInvalidValue;
}
template <int R> static hipError_t launch_az_r(int w, bool inv, int epi, const AzArgs& a, int nq, hipStream_t st) {
    switch (w) {
        case 16: return launch_az_rw<R, 16>(inv, epi, a, nq, st);
        case 32: return launch_az_rw<R, 32>(inv, epi, a, nq, st);
        case 64: return launch_az_rw<R, 64>(inv, epi, a, nq, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_az_tile(int r, int w, bool inv, int epi, const AzArgs& a, int nq, hipStream_t st) {
    switch (r) {
        case 2: return launch_az_r<2>(w, inv, epi, a, nq, st);
        case 4: return launch_az_r<4>(w, inv, epi, a, nq, st);
        case 8: return launch_az_r<8>(w, inv, epi, a, nq, st);
        case 16: return launch_az_r<16>(w, inv, epi, a, nq, st);
        case 32: return launch_az_r<32>(w, inv, epi, a, nq, st);
        case 64: return launch_az_r<64>(w, inv, epi, a, nq, st);
        case 128: return launch_az_r<128>(w, inv, epi, a, nq, st);
        case 256: return w == 32 ? launch_az_rw<256, 32>(inv, epi, a, nq, st) : hipErrorInvalidValue;   // 32768-row columns (general.hip)
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------
// chirp-z middle step: FFT_R . filter spectrum . IFFT_R . conjugate twiddle on a [R rows x W cols] tile (see csa_kernels.h)
// ------------------------------------------------------------------------------
template <int R, int W>
__global__ __launch_bounds__((AzCfg<R, W>::THREADS)) void az_conv_kernel(AzArgs a) {
    using PL = Plan<R>;
    using EF = Edge<R, false>;
    using EI = Edge<R, true>;
    constexpr int P = PL::P;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    const int c = threadIdx.x % W;
    const int t = threadIdx.x / W;
    const int col = blockIdx.x * W + c;
    const int q = blockIdx.y, ra = gridDim.y;
    const size_t base = (size_t)q * R;
    cf v[P];
    constexpr int R0 = EF::R_first;
#pragma unroll
    for (int b = 0; b < P / R0; ++b)
#pragma unroll
        for (int r = 0; r < R0; ++r) v[b * R0 + r] = a.in[(base + EF::in_index(t, b, r)) * a.n_rg + col];
    stockham_run<R, W, false, false>(v, t, c, lds, a.tw_r);
    constexpr int RL = EF::R_last;             // == the reversed plan's first radix: the registers feed the inverse as they are
    static_assert(EF::R_last == EI::R_first, "reversed plan starts with the forward plan's last radix");
#pragma unroll
    for (int b = 0; b < P / RL; ++b)
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            const int k2 = EF::out_index(t, b, r);
            v[b * RL + r] = cmul(v[b * RL + r], a.rowvec[q + ra * k2]);
        }
    if constexpr (PL::nstages > 1) __syncthreads();      // the forward's last gather is finished before the image is reused
    stockham_run<R, W, true, true>(v, t, c, lds, a.tw_r);
    constexpr int RLI = EI::R_last;
#pragma unroll
    for (int b = 0; b < P / RLI; ++b)
#pragma unroll
        for (int r = 0; r < RLI; ++r) {
            const int m = EI::out_index(t, b, r);
            const float rev = (float)(q * m) * a.tw_scale;          // exact: q m < M, 1/M a power of two
            a.out[(base + m) * a.n_rg + col] = cmul(v[b * RLI + r], cis_frac(rev));
        }
}
template <int R> static hipError_t launch_az_conv_r(int ra, const AzArgs& a, hipStream_t st) {
    using CFG = AzCfg<R, 32>;
    dim3 grid(a.n_rg / 32, ra);
    hipLaunchKernelGGL((az_conv_kernel<R, 32>), grid, dim3(CFG::THREADS), CFG::LDS_BYTES, st, a);
    return hipGetLastError();
}
hipError_t launch_az_conv(int s, int ra, const AzArgs& a, hipStream_t st) {
    switch (s) {
        case 16: return launch_az_conv_r<16>(ra, a, st);
        case 32: return launch_az_conv_r<32>(ra, a, st);
        case 64: return launch_az_conv_r<64>(ra, a, st);
        case 128: return launch_az_conv_r<128>(ra, a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace sarx
