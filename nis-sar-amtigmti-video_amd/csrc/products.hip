// Element-wise / streaming kernels around the CSA focus: the ATI interferogram +
// DPCA difference products (sar_ati_dcpa_sim_csa.py:414-419, viewer :42-52), the
// magnitude mask (:447-449), the LDS corner turn, multilook and noise fill.
// All are HBM-bound: 16-byte accesses per lane, grid-stride, no MFMA.
#include "csa_kernels.h"
#include "ati_pixel.hpp"

namespace sarx {

typedef float2 cf;
static constexpr int ATI_THREADS = 256;
static constexpr int ATI_MAX_BLOCKS = 4096;

int ati_blocks(size_t n) {
    size_t quads = (n + 3) / 4;
    size_t b = (quads + ATI_THREADS - 1) / ATI_THREADS;
    if (b > ATI_MAX_BLOCKS) b = ATI_MAX_BLOCKS;
    if (b < 1) b = 1;
    return (int)b;
}

typedef float nt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float4* p, bool nt) {
    if (nt) { const nt_v4f v = __builtin_nontemporal_load(reinterpret_cast<const nt_v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *p;
}
// Four pixels per lane: two 16-byte loads per channel, one 16-byte store per fp32 plane (28 B/pixel in all).
template <bool OPT> __global__ __launch_bounds__(ATI_THREADS) void ati_dpca_kernel(AtiArgs a) {
    const size_t nquad = a.n / 4;
    const size_t stride = (size_t)gridDim.x * ATI_THREADS;
    float vmax = 0.f;
    double sre = 0.0, sim = 0.0;
    const float4* s1 = reinterpret_cast<const float4*>(a.s1);
    const float4* s2 = reinterpret_cast<const float4*>(a.s2);
    // masked variant: max |slc1| is already known (the focus emitted it), thr = float32 product as the host facade computes it
    const bool masked = a.thr_max != nullptr;
    float thr = -1.f;
    if (masked) {                                        // max over the MAX_SHARDS partial maxima the focus left (one per 128-byte line)
        __shared__ float s_thr[ATI_THREADS / 64];
        float m = 0.f;
        for (unsigned k = threadIdx.x; k < MAX_SHARDS; k += ATI_THREADS) m = fmaxf(m, a.thr_max[32 * k]);
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if ((threadIdx.x & 63) == 0) s_thr[threadIdx.x >> 6] = m;
        __syncthreads();
        m = s_thr[0];
        for (int k = 1; k < ATI_THREADS / 64; ++k) m = fmaxf(m, s_thr[k]);
        thr = m * a.mask_frac;
    }
    for (size_t i = (size_t)blockIdx.x * ATI_THREADS + threadIdx.x; i < nquad; i += stride) {
        const float4 x0 = ld4(s1 + 2 * i, a.nt), x1 = ld4(s1 + 2 * i + 1, a.nt), y0 = ld4(s2 + 2 * i, a.nt), y1 = ld4(s2 + 2 * i + 1, a.nt);
        Pix p[4];
        ati_pixel<OPT>(make_float2(x0.x, x0.y), make_float2(y0.x, y0.y), a.cal_c, a.cal_s, p[0]);
        ati_pixel<OPT>(make_float2(x0.z, x0.w), make_float2(y0.z, y0.w), a.cal_c, a.cal_s, p[1]);
        ati_pixel<OPT>(make_float2(x1.x, x1.y), make_float2(y1.x, y1.y), a.cal_c, a.cal_s, p[2]);
        ati_pixel<OPT>(make_float2(x1.z, x1.w), make_float2(y1.z, y1.w), a.cal_c, a.cal_s, p[3]);
        if (masked) {
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k].phase = p[k].m1 > thr ? p[k].phase : 0.f;
        }
        reinterpret_cast<float4*>(a.ati_phase)[i] = make_float4(p[0].phase, p[1].phase, p[2].phase, p[3].phase);
        reinterpret_cast<float4*>(a.mag1)[i] = make_float4(p[0].m1, p[1].m1, p[2].m1, p[3].m1);
        reinterpret_cast<float4*>(a.dpca_mag)[i] = make_float4(p[0].dm, p[1].dm, p[2].dm, p[3].dm);
        if (OPT) {
            if (a.interf) {
                reinterpret_cast<float4*>(a.interf)[2 * i] = make_float4(p[0].interf.x, p[0].interf.y, p[1].interf.x, p[1].interf.y);
                reinterpret_cast<float4*>(a.interf)[2 * i + 1] = make_float4(p[2].interf.x, p[2].interf.y, p[3].interf.x, p[3].interf.y);
            }
            if (a.diff) {
                reinterpret_cast<float4*>(a.diff)[2 * i] = make_float4(p[0].diff.x, p[0].diff.y, p[1].diff.x, p[1].diff.y);
                reinterpret_cast<float4*>(a.diff)[2 * i + 1] = make_float4(p[2].diff.x, p[2].diff.y, p[3].diff.x, p[3].diff.y);
            }
            if (a.mag2) reinterpret_cast<float4*>(a.mag2)[i] = make_float4(p[0].m2, p[1].m2, p[2].m2, p[3].m2);
            if (a.ph1) reinterpret_cast<float4*>(a.ph1)[i] = make_float4(p[0].p1, p[1].p1, p[2].p1, p[3].p1);
            if (a.ph2) reinterpret_cast<float4*>(a.ph2)[i] = make_float4(p[0].p2, p[1].p2, p[2].p2, p[3].p2);
            if (a.dpca_phase) reinterpret_cast<float4*>(a.dpca_phase)[i] = make_float4(p[0].dp, p[1].dp, p[2].dp, p[3].dp);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            vmax = fmaxf(vmax, p[k].m1);
            sre += p[k].sre;
            sim += p[k].sim;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {   // tail pixels
        const size_t i = nquad * 4 + threadIdx.x;
        Pix p;
        ati_pixel<OPT>(a.s1[i], a.s2[i], a.cal_c, a.cal_s, p);
        if (masked) p.phase = p.m1 > thr ? p.phase : 0.f;
        a.ati_phase[i] = p.phase; a.mag1[i] = p.m1; a.dpca_mag[i] = p.dm;
        if (OPT) {
            if (a.interf) a.interf[i] = p.interf;
            if (a.diff) a.diff[i] = p.diff;
            if (a.mag2) a.mag2[i] = p.m2;
            if (a.ph1) a.ph1[i] = p.p1;
            if (a.ph2) a.ph2[i] = p.p2;
            if (a.dpca_phase) a.dpca_phase[i] = p.dp;
        }
        vmax = fmaxf(vmax, p.m1); sre += p.sre; sim += p.sim;
    }
    // wave (64 lanes) then block reduction
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
        sre += __shfl_down(sre, off, 64);
        sim += __shfl_down(sim, off, 64);
    }
    __shared__ float smax[ATI_THREADS / 64];
    __shared__ double sr[ATI_THREADS / 64], si[ATI_THREADS / 64];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smax[w] = vmax; sr[w] = sre; si[w] = sim; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < ATI_THREADS / 64; ++k) { vmax = fmaxf(vmax, smax[k]); sre += sr[k]; sim += si[k]; }
        a.part_max[blockIdx.x] = vmax;
        a.part_sum[blockIdx.x] = make_double2(sre, sim);
    }
}

hipError_t launch_ati_dpca(const AtiArgs& a, hipStream_t st) {
    const int blocks = ati_blocks(a.n);
    const bool opt = a.interf || a.diff || a.mag2 || a.ph1 || a.ph2 || a.dpca_phase;
    if (opt) hipLaunchKernelGGL(ati_dpca_kernel<true>, dim3(blocks), dim3(ATI_THREADS), 0, st, a);
    else hipLaunchKernelGGL(ati_dpca_kernel<false>, dim3(blocks), dim3(ATI_THREADS), 0, st, a);
    return hipGetLastError();
}

// fixed-order final reduction (bitwise reproducible): out3 = {max, sum_re, sum_im}
__global__ void ati_finish_kernel(const float* part_max, const double2* part_sum, int blocks, double* out3) {
    __shared__ float smax[256];
    __shared__ double sr[256], si[256];
    float m = 0.f;
    double re = 0.0, im = 0.0;
    for (int i = threadIdx.x; i < blocks; i += 256) {
        m = fmaxf(m, part_max[i]);
        re += part_sum[i].x;
        im += part_sum[i].y;
    }
    smax[threadIdx.x] = m; sr[threadIdx.x] = re; si[threadIdx.x] = im;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
            sr[threadIdx.x] += sr[threadIdx.x + s];
            si[threadIdx.x] += si[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out3[0] = (double)smax[0]; out3[1] = sr[0]; out3[2] = si[0]; }
}
// the fused form (az_tile_kernel, AZ_EPI_SCALE_ATI): partial sums only, the maximum comes from the focus's shards.  Two fixed-order
// levels (a single workgroup walking 65 536 partials took 0.10 ms): FIN_BLOCKS workgroups reduce one contiguous chunk each, the last
// level adds their results in index order; `scratch` holds FIN_BLOCKS double2.
static constexpr int FIN_BLOCKS = 128;
__device__ __forceinline__ void block_sum(double& re, double& im, double* sr, double* si) {
    sr[threadIdx.x] = re; si[threadIdx.x] = im;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sr[threadIdx.x] += sr[threadIdx.x + s]; si[threadIdx.x] += si[threadIdx.x + s]; }
        __syncthreads();
    }
    re = sr[0]; im = si[0];
}
__global__ __launch_bounds__(256) void ati_finish_sums1_kernel(const double2* part_sum, int n, double2* scratch) {
    __shared__ double sr[256], si[256];
    const int chunk = (n + FIN_BLOCKS - 1) / FIN_BLOCKS, lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    double re = 0.0, im = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) { re += part_sum[i].x; im += part_sum[i].y; }
    block_sum(re, im, sr, si);
    if (threadIdx.x == 0) scratch[blockIdx.x] = make_double2(re, im);
}
__global__ __launch_bounds__(256) void ati_finish_sums2_kernel(const double2* scratch, const float* max_shards, double* out3) {
    __shared__ float smax[256];
    __shared__ double sr[256], si[256];
    double re = 0.0, im = 0.0;
    if (threadIdx.x < FIN_BLOCKS) { re = scratch[threadIdx.x].x; im = scratch[threadIdx.x].y; }
    float m = 0.f;
    for (unsigned k = threadIdx.x; k < MAX_SHARDS; k += 256) m = fmaxf(m, max_shards[32 * k]);
    smax[threadIdx.x] = m;
    block_sum(re, im, sr, si);
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) { out3[0] = (double)smax[0]; out3[1] = re; out3[2] = im; }
}
hipError_t launch_ati_finish_sums(const double2* part_sum, int n, const float* max_shards, double2* scratch, double* out3, hipStream_t st) {
    hipLaunchKernelGGL(ati_finish_sums1_kernel, dim3(FIN_BLOCKS), dim3(256), 0, st, part_sum, n, scratch);
    hipLaunchKernelGGL(ati_finish_sums2_kernel, dim3(1), dim3(256), 0, st, (const double2*)scratch, max_shards, out3);
    return hipGetLastError();
}
hipError_t launch_ati_finish(const float* part_max, const double2* part_sum, int blocks, double* out3, hipStream_t st) {
    hipLaunchKernelGGL(ati_finish_kernel, dim3(1), dim3(256), 0, st, part_max, part_sum, blocks, out3);
    return hipGetLastError();
}

// ati_phase[~(mag > thr)] = 0   (sar_ati_dcpa_sim_csa.py:447-449)
__global__ __launch_bounds__(256) void mask_phase_kernel(const float* phase, const float* mag, size_t n, float thr, float* out) {
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 p = reinterpret_cast<const float4*>(phase)[i];
        const float4 m = reinterpret_cast<const float4*>(mag)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(m.x > thr ? p.x : 0.f, m.y > thr ? p.y : 0.f,
                                                        m.z > thr ? p.z : 0.f, m.w > thr ? p.w : 0.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        out[i] = mag[i] > thr ? phase[i] : 0.f;
    }
}
// the same with thr = frac * max|slc1| taken from the ATI launch's device-side result (no host round trip)
__global__ __launch_bounds__(256) void mask_phase_frac_kernel(const float* phase, const float* mag, size_t n, float frac,
                                                              const double* out3, float* out) {
    const float thr = (float)out3[0] * frac;          // float32 product, as the host facade computes it
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 p = reinterpret_cast<const float4*>(phase)[i];
        const float4 m = reinterpret_cast<const float4*>(mag)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(m.x > thr ? p.x : 0.f, m.y > thr ? p.y : 0.f,
                                                        m.z > thr ? p.z : 0.f, m.w > thr ? p.w : 0.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        out[i] = mag[i] > thr ? phase[i] : 0.f;
    }
}
hipError_t launch_mask_phase_frac(const float* phase, const float* mag, size_t n, float frac, const double* out3, float* out,
                                  hipStream_t st) {
    size_t b = (n / 4 + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(mask_phase_frac_kernel, dim3((unsigned)b), dim3(256), 0, st, phase, mag, n, frac, out3, out);
    return hipGetLastError();
}
hipError_t launch_mask_phase(const float* phase, const float* mag, size_t n, float thr, float* out, hipStream_t st) {
    size_t b = (n / 4 + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(mask_phase_kernel, dim3((unsigned)b), dim3(256), 0, st, phase, mag, n, thr, out);
    return hipGetLastError();
}

// |x| of a complex64 buffer (full-resolution magnitude stack slot), 16 B per lane both ways where n allows
__global__ __launch_bounds__(256) void magnitude_kernel(const cf* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(in)[2 * i], b = reinterpret_cast<const float4*>(in)[2 * i + 1];
        reinterpret_cast<float4*>(out)[i] = make_float4(hypotf(a.x, a.y), hypotf(a.z, a.w), hypotf(b.x, b.y), hypotf(b.z, b.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        out[i] = hypotf(in[i].x, in[i].y);
    }
}
hipError_t launch_magnitude(const cf* in, float* out, size_t n, hipStream_t st) {
    size_t b = (n / 4 + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(magnitude_kernel, dim3((unsigned)b), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}

// max |x| of an fp32 buffer folded into *out (non-negative floats order like their bit patterns, so one atomicMax on the
// bits per wave): the per-rank half of the global display normalisation of a frame stack (sar_batch_sim.py:337-338)
// The maximum is taken on the bit patterns (|x| has the sign bit clear, so the patterns order like the values and every NaN
// orders above infinity): a NaN in the frame comes out as a NaN, as np.max(np.abs(fr)) does - fmaxf would drop it.
// x may sit at any 4-byte alignment (a stack slot of odd pixel count): a scalar head up to the first 16-byte boundary.
__global__ __launch_bounds__(256) void max_abs_f32_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ out) {
    auto bits = [](float v) { return __float_as_uint(fabsf(v)); };
    size_t head = ((16 - ((uintptr_t)x & 15)) & 15) / 4;          // floats before the first 16-byte boundary
    if (head > n) head = n;
    const float* __restrict__ xa = x + head;
    const size_t na = n - head, n4 = na / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned m = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(xa)[i];
        m = max(max(m, max(bits(a.x), bits(a.y))), max(bits(a.z), bits(a.w)));
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) m = max(m, bits(x[threadIdx.x]));
        if (threadIdx.x < (na & 3)) m = max(m, bits(xa[n4 * 4 + threadIdx.x]));
    }
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0u) atomicMax(out, m);
}
hipError_t launch_max_abs_f32(const float* x, size_t n, float* out, hipStream_t st) {
    size_t b = (n / 4 + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(max_abs_f32_kernel, dim3((unsigned)b), dim3(256), 0, st, x, n, reinterpret_cast<unsigned*>(out));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------
// corner turn: out[c][r] = in[r][c].  64x64 complex64 tiles staged through a
// padded LDS image; both the read and the write are 512-byte row segments.
// A thread's sixteen loads are issued before the first of them is used: written as `if (inside) tile[..] = in[..]` every
// load was waited for inside its own predicated block before the next was issued (tools/isa_load_waits.py: 16 loads, 16
// waits to zero).  16384^2: 0.986 -> 0.864 ms = 4.97 TB/s = 0.62 of 8 TB/s; 32768 x 2048: 0.246 -> 0.197 ms; nontemporal
// accesses, which help the azimuth tiles, cost 4 % here (profiles/r05_bf_corner_turn.log).
// ------------------------------------------------------------------------------
static constexpr int CT = 64;
__global__ __launch_bounds__(256) void corner_turn_kernel(const cf* __restrict__ in, cf* __restrict__ out, int rows, int cols) {
    __shared__ cf tile[CT][CT + 1];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 x 4
    const int r0 = blockIdx.y * CT, c0 = blockIdx.x * CT;
    cf v[CT / 4];
#pragma unroll
    for (int k = 0; k < CT; k += 4) {
        const int r = r0 + ty + k, c = c0 + tx;
        v[k / 4] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < CT; k += 4) tile[ty + k][tx] = v[k / 4];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CT; k += 4) {
        const int c = c0 + ty + k, r = r0 + tx;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][ty + k];
    }
}
hipError_t launch_corner_turn(const cf* in, cf* out, int rows, int cols, hipStream_t st) {
    dim3 grid((cols + CT - 1) / CT, (rows + CT - 1) / CT);
    hipLaunchKernelGGL(corner_turn_kernel, grid, dim3(256), 0, st, in, out, rows, cols);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------
// multilook: out[R/L x C/L] = mean_{LxL} |in|^2.  One workgroup per output row
// and 512-column chunk; reads are 16 B per lane along the contiguous dimension.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void multilook_kernel(const cf* __restrict__ in, float* __restrict__ out, int rows, int cols, int L) {
    __shared__ float part[512];
    const int orow = blockIdx.y;
    const int c0 = blockIdx.x * 512 + threadIdx.x * 2;
    float a0 = 0.f, a1 = 0.f;
    if (c0 + 1 < cols) {
        for (int l = 0; l < L; ++l) {
            const float4 x = *reinterpret_cast<const float4*>(in + (size_t)(orow * L + l) * cols + c0);
            a0 += x.x * x.x + x.y * x.y;
            a1 += x.z * x.z + x.w * x.w;
        }
    }
    part[threadIdx.x * 2] = a0;
    part[threadIdx.x * 2 + 1] = a1;
    __syncthreads();
    const int nout = 512 / L;
    const int ocols = cols / L;
    for (int o = threadIdx.x; o < nout; o += 256) {
        const int oc = blockIdx.x * nout + o;
        if (oc < ocols) {
            float s = 0.f;
            for (int l = 0; l < L; ++l) s += part[o * L + l];
            out[(size_t)orow * ocols + oc] = s / (float)(L * L);
        }
    }
}
hipError_t launch_multilook(const cf* in, float* out, int rows, int cols, int looks, hipStream_t st) {
    dim3 grid((cols + 511) / 512, rows / looks);
    hipLaunchKernelGGL(multilook_kernel, grid, dim3(256), 0, st, in, out, rows, cols, looks);
    return hipGetLastError();
}

// second half of the multilook fused into the focus (AZ_EPI_SCALE_LOOK wrote row-wise sums over `looks` columns): sum the
// `looks` rows of each block in a fixed order and divide by looks^2.  Reads n_az x n_rg/looks floats (64 MiB at 16384^2 / 16
// looks) instead of the 2 GiB image.
__global__ __launch_bounds__(256) void look_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int out_rows,
                                                          int cols, int L) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= cols || r >= out_rows) return;
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += part[(size_t)(r * L + l) * cols + c];
    out[(size_t)r * cols + c] = s / (float)(L * L);
}
hipError_t launch_look_finish(const float* part, float* out, int out_rows, int cols, int looks, hipStream_t st) {
    dim3 grid((cols + 255) / 256, out_rows);
    hipLaunchKernelGGL(look_finish_kernel, grid, dim3(256), 0, st, part, out, out_rows, cols, looks);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------
// counter-based complex Gaussian noise: sample i depends only on (seed, i)
// ------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void fill_noise_kernel(cf* buf, size_t n, uint64_t seed) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint64_t h = mix64(seed * 0xD1342543DE82EF95ull + i);
        const float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);      // (0,1)
        const float u2 = ((float)(uint32_t)((h >> 8) & 0xFFFFFF)) * (1.0f / 16777216.0f); // [0,1)
        const float r = sqrtf(-2.0f * __logf(u1));
        float s, c;
        __sincosf(6.28318530718f * u2, &s, &c);
        buf[i] = make_float2(r * c, r * s);
    }
}
hipError_t launch_fill_noise(cf* buf, size_t n, uint64_t seed, hipStream_t st) {
    size_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(fill_noise_kernel, dim3((unsigned)b), dim3(256), 0, st, buf, n, seed);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------
// thermal noise + K-distributed sea clutter added in place (add_ocean_noise, sar_satellite_sim.py:331-344;
// generate_noise_tensor, sar_batch_sim.py:66-82):
//   x[i] += sigma (N1 + j N2) + sqrt(Pc * G * E) exp(j 2 pi U),  G ~ Gamma(shape nu, mean 1), E ~ Exp(1)
// Counter-based: sample i depends only on (seed, i); the Gamma draw is Marsaglia-Tsang with its own counter
// stream (shape < 1 boosted through Gamma(nu + 1) U^(1/nu)).
// ------------------------------------------------------------------------------
struct Rng {
    uint64_t key, ctr;
    __device__ __forceinline__ uint64_t next() { return mix64(key + (ctr++) * 0x9E3779B97F4A7C15ull); }
    __device__ __forceinline__ float uniform() {                       // (0,1)
        return ((float)(uint32_t)(next() >> 40) + 0.5f) * (1.0f / 16777216.0f);
    }
    __device__ __forceinline__ float2 normal2() {
        const uint64_t h = next();
        const float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(uint32_t)((h >> 8) & 0xFFFFFF)) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * __logf(u1));
        return make_float2(r * __builtin_amdgcn_cosf(u2), r * __builtin_amdgcn_sinf(u2));     // v_sin / v_cos take revolutions
    }
};
__device__ float gamma_unit_mean(Rng& g, float nu) {
    const float k = nu < 1.0f ? nu + 1.0f : nu;
    const float d = k - 1.0f / 3.0f, c = rsqrtf(9.0f * d);
    float out = d;
    for (int it = 0; it < 64; ++it) {                                  // acceptance > 95 % per round
        const float x = g.normal2().x;
        float v = 1.0f + c * x;
        if (v <= 0.f) continue;
        v = v * v * v;
        const float u = g.uniform();
        if (__logf(u) < 0.5f * x * x + d - d * v + d * __logf(v)) { out = d * v; break; }
    }
    if (nu < 1.0f) out *= __powf(g.uniform(), 1.0f / nu);
    return out / nu;                                                   // scale 1/nu: unit mean
}
// levels: optional {sigma, clutter_power} on the device (noise_levels_kernel), so that the noise can follow the power reduction
// without a host round trip; the host-parameter form passes nullptr
__global__ __launch_bounds__(256) void ocean_noise_kernel(cf* buf, size_t n, float sigma, float clutter_power, float nu,
                                                          uint64_t seed, const float* __restrict__ levels) {
    if (levels) { sigma = levels[0]; clutter_power = levels[1]; }
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        Rng g{mix64(seed * 0xD1342543DE82EF95ull + i), 0};
        const float2 th = g.normal2();
        cf x = buf[i];
        x.x += sigma * th.x;
        x.y += sigma * th.y;
        if (clutter_power > 0.f) {
            float tex, spk;
            if (nu == 1.0f) {
                // shape 1 - K_NU of every script of the reference (sar_satellite_sim.py:317, sar_vehicle_sim.py:138, sar_batch_sim.py:49):
                // Gamma(1, 1) IS Exp(1), so texture and speckle are two logarithms of one hash's two 24-bit halves.  The rejection
                // loop below runs until the LAST lane of a wave accepts (all 64 in the first round: 0.95^64 = 4 %), two to three
                // rounds of a normal pair, a uniform and two logarithms: 0.54 -> 0.24 ms per 2500 x 22004 VideoSAR frame.
                const uint64_t h = g.next();
                tex = -__logf(((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f));
                spk = -__logf(((float)(uint32_t)((h >> 16) & 0xFFFFFF) + 0.5f) * (1.0f / 16777216.0f));
            } else {
                tex = gamma_unit_mean(g, nu);
                spk = -__logf(g.uniform());
            }
            const float amp = sqrtf(clutter_power * tex * spk);
            const float u = g.uniform();
            x.x += amp * __builtin_amdgcn_cosf(u);
            x.y += amp * __builtin_amdgcn_sinf(u);
        }
        buf[i] = x;
    }
}
hipError_t launch_ocean_noise(cf* buf, size_t n, float sigma, float clutter_power, float nu, uint64_t seed, hipStream_t st,
                              const float* levels) {
    size_t b = (n + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(ocean_noise_kernel, dim3((unsigned)b), dim3(256), 0, st, buf, n, sigma, clutter_power, nu, seed, levels);
    return hipGetLastError();
}

// max and sum of |x|^2 (signal power for the noise level: sar_satellite_sim.py:333, sar_batch_sim.py:316);
// per-block partials in fp64, finished on the host in block order
__global__ __launch_bounds__(256) void power_stats_kernel(const cf* buf, size_t n, double* part) {
    __shared__ double s_sum[256];
    __shared__ float s_max[256];
    const size_t stride = (size_t)gridDim.x * 256;
    double sum = 0.0;
    float mx = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const cf x = buf[i];
        const float p = x.x * x.x + x.y * x.y;
        sum += (double)p;
        mx = fmaxf(mx, p);
    }
    s_sum[threadIdx.x] = sum;
    s_max[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + o];
            s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s_sum[0]; part[2 * blockIdx.x + 1] = (double)s_max[0]; }
}
hipError_t launch_power_stats(const cf* buf, size_t n, double* part, int blocks, hipStream_t st) {
    hipLaunchKernelGGL(power_stats_kernel, dim3(blocks), dim3(256), 0, st, buf, n, part);
    return hipGetLastError();
}
// The host's finish of the partials and its noise levels, on the device and in the same order and arithmetic: reference power =
// max |x|^2 (sar_batch_sim.py:313) or mean |x|^2 (sar_satellite_sim.py:333); sigma = sqrt(ref / snr_lin / 2), clutter = ref / scr_lin
// (scr_lin = 0: thermal noise only) - sar_batch_sim.py:67-78, sar_satellite_sim.py:334-343.
// One wave: the partials are fetched by all 64 lanes at once into LDS, then lane 0 adds them in block order - the host's order, so the
// levels stay bit-identical to the two-call form.  With lane 0 fetching them itself every addition waited for its own load from L2:
// 104 us per frame for 1024 partials, on the critical path of every VideoSAR frame (profiles/r05_bo_videosar_trace_busy.log).
static constexpr int NOISE_LEVEL_PARTS = 1024;
__global__ __launch_bounds__(64) void noise_levels_kernel(const double* __restrict__ part, int blocks, double n, int ref_is_max,
                                                          double snr_lin, double scr_lin, float* __restrict__ levels) {
    __shared__ double s_part[2 * NOISE_LEVEL_PARTS];
    double sum = 0.0, mx = 0.0;
    for (int b0 = 0; b0 < blocks; b0 += NOISE_LEVEL_PARTS) {
        const int nb = min(NOISE_LEVEL_PARTS, blocks - b0);
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * nb; i += 64) s_part[i] = part[2 * b0 + i];
        __syncthreads();
        if (threadIdx.x == 0)
            for (int b = 0; b < nb; ++b) { sum += s_part[2 * b]; if (s_part[2 * b + 1] > mx) mx = s_part[2 * b + 1]; }
    }
    if (threadIdx.x != 0) return;
    const double ref = ref_is_max ? mx : sum / n;
    levels[0] = (float)sqrt(ref / snr_lin / 2.0);
    levels[1] = scr_lin > 0.0 ? (float)(ref / scr_lin) : 0.f;
}
hipError_t launch_noise_levels(const double* part, int blocks, size_t n, int ref_is_max, double snr_lin, double scr_lin, float* levels,
                               hipStream_t st) {
    hipLaunchKernelGGL(noise_levels_kernel, dim3(1), dim3(64), 0, st, part, blocks, (double)n, ref_is_max, snr_lin, scr_lin, levels);
    return hipGetLastError();
}

// a launch that occupies `gridDim.x` CUs' worth of one small workgroup each for about `cycles` shader cycles (lane concurrency probe)
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long cycles, unsigned* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    unsigned x = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) x += 1;
    if (x == 0xffffffffu) *sink = x;
}
hipError_t launch_spin(int blocks, unsigned long long cycles, unsigned* sink, hipStream_t st) {
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), 0, st, cycles, sink);
    return hipGetLastError();
}

}  // namespace sarx
