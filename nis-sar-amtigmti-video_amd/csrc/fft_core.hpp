// sarx FFT core for gfx950 (MI355X): in-register radix butterflies and a
// register-resident Stockham driver whose inter-stage transposes go through LDS.
//
// Data model: one FFT of length N is owned by T = N/P threads (P points per
// thread, held in VGPRs).  A stage of radix R does P/R butterflies per thread;
// butterfly j reads points j + r*N/R and writes (j/Ns)*Ns*R + j%Ns + r*Ns
// (Stockham autosort, output in natural order).  Between stages the points
// are exchanged through an LDS image; the first stage's inputs come straight
// from global memory and the last stage's outputs go straight back, both
// coalesced, so a pass touches HBM exactly once per sample in each direction.
//
// W > 1 interleaves W independent FFTs ("columns") with the column index
// fastest in LDS and across lanes; that is the azimuth (corner-turn-free)
// form: a [R rows x W cols] tile, FFT along rows.
#pragma once
#include <hip/hip_runtime.h>

namespace sarx {

typedef float2 cf;

__device__ __forceinline__ cf cmul(cf a, cf b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cconj(cf a) { return make_float2(a.x, -a.y); }
// multiply by -i (forward) or +i (inverse)
template <bool INV> __device__ __forceinline__ cf mul_mi(cf a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// multiply by exp(-+ i*pi/4), exp(-+ 3i*pi/4)
template <bool INV> __device__ __forceinline__ cf mul_w8_1(cf a) {
    const float h = 0.70710678118654752440f;
    return INV ? make_float2(h * (a.x - a.y), h * (a.x + a.y)) : make_float2(h * (a.x + a.y), h * (a.y - a.x));
}
template <bool INV> __device__ __forceinline__ cf mul_w8_3(cf a) {
    const float h = 0.70710678118654752440f;
    return INV ? make_float2(-h * (a.x + a.y), h * (a.x - a.y)) : make_float2(h * (a.y - a.x), -h * (a.x + a.y));
}

// ---- streaming image accesses ---------------------------------------------------------------------------------------
// Compile-time nontemporal variants (NT) for kernels whose register budget has no room for a run-time choice.
typedef float nt_f2 __attribute__((ext_vector_type(2)));
typedef float nt_f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 ld16(const cf* p) {
    if constexpr (NT) { const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    else return *reinterpret_cast<const float4*>(p);
}
template <bool NT> __device__ __forceinline__ void st16(cf* p, float4 x) {
    if constexpr (NT) __builtin_nontemporal_store(nt_f4{x.x, x.y, x.z, x.w}, reinterpret_cast<nt_f4*>(p));
    else *reinterpret_cast<float4*>(p) = x;
}
template <bool NT> __device__ __forceinline__ cf ld8(const cf* p) {
    if constexpr (NT) { const nt_f2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f2*>(p)); return make_float2(v.x, v.y); }
    else return *p;
}
template <bool NT> __device__ __forceinline__ void st8(cf* p, cf x) {
    if constexpr (NT) __builtin_nontemporal_store(nt_f2{x.x, x.y}, reinterpret_cast<nt_f2*>(p));
    else *p = x;
}

// ---- in-register DFTs, natural order in and out; stride S between elements ----
template <bool INV> __device__ __forceinline__ void dft2(cf& a, cf& b) {
    cf t = a;
    a = cadd(t, b);
    b = csub(t, b);
}
template <bool INV> __device__ __forceinline__ void dft4(cf& v0, cf& v1, cf& v2, cf& v3) {
    cf a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), a3 = mul_mi<INV>(csub(v1, v3));
    v0 = cadd(a0, a2);
    v1 = cadd(a1, a3);
    v2 = csub(a0, a2);
    v3 = csub(a1, a3);
}
template <bool INV> __device__ __forceinline__ void dft8(cf* v) {
    // n = n1 + 2*n2, k = 4*k1 + k2: four 2-pt DFTs, twiddle, two 4-pt DFTs
    cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4<INV>(e0, e1, e2, e3);
    dft4<INV>(o0, o1, o2, o3);
    o1 = mul_w8_1<INV>(o1);
    o2 = mul_mi<INV>(o2);
    o3 = mul_w8_3<INV>(o3);
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}
template <bool INV> __device__ __forceinline__ void dft16(cf* v) {
    // n = n1 + 4*n2, k = 4*k1 + k2
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    const float sg = INV ? 1.0f : -1.0f;
    cf y[4][4];
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) {
        cf a = v[n1], b = v[n1 + 4], c = v[n1 + 8], d = v[n1 + 12];
        dft4<INV>(a, b, c, d);
        y[n1][0] = a; y[n1][1] = b; y[n1][2] = c; y[n1][3] = d;
    }
    // y[n1][k2] *= W16^(n1*k2)
    const cf w1 = make_float2(c1, sg * s1), w2 = make_float2(h, sg * h), w3 = make_float2(s1, sg * c1);
    const cf w6 = make_float2(-h, sg * h), w9 = make_float2(-c1, -sg * s1);
    y[1][1] = cmul(y[1][1], w1); y[1][2] = cmul(y[1][2], w2); y[1][3] = cmul(y[1][3], w3);
    y[2][1] = cmul(y[2][1], w2); y[2][2] = mul_mi<INV>(y[2][2]); y[2][3] = cmul(y[2][3], w6);
    y[3][1] = cmul(y[3][1], w3); y[3][2] = cmul(y[3][2], w6); y[3][3] = cmul(y[3][3], w9);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        cf a = y[0][k2], b = y[1][k2], c = y[2][k2], d = y[3][k2];
        dft4<INV>(a, b, c, d);
        v[k2] = a; v[4 + k2] = b; v[8 + k2] = c; v[12 + k2] = d;
    }
}
// 32 points: two 16-point DFTs of the even and odd samples, then X[k] = E[k] +- W32^k O[k]
template <bool INV> __device__ __forceinline__ void dft32(cf* v) {
    cf e[16], o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    dft16<INV>(e);
    dft16<INV>(o);
    // cos, sin of pi*k/16, k = 0..15
    constexpr float C32[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                               0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                               0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f,
                               -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                               -0.92387953251128675613f, -0.98078528040323044913f};
    constexpr float S32[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                               0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f,
                               0.98078528040323044913f, 1.0f, 0.98078528040323044913f, 0.92387953251128675613f,
                               0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                               0.38268343236508977173f, 0.19509032201612826785f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        cf t;
        if (k == 0) t = o[0];
        else if (k == 8) t = mul_mi<INV>(o[8]);
        else t = cmul(o[k], make_float2(C32[k], INV ? S32[k] : -S32[k]));
        v[k] = cadd(e[k], t);
        v[k + 16] = csub(e[k], t);
    }
}
// v[r] *= w^r, r = 1..31; w^r = w^(r/2) * w^(r - r/2): depth 5
__device__ __forceinline__ void apply_twiddle_powers32(cf* v, cf w1) {
    cf w[32];
    w[1] = w1;
#pragma unroll
    for (int r = 2; r < 32; ++r) w[r] = cmul(w[r / 2], w[r - r / 2]);
#pragma unroll
    for (int r = 1; r < 32; ++r) v[r] = cmul(v[r], w[r]);
}

template <int R, bool INV> __device__ __forceinline__ void dft(cf* v) {
    if constexpr (R == 2) dft2<INV>(v[0], v[1]);
    else if constexpr (R == 4) dft4<INV>(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 8) dft8<INV>(v);
    else if constexpr (R == 16) dft16<INV>(v);
    else static_assert(R == 2, "unsupported radix");
}

// powers of one twiddle: v[r] *= w^r, r = 1..R-1 (multiplication depth <= 4)
template <int R> __device__ __forceinline__ void apply_twiddle_powers(cf* v, cf w1) {
    if constexpr (R >= 2) v[1] = cmul(v[1], w1);
    if constexpr (R >= 4) {
        cf w2 = cmul(w1, w1), w3 = cmul(w2, w1);
        v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3);
        if constexpr (R >= 8) {
            cf w4 = cmul(w2, w2), w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3);
            v[4] = cmul(v[4], w4); v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], w7);
            if constexpr (R >= 16) {
                cf w8 = cmul(w4, w4);
                v[8] = cmul(v[8], w8);
                v[9] = cmul(v[9], cmul(w8, w1));
                v[10] = cmul(v[10], cmul(w5, w5));
                v[11] = cmul(v[11], cmul(w8, w3));
                v[12] = cmul(v[12], cmul(w6, w6));
                v[13] = cmul(v[13], cmul(w8, w5));
                v[14] = cmul(v[14], cmul(w7, w7));
                v[15] = cmul(v[15], cmul(w8, w7));
            }
        }
    }
}

// ---- radix plans ---------------------------------------------------------------
// Plan<N>: as many radix-16 stages as fit, then one remainder stage (2, 4 or 8).
// REV = true puts the remainder stage first (used for the inverse half of the
// fused range pass so that it can start from the forward half's registers).
template <int N> struct Plan {
    static constexpr int log2n() { int l = 0; for (int n = N; n > 1; n >>= 1) ++l; return l; }
    static constexpr int L = log2n();
    static constexpr int n16 = L / 4;
    static constexpr int rem = 1 << (L % 4);
    static constexpr int nstages = n16 + (rem > 1 ? 1 : 0);
    static constexpr int P = (N >= 16) ? 16 : N;   // points per thread
    static constexpr int T = N / P;                // threads per transform
    template <bool REV> static constexpr int radix(int s) {
        if (rem == 1) return 16;
        if (REV) return s == 0 ? rem : 16;
        return s == nstages - 1 ? rem : 16;
    }
    template <bool REV> static constexpr int ns_before(int s) {   // product of earlier radices
        int p = 1;
        for (int i = 0; i < s; ++i) p *= radix<REV>(i);
        return p;
    }
};

// LDS index of point `idx` of transform-local storage.  W == 1 (range lines):
// one pad element per 16 so the stride-R writes of the first exchange hit
// distinct banks.  W > 1 (azimuth tiles): column index fastest, no pad needed.
template <int W> __device__ __forceinline__ int lds_index(int idx, int c) {
    if constexpr (W == 1) return idx + (idx >> 4);
    else return idx * W + c;
}
template <int N, int W> struct LdsSize {
    static constexpr int value = (W == 1) ? (N + (N >> 4)) : N * W;   // in cf elements
};

// Base twiddle of a stage: exp(-+ 2 pi i k / D), D = Ns*R a power of two, k < Ns.
//  SARX_HW_TWIDDLE=1: v_sin_f32 / v_cos_f32 on the exact fp32 value k/D (in revolutions).  No
//    memory operation, so the only VMEM traffic of a range kernel is the line itself and a
//    prefetched next line can stay in flight (vmcnt retires in order: a table load issued after
//    the prefetch would have to wait for it).
//  SARX_HW_TWIDDLE=0: fp64-evaluated table tw[m] = exp(-2 pi i m / N).
#ifndef SARX_HW_TWIDDLE
#define SARX_HW_TWIDDLE 1
#endif
__device__ __forceinline__ cf cis_frac(float x);
template <int N, int D, bool INV> __device__ __forceinline__ cf stage_twiddle(int k, const cf* __restrict__ tw) {
#if SARX_HW_TWIDDLE
    const float x = (float)k * (1.0f / (float)D);          // exact: D is a power of two, k < 2^24
    return cis_frac(INV ? x : -x);
#else
    cf w = tw[k * (N / D)];
    return INV ? cconj(w) : w;
#endif
}

// One Stockham stage on registers.  v holds P points: butterfly b (j = t + b*T)
// occupies v[b*R .. b*R+R-1].
template <int N, int T, int R, int NS, bool INV>
__device__ __forceinline__ void stage_compute(cf* v, int t, const cf* __restrict__ tw) {
    constexpr int P = N / T;
    constexpr int B = P / R;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        if constexpr (NS > 1) {
            const int j = t + b * T;
            apply_twiddle_powers<R>(v + b * R, stage_twiddle<N, NS * R, INV>(j % NS, tw));
        }
        dft<R, INV>(v + b * R);
    }
}

// write stage outputs to the LDS image
template <int N, int T, int R, int NS, int W>
__device__ __forceinline__ void stage_scatter(const cf* v, int t, int c, cf* lds) {
    constexpr int P = N / T;
    constexpr int B = P / R;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int j = t + b * T;
        const int base = (j / NS) * (NS * R) + (j % NS);
#pragma unroll
        for (int r = 0; r < R; ++r) lds[lds_index<W>(base + r * NS, c)] = v[b * R + r];
    }
}
// read next stage's inputs from the LDS image
template <int N, int T, int R, int W>
__device__ __forceinline__ void stage_gather(cf* v, int t, int c, const cf* lds) {
    constexpr int P = N / T;
    constexpr int B = P / R;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int j = t + b * T;
#pragma unroll
        for (int r = 0; r < R; ++r) v[b * R + r] = lds[lds_index<W>(j + r * (N / R), c)];
    }
}

// Runs stages S..nstages-1 of Plan<N> (order REV) on registers that already
// hold stage S's inputs.  Leaves the last stage's outputs in v: butterfly b,
// point r is output index  (t + b*T) + r*(N/Rlast).
//
// WAVE_LOCAL: the whole transform lives in one wavefront (T == 64) and `lds` is
// a region no other wave touches.  LDS instructions of one wave execute in
// issue order, so the exchanges need no s_barrier, only a compiler fence; the
// waves of a workgroup then drift apart and overlap each other's LDS and VALU phases.
template <bool WAVE_LOCAL> __device__ __forceinline__ void exchange_sync() {
    if constexpr (WAVE_LOCAL) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}
template <int N, int W, bool INV, bool REV, int S = 0, bool WAVE_LOCAL = false>
__device__ __forceinline__ void stockham_run(cf* v, int t, int c, cf* lds, const cf* __restrict__ tw) {
    using PL = Plan<N>;
    static_assert(!WAVE_LOCAL || PL::T == 64, "wave-local transforms span exactly one wavefront");
    constexpr int R = PL::template radix<REV>(S);
    constexpr int NS = PL::template ns_before<REV>(S);
    stage_compute<N, PL::T, R, NS, INV>(v, t, tw);
    if constexpr (S + 1 < PL::nstages) {
        constexpr int R2 = PL::template radix<REV>(S + 1);
        if constexpr (S > 0) exchange_sync<WAVE_LOCAL>();   // previous gather done before overwrite
        stage_scatter<N, PL::T, R, NS, W>(v, t, c, lds);
        exchange_sync<WAVE_LOCAL>();
        stage_gather<N, PL::T, R2, W>(v, t, c, lds);
        stockham_run<N, W, INV, REV, S + 1, WAVE_LOCAL>(v, t, c, lds, tw);
    }
}

// index helpers for the first-stage load / last-stage store patterns
template <int N, bool REV> struct Edge {
    using PL = Plan<N>;
    static constexpr int R_first = PL::template radix<REV>(0);
    static constexpr int R_last = PL::template radix<REV>(PL::nstages - 1);
    // point (b, r) of the first stage reads input index:
    __device__ static __forceinline__ int in_index(int t, int b, int r) { return t + b * PL::T + r * (N / R_first); }
    // point (b, r) of the last stage is output index:
    __device__ static __forceinline__ int out_index(int t, int b, int r) { return t + b * PL::T + r * (N / R_last); }
};

// ---- phase helpers ---------------------------------------------------------------
// exp(2*pi*i*f) for f already reduced to [-0.5, 0.5] revolutions.
//  SARX_HW_SINCOS=1: v_sin_f32 / v_cos_f32 (argument in revolutions, quarter rate)
//  SARX_HW_SINCOS=0: quadrant split + Taylor polynomials on |x| <= 1/8 (truncation < 2e-9)
#ifndef SARX_HW_SINCOS
#define SARX_HW_SINCOS 1
#endif
__device__ __forceinline__ cf cis_frac(float x) {
#if SARX_HW_SINCOS
    return make_float2(__builtin_amdgcn_cosf(x), __builtin_amdgcn_sinf(x));
#else
    const float q = rintf(4.0f * x);         // -2..2
    const float r = fmaf(q, -0.25f, x);      // [-1/8, 1/8]
    const float th = 6.28318530717958647692f * r;
    const float t2 = th * th;
    float s = fmaf(t2, 2.7557319e-6f, -1.9841270e-4f);
    s = fmaf(s, t2, 8.3333333e-3f);
    s = fmaf(s, t2, -1.6666667e-1f);
    s = fmaf(s * t2, th, th);
    float c = fmaf(t2, 2.4801587e-5f, -1.3888889e-3f);
    c = fmaf(c, t2, 4.1666667e-2f);
    c = fmaf(c, t2, -0.5f);
    c = fmaf(c, t2, 1.0f);
    const int qi = (int)q & 3;               // rotate by qi quarter turns
    const float cs = (qi & 1) ? -s : c;
    const float sn = (qi & 1) ? c : s;
    return (qi & 2) ? make_float2(-cs, -sn) : make_float2(cs, sn);
#endif
}
// exp(2*pi*i*p) for a phase p in revolutions, fp64 in (|p| reaches 3e7 in Phi_3), fp32 out.
__device__ __forceinline__ cf cis_rev(double p) { return cis_frac((float)(p - rint(p))); }

}  // namespace sarx
