// In-register DFTs of any small length for the mixed-radix line transforms (range_mixed.hip):
// the reference's native range extent is 13200 = 24 * 22 * 25 samples (sar_ati_dcpa_sim_csa.py:111).
//
//   odd primes (3, 5, 7, 11, 13): the symmetric form  X[k], X[p-k] = (x0 + sum_n c_nk a_n) -+ i (sum_n s_nk b_n),
//       a_n = x_n + x_{p-n}, b_n = x_n - x_{p-n}: (p-1)^2/2 real-by-complex multiply-adds, constants folded at compile time
//   composites: one Cooley-Tukey split R = Ra * Rb in registers (n = Rb n1 + n2, k = k1 + Ra k2) with constant twiddles;
//       the power-of-two factor (<= 16) goes last so its butterflies are the hand-written ones of fft_core.hpp
// Every index below is a compile-time constant after unrolling: the arrays live in VGPRs.
#pragma once
#include "fft_core.hpp"

namespace sarx {
namespace mix {

// ---- compile-time sine / cosine of 2 pi num / den (double, Taylor after folding into [0, pi/4]) -----------------------
constexpr double PI = 3.14159265358979323846264338327950288;
constexpr double taylor_sin(double x) {
    double term = x, sum = x;
    for (int k = 1; k < 12; ++k) { term *= -x * x / ((2.0 * k) * (2.0 * k + 1.0)); sum += term; }
    return sum;
}
constexpr double taylor_cos(double x) {
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 12; ++k) { term *= -x * x / ((2.0 * k - 1.0) * (2.0 * k)); sum += term; }
    return sum;
}
// cos and sin of 2 pi num/den via the octant of 8 num / den (exact integer reduction)
struct CS { double c, s; };
constexpr CS unit_root(long long num, long long den) {
    num %= den;
    if (num < 0) num += den;
    // angle = 2 pi num/den = (pi/4) * (8 num/den); octant o, remainder fraction f in [0,1)
    const long long e = 8 * num;
    const int o = (int)(e / den);
    const double f = (double)(e % den) / (double)den;          // in [0, 1)
    // fold: odd octants measure from the next multiple of pi/4 downwards
    const double x = (o & 1) ? (1.0 - f) * (PI / 4) : f * (PI / 4);
    const double sx = taylor_sin(x), cx = taylor_cos(x);
    switch (o) {
        case 0: return {cx, sx};
        case 1: return {sx, cx};
        case 2: return {-sx, cx};
        case 3: return {-cx, sx};
        case 4: return {-cx, -sx};
        case 5: return {-sx, -cx};
        case 6: return {sx, -cx};
        default: return {cx, -sx};
    }
}
// cos / sin of 2 pi m / R for m = 0..R-1 as a compile-time table; indexed with loop variables of fully unrolled
// loops, so every use folds to a literal constant
template <int R> struct Roots {
    float c[R], s[R];
    constexpr Roots() : c(), s() {
        for (int m = 0; m < R; ++m) { const CS v = unit_root(m, R); c[m] = (float)v.c; s[m] = (float)v.s; }
    }
};
template <int R> struct RootTab { static constexpr Roots<R> t = Roots<R>(); };

constexpr bool is_prime(int n) {
    if (n < 2) return false;
    for (int d = 2; d * d <= n; ++d)
        if (n % d == 0) return false;
    return true;
}
constexpr bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
constexpr int pow2_part(int r) { int p = 1; while (r % 2 == 0 && p < 16) { r /= 2; p *= 2; } return p; }
constexpr int smallest_factor(int r) { for (int d = 2; d <= r; ++d) if (r % d == 0) return d; return r; }
// R = Ra * Rb: the power-of-two part last, otherwise the smallest prime first
template <int R> struct Split {
    static constexpr int b = (pow2_part(R) > 1 && pow2_part(R) != R) ? pow2_part(R) : R / smallest_factor(R);
    static constexpr int a = R / b;
};

template <int R, bool INV> __device__ __forceinline__ void dft_any(cf* v);

// odd prime length
template <int P, bool INV> __device__ __forceinline__ void dft_prime(cf* v) {
    constexpr int H = (P - 1) / 2;
    constexpr Roots<P> tab = RootTab<P>::t;
    cf a[H + 1], b[H + 1];
#pragma unroll
    for (int n = 1; n <= H; ++n) { a[n] = cadd(v[n], v[P - n]); b[n] = csub(v[n], v[P - n]); }
    const cf x0 = v[0];
    cf s0 = x0;
#pragma unroll
    for (int n = 1; n <= H; ++n) s0 = cadd(s0, a[n]);
    v[0] = s0;
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        cf C = x0, S = make_float2(0.f, 0.f);
#pragma unroll
        for (int n = 1; n <= H; ++n) {
            const float c = tab.c[(n * k) % P], sn = tab.s[(n * k) % P];
            C.x = fmaf(c, a[n].x, C.x); C.y = fmaf(c, a[n].y, C.y);
            S.x = fmaf(sn, b[n].x, S.x); S.y = fmaf(sn, b[n].y, S.y);
        }
        // forward: X[k] = C - iS, X[P-k] = C + iS; inverse: the other way round
        const cf lo = make_float2(C.x + S.y, C.y - S.x), hi = make_float2(C.x - S.y, C.y + S.x);
        v[k] = INV ? hi : lo;
        v[P - k] = INV ? lo : hi;
    }
}

// multiply by the constant W_R^m (forward) or its conjugate (inverse); m is a loop variable of an unrolled loop, so the
// trivial cases fold away
template <int R, bool INV> __device__ __forceinline__ cf mul_root(cf x, int m) {
    constexpr Roots<R> tab = RootTab<R>::t;
    m %= R;
    if (m == 0) return x;
    if (4 * m == R) return mul_mi<INV>(x);                       // -i / +i
    if (2 * m == R) return make_float2(-x.x, -x.y);
    if (4 * m == 3 * R) return mul_mi<!INV>(x);
    const float c = tab.c[m], sn = INV ? tab.s[m] : -tab.s[m];   // exp(-+ 2 pi i m / R)
    return make_float2(fmaf(x.x, c, -x.y * sn), fmaf(x.x, sn, x.y * c));
}

template <int R, bool INV> __device__ __forceinline__ void dft_composite(cf* v) {
    constexpr int A = Split<R>::a, B = Split<R>::b;
    cf y[R];
    // step 1: A-point DFTs over n1 (elements B n1 + n2), for every n2;  y[n2 * A + k1] *= W_R^(n2 k1)
#pragma unroll
    for (int n2 = 0; n2 < B; ++n2) {
        cf t[A];
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1) t[n1] = v[B * n1 + n2];
        dft_any<A, INV>(t);
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) y[n2 * A + k1] = mul_root<R, INV>(t[k1], n2 * k1);
    }
    // step 3: B-point DFTs over n2 for every k1;  X[k1 + A k2]
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) {
        cf t[B];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) t[n2] = y[n2 * A + k1];
        dft_any<B, INV>(t);
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) v[k1 + A * k2] = t[k2];
    }
}

template <int R, bool INV> __device__ __forceinline__ void dft_any(cf* v) {
    if constexpr (R == 1) return;
    else if constexpr (R == 2 || R == 4 || R == 8 || R == 16) dft<R, INV>(v);
    else if constexpr (R == 32) dft32<INV>(v);
    else if constexpr (is_prime(R)) dft_prime<R, INV>(v);
    else dft_composite<R, INV>(v);
}

// v[r] *= w^r, r = 1..R-1, by a multiplication tree of depth ceil(log2 R); the upper half of the tree is consumed as it is
// produced, so at most R/2 + 1 powers are alive beside the R data registers
template <int R> __device__ __forceinline__ void apply_powers_progressive(cf* v, cf w1) {
    constexpr int H = (R + 1) / 2;
    cf w[H + 1];
    w[1] = w1;
#pragma unroll
    for (int r = 2; r <= H; ++r) w[r] = cmul(w[r / 2], w[r - r / 2]);
#pragma unroll
    for (int r = H + 1; r < R; ++r) v[r] = cmul(v[r], cmul(w[r / 2], w[r - r / 2]));
#pragma unroll
    for (int r = 1; r <= H; ++r) v[r] = cmul(v[r], w[r]);
}
template <int R> __device__ __forceinline__ void apply_powers(cf* v, cf w1) {
    if constexpr (R <= 25) {
        cf w[R];
        w[1] = w1;
#pragma unroll
        for (int r = 2; r < R; ++r) w[r] = cmul(w[r / 2], w[r - r / 2]);
#pragma unroll
        for (int r = 1; r < R; ++r) v[r] = cmul(v[r], w[r]);
    } else {
        apply_powers_progressive<R>(v, w1);       // long radices: the plain form keeps all R powers and spills under a 128-VGPR cap
    }
}

}  // namespace mix
}  // namespace sarx
