// Complex arithmetic on (re, im) register pairs with gfx950's packed fp32 instructions (v_pk_add_f32, v_pk_mul_f32,
// v_pk_fma_f32).  Every swizzle a butterfly needs - multiply by -+i, swap, broadcast of one half, negate one half - rides on
// the VOP3P operand modifiers (op_sel / op_sel_hi pick the half of each source that feeds the low / high result lane,
// neg_lo / neg_hi negate a source for one lane), so a complex add or subtract is one instruction, +-i times a number costs
// nothing, a complex multiply is two.  The compiler does not fold swap-and-negate into the modifiers on its own (it emits
// v_xor + v_mov), hence inline assembly for the primitives; everything above them is ordinary C++.
//
// Why: a wave issues a packed instruction in about the time of a scalar one when it is alone on its SIMD
// (tools/valubench.hip: 2.74 vs 2.53 ns at one wave per SIMD, 2.27 vs 1.30 at two, 2.05 vs 1.20 at four), so at the
// fused range kernel's two waves per SIMD the same arithmetic costs 0.54-0.87x the issue time.
#pragma once
#include <hip/hip_runtime.h>

namespace sarx {
namespace pk {

typedef float v2 __attribute__((ext_vector_type(2)));    // .x = re (low register), .y = im (high register)

__device__ __forceinline__ v2 add(v2 a, v2 b) {
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2 sub(v2 a, v2 b) {
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + (-i) b = (a.re + b.im, a.im - b.re)
__device__ __forceinline__ v2 add_mi(v2 a, v2 b) {
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + (+i) b = (a.re - b.im, a.im + b.re)
__device__ __forceinline__ v2 add_pi(v2 a, v2 b) {
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + (-+i) b for the forward / inverse transform, and a - (-+i) b
template <bool INV> __device__ __forceinline__ v2 add_rot(v2 a, v2 b) { return INV ? add_pi(a, b) : add_mi(a, b); }
template <bool INV> __device__ __forceinline__ v2 sub_rot(v2 a, v2 b) { return INV ? add_mi(a, b) : add_pi(a, b); }

// a * w:  t = (a.re w.re, a.im w.re);  d = (t.lo - a.im w.im, t.hi + a.re w.im)
__device__ __forceinline__ v2 cmul(v2 a, v2 w) {
    v2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}
// a * conj(w):  d = (t.lo + a.im w.im, t.hi - a.re w.im)
__device__ __forceinline__ v2 cmul_conj(v2 a, v2 w) {
    v2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}
// a * w with w = exp(-+ i theta) given as the forward value (c, -s): the inverse uses its conjugate
template <bool INV> __device__ __forceinline__ v2 cmul_dir(v2 a, v2 w) { return INV ? cmul_conj(a, w) : cmul(a, w); }
// a * s for a real scalar pair (s, s)
__device__ __forceinline__ v2 scale(v2 a, v2 ss) {
    v2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(ss));
    return d;
}

// 4-point DFT, natural order in and out.  C_ROT: input c still carries a pending factor (-+i) (folded into the first layer)
template <bool INV, bool C_ROT = false> __device__ __forceinline__ void dft4(v2& v0, v2& v1, v2& v2_, v2& v3) {
    const v2 a0 = C_ROT ? add_rot<INV>(v0, v2_) : add(v0, v2_);
    const v2 a1 = C_ROT ? sub_rot<INV>(v0, v2_) : sub(v0, v2_);
    const v2 a2 = add(v1, v3), a3 = sub(v1, v3);          // a3 is to be multiplied by -+i
    v0 = add(a0, a2);
    v2_ = sub(a0, a2);
    v1 = add_rot<INV>(a1, a3);
    v3 = sub_rot<INV>(a1, a3);
}

// 16-point DFT as 4 x 4 (n = n1 + 4 n2, k = 4 k1 + k2), natural order in and out; 80 packed instructions
template <bool INV> __device__ __forceinline__ void dft16(v2* v) {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    // forward values W16^m = (cos, -sin); the inverse multiplies by their conjugates
    const v2 w1 = {c1, -s1}, w2 = {h, -h}, w3 = {s1, -c1}, w6 = {-h, -h}, w9 = {-c1, s1};
    v2 y[4][4];
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) {
        v2 a = v[n1], b = v[n1 + 4], c = v[n1 + 8], d = v[n1 + 12];
        dft4<INV>(a, b, c, d);
        y[n1][0] = a; y[n1][1] = b; y[n1][2] = c; y[n1][3] = d;
    }
    y[1][1] = cmul_dir<INV>(y[1][1], w1); y[1][2] = cmul_dir<INV>(y[1][2], w2); y[1][3] = cmul_dir<INV>(y[1][3], w3);
    y[2][1] = cmul_dir<INV>(y[2][1], w2); /* y[2][2] *= -+i: folded into the k2 = 2 column below */ y[2][3] = cmul_dir<INV>(y[2][3], w6);
    y[3][1] = cmul_dir<INV>(y[3][1], w3); y[3][2] = cmul_dir<INV>(y[3][2], w6); y[3][3] = cmul_dir<INV>(y[3][3], w9);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        v2 a = y[0][k2], b = y[1][k2], c = y[2][k2], d = y[3][k2];
        if (k2 == 2) dft4<INV, true>(a, b, c, d);
        else dft4<INV, false>(a, b, c, d);
        v[k2] = a; v[4 + k2] = b; v[8 + k2] = c; v[12 + k2] = d;
    }
}

// 32 points: two 16-point DFTs of the even and odd samples, then X[k] = E[k] +- W32^k O[k]
template <bool INV> __device__ __forceinline__ void dft32(v2* v) {
    v2 e[16], o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    dft16<INV>(e);
    dft16<INV>(o);
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
        if (k == 0) { v[0] = add(e[0], o[0]); v[16] = sub(e[0], o[0]); }
        else if (k == 8) { v[8] = add_rot<INV>(e[8], o[8]); v[24] = sub_rot<INV>(e[8], o[8]); }     // W32^8 = -+i
        else {
            const v2 w = {C32[k], -S32[k]};
            const v2 t = cmul_dir<INV>(o[k], w);
            v[k] = add(e[k], t);
            v[k + 16] = sub(e[k], t);
        }
    }
}

}  // namespace pk
}  // namespace sarx
