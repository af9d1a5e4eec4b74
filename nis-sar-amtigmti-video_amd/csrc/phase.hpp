// CSA phase functions Phi_1..Phi_3 (sar_ati_dcpa_sim_csa.py:262-380).
//
// All three are quadratic in the sample index along a line.  A thread's
// samples form arithmetic progressions of that index, so each progression is
// seeded once in fp64 (arguments reach 3e7 revolutions in Phi_3: fp32 cannot
// hold them) and then stepped as a 32-bit fixed-point phase accumulator in
// revolutions (p += d; d += dd, wrapping = reduction mod 1 for free), the
// way a hardware NCO does.  Quantising p, d, dd to 2^-32 rev bounds the error
// after m steps by 2^-33 (1 + m + m(m-1)/2) rev: 9e-8 rad at m = 15, below
// the fp32 rounding of the sine/cosine argument itself.
#pragma once
#include "fft_core.hpp"

namespace sarx {

struct FixPhase {
    unsigned p, d, dd;       // revolutions * 2^32
    __device__ __forceinline__ cf next() {
        const float x = (float)(int)p * 2.3283064365386963e-10f;   // [-0.5, 0.5)
        p += d;
        d += dd;
        return cis_frac(x);
    }
};
__device__ __forceinline__ unsigned rev_to_fix(double r) {
    const double f = r - rint(r);
    return (unsigned)(long long)rint(f * 4294967296.0);
}
__device__ __forceinline__ FixPhase make_fix(double p, double d, double dd) {
    FixPhase q;
    q.p = rev_to_fix(p);
    q.d = rev_to_fix(d);
    q.dd = rev_to_fix(dd);
    return q;
}

// Phi_2[i,k] = exp(j[pi fr_k^2/(Kr(1+Cs_i)) + 4 pi R_ref Cs_i fr_k / c])      (:318-324)
//   c2[i] = { 0.5/(Kr(1+Cs_i)),  2 R_ref Cs_i / c };  fr = ks*df, ks the signed fftfreq index.
// Progression ks0, ks0+step, ...
__device__ __forceinline__ FixPhase phi2_seed(int ks0, int step, double2 c2, double df) {
    const double f = (double)ks0 * df, h = (double)step * df;
    return make_fix(f * fma(c2.x, f, c2.y), h * fma(c2.x, 2.0 * f + h, c2.y), 2.0 * c2.x * h * h);
}
// Phi_3[i,j] = exp(j[4 pi (c tau_j/2) D_i/lam - pi Kr Cs_i(1+Cs_i)(tau_j - 2R_ref/c)^2])   (:359,375-380)
//   c3[i] = { c D_i / lam,  -0.5 Kr Cs_i (1+Cs_i) };  progression j0, j0+step, ...
__device__ __forceinline__ FixPhase phi3_seed(int j0, int step, double2 c3, double dt, double t_start, double t0) {
    const double tau = __dadd_rn(t_start, __dmul_rn((double)j0, dt));   // :219, unfused like NumPy
    const double d0 = tau - t0, h = (double)step * dt;
    return make_fix(fma(c3.x, tau, c3.y * d0 * d0), h * fma(c3.y, 2.0 * d0 + h, c3.x), 2.0 * c3.y * h * h);
}
// One double2 of a per-row table through the SCALAR cache (p is wave-uniform).  vmcnt retires in order, so a vector load
// of the row constants issued after the next line's prefetch would make the wave wait for the whole prefetch right there
// (the compiler emits a vector load: it cannot prove the table is not written by the kernel).
__device__ __forceinline__ double2 sload_double2(const double2* p) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 r;
    // the address is wave-uniform by construction; readfirstlane tells the compiler so (it does not move a VGPR pair into the
    // "s" operand by itself when its own analysis calls the value divergent)
    const unsigned long long pv = (unsigned long long)p;
    const unsigned long long ps = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pv);
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(ps) : "memory");
    const unsigned long long lo = ((unsigned long long)r[1] << 32) | r[0], hi = ((unsigned long long)r[3] << 32) | r[2];
    return make_double2(__longlong_as_double((long long)lo), __longlong_as_double((long long)hi));
}
// direct forms (one fp64 evaluation per sample): any line length, no progression needed
__device__ __forceinline__ cf phi2_at(int ks, double2 c2, double df) {
    const double f = (double)ks * df;
    return cis_rev(f * fma(c2.x, f, c2.y));
}
__device__ __forceinline__ cf phi3_at(int j, double2 c3, double dt, double t_start, double t0) {
    const double tau = __dadd_rn(t_start, __dmul_rn((double)j, dt));   // :219, unfused like NumPy
    const double d = tau - t0;
    return cis_rev(fma(c3.x, tau, c3.y * d * d));
}
// Phi_1[i,j] = exp(-j pi Kr Cs_i (tau_j - tau_ref_i)^2)                           (:262-272)
//   c1[i] = { -0.5 Kr Cs_i, tau_ref_i }.  In the azimuth tile a thread's samples
// share j and differ in i, so this one is evaluated directly.
__device__ __forceinline__ cf phi1(int j, double2 c1, double dt, double t_start) {
    const double tau = __dadd_rn(t_start, __dmul_rn((double)j, dt));
    const double d = tau - c1.y;
    return cis_rev(c1.x * d * d);
}

}  // namespace sarx
