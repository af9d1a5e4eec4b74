// One pixel of the ATI / DPCA products (sar_ati_dcpa_sim_csa.py:414-419, viewer :42-52): shared by the streaming kernel of
// products.hip and by the azimuth tile epilogue that emits the products while the second channel's image is being written.
#pragma once
#include <hip/hip_runtime.h>

namespace sarx {

typedef float2 cf;

struct Pix {
    float phase, m1, dm, m2, p1, p2, dp;
    cf interf, diff;
    double sre, sim;
};

template <bool OPT> __device__ __forceinline__ void ati_pixel(cf a, cf b, float cc, float cs, Pix& o) {
    // uncalibrated interferogram feeds the phase-balance sum (viewer :249)
    o.sre = (double)a.x * b.x + (double)a.y * b.y;
    o.sim = (double)a.y * b.x - (double)a.x * b.y;
    const cf bc = make_float2(b.x * cc - b.y * cs, b.x * cs + b.y * cc);   // s2 * exp(i cal)  (viewer :43)
    const cf in = make_float2(fmaf(a.x, bc.x, a.y * bc.y), fmaf(a.y, bc.x, -a.x * bc.y));   // a * conj(bc)  (:414)
    const cf df = make_float2(a.x - bc.x, a.y - bc.y);                                     // (:418)
    o.phase = atan2f(in.y, in.x);      // (:415)
    o.m1 = hypotf(a.x, a.y);           // (:416)
    o.dm = hypotf(df.x, df.y);         // (:419)
    o.interf = in;
    o.diff = df;
    if (OPT) {
        o.m2 = hypotf(bc.x, bc.y);
        o.p1 = atan2f(a.y, a.x);
        o.p2 = atan2f(bc.y, bc.x);
        o.dp = atan2f(df.y, df.x);
    }
}


}  // namespace sarx
