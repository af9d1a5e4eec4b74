// In-register DFTs of fft_mixed.hpp against a direct O(R^2) sum in double (GPU box):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Inis-sar-amtigmti-video_amd/csrc tools/dfttest.hip -o tools/dfttest.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>
#include "fft_mixed.hpp"
using namespace sarx;

template <int R, bool INV> __global__ void k(const cf* in, cf* out) {
    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = in[threadIdx.x * R + i];
    mix::dft_any<R, INV>(v);
#pragma unroll
    for (int i = 0; i < R; ++i) out[threadIdx.x * R + i] = v[i];
}

template <int R, bool INV> static int check() {
    const int T = 64;
    std::vector<cf> h(T * R), o(T * R);
    for (int i = 0; i < T * R; ++i) h[i] = make_float2((float)std::sin(0.37 * i + 1.0), (float)std::cos(0.91 * i));
    cf *di, *dout;
    hipMalloc(&di, h.size() * sizeof(cf)); hipMalloc(&dout, h.size() * sizeof(cf));
    hipMemcpy(di, h.data(), h.size() * sizeof(cf), hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<R, INV>), dim3(1), dim3(T), 0, 0, di, dout);
    hipMemcpy(o.data(), dout, h.size() * sizeof(cf), hipMemcpyDeviceToHost);
    double err = 0, nrm = 0;
    for (int t = 0; t < T; ++t)
        for (int kk = 0; kk < R; ++kk) {
            std::complex<double> s = 0;
            for (int n = 0; n < R; ++n)
                s += std::complex<double>(h[t * R + n].x, h[t * R + n].y) * std::polar(1.0, (INV ? 2.0 : -2.0) * M_PI * n * kk / R);
            const std::complex<double> g(o[t * R + kk].x, o[t * R + kk].y);
            err += std::norm(g - s); nrm += std::norm(s);
        }
    const double rel = std::sqrt(err / nrm);
    printf("dft<%2d,%s> rel-L2 %.2e %s\n", R, INV ? "inv" : "fwd", rel, rel < 1e-6 ? "ok" : "FAIL");
    hipFree(di); hipFree(dout);
    return rel < 1e-6 ? 0 : 1;
}

int main() {
    int bad = 0;
    bad += check<3, false>() + check<3, true>() + check<5, false>() + check<5, true>() + check<7, false>() + check<11, false>() +
           check<11, true>() + check<13, true>() + check<6, false>() + check<9, false>() + check<10, true>() + check<12, false>() +
           check<15, true>() + check<20, false>() + check<22, false>() + check<22, true>() + check<24, false>() + check<24, true>() +
           check<25, false>() + check<25, true>() + check<30, false>() + check<33, true>();
    printf(bad ? "FAILED %d\n" : "all ok\n", bad);
    return bad;
}
