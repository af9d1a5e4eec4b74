// Is packed fp32 (v_pk_*_f32 with op_sel / neg modifiers) worth it for the FFT butterflies at the fused range kernel's
// occupancy (2 waves per SIMD)?  Times a register-resident loop of  dft16 + 15 twiddle multiplies  per thread in two
// forms: the scalar butterflies of fft_core.hpp, and the same arithmetic on (re, im) register pairs with packed
// instructions whose swizzles (multiply by -+i, broadcast, swap) ride on the instruction modifiers.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -fno-slp-vectorize -Inis-sar-amtigmti-video_amd/csrc -Itools tools/pkbench.hip -o tools/pkbench.bin
// Result on MI355X (profiles/r02_pkbench.log): packed 0.152 ms, scalar 0.145 ms - 0.96x.  Half the instructions, no gain:
// with both waves of a SIMD issuing, a packed instruction occupies the pipe twice as long.  Not adopted.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fft_core.hpp"
#include "fft_pk.hpp"

using namespace sarx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 256;

template <bool INV> __global__ __launch_bounds__(512, 2) void scalar_k(const cf* in, cf* out, const cf* tw) {
    extern __shared__ char smem[];
    cf v[16], w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = in[threadIdx.x * 16 + i]; w[i] = tw[(threadIdx.x % 64) * 16 + i]; }
    for (int it = 0; it < ITERS; ++it) {
        dft16<INV>(v);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = cmul(v[i], w[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[(blockIdx.x * 512 + threadIdx.x) * 16 + i] = v[i];
    if (in == nullptr) smem[threadIdx.x] = 0;
}
template <bool INV> __global__ __launch_bounds__(512, 2) void packed_k(const cf* in, cf* out, const cf* tw) {
    extern __shared__ char smem[];
    pk::v2 v[16], w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const cf a = in[threadIdx.x * 16 + i], b = tw[(threadIdx.x % 64) * 16 + i];
        v[i] = pk::v2{a.x, a.y}; w[i] = pk::v2{b.x, b.y};
    }
    for (int it = 0; it < ITERS; ++it) {
        pk::dft16<INV>(v);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = pk::cmul(v[i], w[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[(blockIdx.x * 512 + threadIdx.x) * 16 + i] = make_float2(v[i].x, v[i].y);
    if (in == nullptr) smem[threadIdx.x] = 0;
}

int main() {
    const int grid = 256, n = 512 * 16;
    std::vector<cf> h(n), tw(64 * 16);
    for (int i = 0; i < n; ++i) h[i] = make_float2((float)std::sin(0.37 * i + 1.0), (float)std::cos(0.91 * i));
    for (int i = 0; i < 64 * 16; ++i) tw[i] = make_float2(0.25f * (float)std::cos(0.37 * i), 0.25f * (float)std::sin(0.37 * i));   // modulus 1/4 = 1/sqrt(16): magnitudes stay put on average
    cf *din, *dtw, *o1, *o2;
    CK(hipMalloc(&din, n * sizeof(cf))); CK(hipMalloc(&dtw, tw.size() * sizeof(cf)));
    CK(hipMalloc(&o1, (size_t)grid * n * sizeof(cf))); CK(hipMalloc(&o2, (size_t)grid * n * sizeof(cf)));
    CK(hipMemcpy(din, h.data(), n * sizeof(cf), hipMemcpyHostToDevice));
    CK(hipMemcpy(dtw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
    const size_t lds = 139264;          // one workgroup of 8 waves per CU, like range_fused_wl_kernel
    CK(hipFuncSetAttribute((const void*)scalar_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)packed_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto time = [&](auto kern, cf* out, const char* name) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, din, out, dtw);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, din, out, dtw);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("%-8s %8.3f ms  (%d x [dft16 + 15 cmul] per thread, 2 waves per SIMD)\n", name, ms, ITERS);
        return ms;
    };
    const float ts = time(scalar_k<false>, o1, "scalar");
    const float tp = time(packed_k<false>, o2, "packed");
    std::vector<cf> r1(n), r2(n);
    CK(hipMemcpy(r1.data(), o1, n * sizeof(cf), hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), o2, n * sizeof(cf), hipMemcpyDeviceToHost));
    double err = 0, nrm = 0;
    for (int i = 0; i < n; ++i) {
        err += (double)(r1[i].x - r2[i].x) * (r1[i].x - r2[i].x) + (double)(r1[i].y - r2[i].y) * (r1[i].y - r2[i].y);
        nrm += (double)r1[i].x * r1[i].x + (double)r1[i].y * r1[i].y;
    }
    printf("packed vs scalar result: rel-L2 %.2e (both run %d rounds of unnormalised dft16: growth 16^%d, compared relative)\n",
           std::sqrt(err / nrm), ITERS, ITERS);
    printf("speed-up of the packed form: %.2fx\n", ts / tp);
    return 0;
}
