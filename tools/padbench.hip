// Does a padded leading dimension help the azimuth-tile access pattern?  Rows of a 16384-sample image are
// 128 KiB apart: every row of a tile starts at the same offset within any power-of-two interleave.
// Tile copies (128 rows x 32 cols, 256-byte row segments) for step A (rows q + 128 m) and step B (rows 128 q + m in,
// q + 128 m out) with ld = 16384 + pad samples.
// build: hipcc -O3 --offload-arch=gfx950 tools/padbench.hip -o tools/padbench.bin ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int W>
__global__ __launch_bounds__(8 * W) void tile_copy(const float2* __restrict__ in, float2* __restrict__ out, size_t ld_in, size_t ld_out,
                                                   int in_q, int in_m, int out_q, int out_m) {
    const int c = threadIdx.x % W, t = threadIdx.x / W;
    const int col = blockIdx.x * W + c, q = blockIdx.y;
    float2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = in[((size_t)q * in_q + (size_t)(t + 8 * i) * in_m) * ld_in + col];
#pragma unroll
    for (int i = 0; i < 16; ++i) out[((size_t)q * out_q + (size_t)(t + 8 * i) * out_m) * ld_out + col] = v[i];
}

template <class F> static float time_ms(F f, int iters = 10) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / iters;
}

int main() {
    const int n = 16384, S = 128;
    const size_t max_ld = n + 1024;
    float2 *in, *out;
    CK(hipMalloc(&in, max_ld * n * 8)); CK(hipMalloc(&out, max_ld * n * 8));
    CK(hipMemset(in, 1, max_ld * n * 8)); CK(hipMemset(out, 0, max_ld * n * 8));
    const double gb = 16.0 * n * n / 1e9;
    for (int pad_in : {0, 32, 64, 128, 256, 544})
        for (int pad_out : {0, 32, 64, 128, 256, 544}) {
            if (pad_in != pad_out && pad_in != 0 && pad_out != 0) continue;
            const size_t li = n + pad_in, lo = n + pad_out;
            const float a = time_ms([&] { hipLaunchKernelGGL((tile_copy<32>), dim3(n / 32, S), dim3(256), 0, 0, in, out, li, lo, 1, S, 1, S); });
            const float b = time_ms([&] { hipLaunchKernelGGL((tile_copy<32>), dim3(n / 32, S), dim3(256), 0, 0, in, out, li, lo, S, 1, 1, S); });
            printf("pad in %4d out %4d : step A %.3f ms %6.0f GB/s | step B %.3f ms %6.0f GB/s\n", pad_in, pad_out, a, gb / (a * 1e-3), b,
                   gb / (b * 1e-3));
        }
    return 0;
}
