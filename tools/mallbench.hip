// Does the 256 MiB Infinity Cache (memory-side) serve a slab that one kernel writes and the next reads?
// Decides whether the middle of the CSA chain (azimuth step B -> fused range pass -> inverse azimuth step A, which all
// work on the same 128-row groups) should run slab by slab instead of image by image.
//   A  ping-pong copies T1 <-> T2 of S MiB each (everything stays on die if the cache keeps written lines)
//   B  pipeline over a 2 GiB image in slabs of S MiB:  IN[slab] -> T1,  T1 -> T2,  T2 -> OUT[slab]
//      against three image-sized copies (what five-launch focusing costs per middle section today)
// build: hipcc -O3 --offload-arch=gfx950 tools/mallbench.hip -o gpurun_out/mallbench ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_k(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += stride) {
        float4 a = in[i], b = in[i + 256], c = in[i + 512], d = in[i + 768];
        out[i] = a; out[i + 256] = b; out[i + 512] = c; out[i + 768] = d;
    }
}
static void copy(const void* in, void* out, size_t bytes, hipStream_t st = 0) {
    const size_t n4 = bytes / 16;
    size_t blocks = n4 / 1024;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(copy_k, dim3((unsigned)blocks), dim3(256), 0, st, (const float4*)in, (float4*)out, n4);
}

template <class F> static float time_ms(F f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
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
    const size_t IMG = (size_t)2 << 30;
    char *in, *out, *t1, *t2;
    CK(hipMalloc(&in, IMG)); CK(hipMalloc(&out, IMG));
    CK(hipMalloc(&t1, (size_t)512 << 20)); CK(hipMalloc(&t2, (size_t)512 << 20));
    CK(hipMemset(in, 1, IMG)); CK(hipMemset(out, 0, IMG)); CK(hipMemset(t1, 0, (size_t)512 << 20)); CK(hipMemset(t2, 0, (size_t)512 << 20));

    const float full = time_ms([&] { copy(in, out, IMG); }, 5);
    printf("image copy 2 GiB -> 2 GiB                     %7.3f ms  %7.1f GB/s (read+write)\n", full, 2.0 * IMG / full / 1e6);

    for (int mib : {4, 8, 16, 32, 64, 96, 128, 256, 512}) {
        const size_t S = (size_t)mib << 20;
        const int reps = (int)(((size_t)4 << 30) / S);
        const float ms = time_ms([&] { for (int r = 0; r < reps; r += 2) { copy(t1, t2, S); copy(t2, t1, S); } }, 2) / reps;
        printf("A ping-pong  slab %4d MiB                      %7.2f us per copy  %7.1f GB/s (read+write)\n", mib, ms * 1e3, 2.0 * S / ms / 1e6);
    }
    for (int mib : {8, 16, 32, 64, 128}) {
        const size_t S = (size_t)mib << 20;
        const size_t slabs = IMG / S;
        const float ms = time_ms([&] {
            for (size_t s = 0; s < slabs; ++s) { copy(in + s * S, t1, S); copy(t1, t2, S); copy(t2, out + s * S, S); }
        }, 3);
        printf("B slab pipeline %4d MiB: in->T1->T2->out       %7.3f ms per image (3 image copies: %7.3f ms)\n", mib, ms, 3 * full);
        const float ms2 = time_ms([&] {
            for (size_t s = 0; s < slabs; ++s) { copy(in + s * S, t1, S); copy(t1, t1, S); copy(t1, out + s * S, S); }
        }, 3);
        printf("B slab pipeline %4d MiB: in->T1->T1->out       %7.3f ms per image\n", mib, ms2);
        const float ms3 = time_ms([&] {
            for (size_t s = 0; s < slabs; ++s) { copy(in + s * S, t1, S); copy(t1, out + s * S, S); }
        }, 3);
        printf("B slab pipeline %4d MiB: in->T1->out           %7.3f ms per image (2 image copies: %7.3f ms)\n", mib, ms3, 2 * full);
    }
    return 0;
}
