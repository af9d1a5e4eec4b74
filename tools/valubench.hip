// VALU issue-rate probe on MI355X: scalar v_fma_f32 / v_add_f32 vs packed v_pk_fma_f32 / v_pk_add_f32,
// at 1, 2 and 4 waves per SIMD.  Answers: how many cycles does a SIMD need per wave64 FP32 instruction?
// build: hipcc -O3 --offload-arch=gfx950 tools/valubench.hip -o tools/valubench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096, ACC = 16;

template <int MODE> __global__ __launch_bounds__(256) void probe(float* out, float s) {
    float a[ACC];
    v2f p[ACC / 2];
#pragma unroll
    for (int i = 0; i < ACC; ++i) a[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < ACC / 2; ++i) p[i] = v2f{a[2 * i], a[2 * i + 1]};
    const v2f s2 = {s, s * 1.0001f};
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (MODE == 0) {          // scalar fma: ACC independent chains
#pragma unroll
            for (int i = 0; i < ACC; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
        } else if constexpr (MODE == 1) {   // scalar add
#pragma unroll
            for (int i = 0; i < ACC; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        } else if constexpr (MODE == 2) {   // packed fma: ACC/2 chains, 2 floats each
#pragma unroll
            for (int i = 0; i < ACC / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(s2));
        } else if constexpr (MODE == 3) {   // packed add
#pragma unroll
            for (int i = 0; i < ACC / 2; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
        } else if constexpr (MODE == 4) {   // packed mul
#pragma unroll
            for (int i = 0; i < ACC / 2; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
        } else if constexpr (MODE == 5) {   // v_sin_f32
#pragma unroll
            for (int i = 0; i < ACC; ++i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
        } else if constexpr (MODE == 6) {   // fp64 fma
#pragma unroll
            for (int i = 0; i < ACC / 2; ++i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(*(double*)&p[i]) : "v"(*(const double*)&s2));
        } else if constexpr (MODE == 7) {   // int add
#pragma unroll
            for (int i = 0; i < ACC; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) r += a[i];
#pragma unroll
    for (int i = 0; i < ACC / 2; ++i) r += p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE> static void run(const char* name, int insts_per_iter, int wgs_per_cu, float* out) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * wgs_per_cu;                  // 256 threads = 4 waves = 1 wave per SIMD per workgroup
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, 1.0001f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, 1.0001f);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    // wave-instructions per SIMD = wgs_per_cu * ITERS * insts_per_iter ; report ns and cycles@2.4GHz per wave-instruction per SIMD
    const double wi = (double)wgs_per_cu * ITERS * insts_per_iter;
    printf("%-14s waves/SIMD %d : %8.3f ms  -> %6.3f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", name, wgs_per_cu, ms,
           ms * 1e6 / wi, ms * 1e6 / wi * 2.4);
}

int main() {
    float* out;
    CK(hipMalloc(&out, 256 * 256 * 8 * sizeof(float)));
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", ACC, w, out);
        run<1>("v_add_f32", ACC, w, out);
        run<2>("v_pk_fma_f32", ACC / 2, w, out);
        run<3>("v_pk_add_f32", ACC / 2, w, out);
        run<4>("v_pk_mul_f32", ACC / 2, w, out);
        run<5>("v_sin_f32", ACC, w, out);
        run<6>("v_fma_f64", ACC / 2, w, out);
        run<7>("v_add_u32", ACC, w, out);
    }
    return 0;
}
