// Can ONE launch do a 16384-point column (azimuth) transform, i.e. one HBM round trip of the image per transform instead of the
// two of the four-step pair?  A whole column must then sit on one CU: 16384 x 8 B = 128 KiB per column, so a CU can hold a strip
// only WB = 8 or 16 bytes wide (one or two columns), and its global accesses are WB bytes per row with rows 128 KiB apart - far
// below the 128-byte line.  The bet: the CUs of ONE XCD take the 8 (16) neighbouring strips of the same 128-byte lines at the same
// time, so every line is fetched once into that XCD's L2 and the partial stores merge there.
// Traffic-only experiment (load a strip into registers, workgroup barrier, store it): what rate does this pattern reach?
//   MAP 0: strip = it * grid + blockIdx              neighbouring strips on DIFFERENT XCDs (workgroup b runs on XCD b % 8)
//   MAP 1: the 128 / WB strips of a line on consecutive workgroups of one XCD; an XCD covers consecutive lines, the eight XCDs
//          together a contiguous run of each row
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/colbench.hip -o tools/colbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int WB> struct Vec;
template <> struct Vec<16> { using T = f4; };
template <> struct Vec<8> { using T = f2; };

constexpr int N = 16384;

template <int WB, int THREADS, int MAP, bool NT>
__global__ __launch_bounds__(THREADS) void strip_kernel(const char* __restrict__ in, char* __restrict__ out, size_t pitch, int n_strips,
                                                        unsigned* xcd_mismatch) {
    extern __shared__ char lds[];
    using V = typename Vec<WB>::T;
    constexpr int RPT = N / THREADS;
    const int b = blockIdx.x, G = gridDim.x;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((xcc & 7u) != (unsigned)(b & 7)) atomicAdd(xcd_mismatch, 1u);
    }
    for (int it = 0;; ++it) {
        int strip;
        if (MAP == 0) strip = it * G + b;
        else {
            constexpr int PER = 128 / WB;                 // strips per 128-byte line
            const int x = b & 7, i = b >> 3;
            const int lg = (G / 8) / PER;                 // lines per XCD and iteration
            strip = ((it * 8 + x) * lg + i / PER) * PER + i % PER;
        }
        if (strip >= n_strips) break;
        V v[RPT];
        const char* src = in + (size_t)strip * WB + (size_t)threadIdx.x * pitch;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const V* p = (const V*)(src + (size_t)k * THREADS * pitch);
            v[k] = NT ? __builtin_nontemporal_load(p) : *p;
        }
        __syncthreads();
        char* dst = out + (size_t)strip * WB + (size_t)threadIdx.x * pitch;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            V* p = (V*)(dst + (size_t)k * THREADS * pitch);
            const V o = v[k] * 0.5f;
            if (NT) __builtin_nontemporal_store(o, p); else *p = o;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ in, f4* __restrict__ out, size_t n4) {
    const size_t i0 = (size_t)blockIdx.x * 2048 + threadIdx.x;
    f4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(in + i0 + k * 256);
#pragma unroll
    for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k] * 0.5f, out + i0 + k * 256);
}

static float* d_in; static float* d_out; static unsigned* d_mis;

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

static bool check(float scale_expected) {
    std::vector<float> a(4096), b(4096);
    bool ok = true;
    for (size_t row : {(size_t)0, (size_t)777, (size_t)16383}) {
        CK(hipMemcpy(a.data(), d_in + row * N * 2 + 4096 * 3, 4096 * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), d_out + row * N * 2 + 4096 * 3, 4096 * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < 4096; ++i) if (b[i] != a[i] * scale_expected) { ok = false; break; }
    }
    return ok;
}

template <int WB, int THREADS, int MAP, bool NT> static void run(int wgs_per_cu, bool inplace) {
    const size_t pitch = (size_t)N * 8;
    const int n_strips = N * 8 / WB;
    const int grid = 256 * wgs_per_cu;
    const size_t lds = wgs_per_cu == 1 ? 100 * 1024 : 60 * 1024;
    auto k = strip_kernel<WB, THREADS, MAP, NT>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(d_mis, 0, 4));
    CK(hipMemset(d_out, 0, (size_t)N * N * 8));
    char* o = inplace ? (char*)d_in : (char*)d_out;
    if (!inplace) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, (const char*)d_in, o, pitch, n_strips, d_mis);
        CK(hipDeviceSynchronize());
    }
    const bool ok = inplace ? true : check(0.5f);
    const float ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, (const char*)d_in, o, pitch, n_strips, d_mis); }, 5);
    unsigned mis = 0;
    CK(hipMemcpy(&mis, d_mis, 4, hipMemcpyDeviceToHost));
    printf("strip %2d B x %4d thr  map %d  %s  %d WG/CU  %s   %7.3f ms  %6.2f TB/s  %s  xcd-mismatch %u\n", WB, THREADS, MAP, NT ? "nt   " : "plain",
           wgs_per_cu, inplace ? "in place " : "out of pl", ms, 2.0 * N * N * 8 / ms / 1e9, ok ? "ok" : "WRONG", mis);
    fflush(stdout);
}

int main() {
    const size_t bytes = (size_t)N * N * 8;
    CK(hipMalloc(&d_in, bytes)); CK(hipMalloc(&d_out, bytes)); CK(hipMalloc(&d_mis, 4));
    std::vector<float> h((size_t)1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) + 1.0f;
    for (size_t off = 0; off < bytes; off += h.size() * 4) CK(hipMemcpy((char*)d_in + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const float c = time_ms([&] { hipLaunchKernelGGL(copy_k, dim3((unsigned)(bytes / 16 / 2048)), dim3(256), 0, 0, (const f4*)d_in, (f4*)d_out, bytes / 16); }, 5);
    printf("flat copy 2 GiB -> 2 GiB (8 x float4 per lane, nontemporal)   %7.3f ms  %6.2f TB/s\n", c, 2.0 * bytes / c / 1e9);

    run<16, 512, 1, false>(1, false);
    run<16, 512, 0, false>(1, false);
    run<16, 512, 1, true>(1, false);
    run<16, 1024, 1, false>(1, false);
    run<16, 1024, 1, false>(2, false);
    run<16, 512, 1, false>(2, false);
    run<16, 512, 1, false>(1, true);
    run<8, 1024, 1, false>(1, false);
    run<8, 1024, 1, false>(2, false);
    run<8, 1024, 0, false>(1, false);
    return 0;
}
