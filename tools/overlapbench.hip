// Do in-flight global loads overlap with arithmetic on this part the way the prefetching kernels assume?
// Persistent workgroups (one per CU, 512 threads), tile = 64 KiB in + 64 KiB out, 2 GiB each way in total.
//   mode 0: load tile, wait, FMA loop of `work` iterations, store                         (phases in sequence, like the product kernels)
//   mode 1: loads of tile i+1 issued before the FMA loop of tile i, consumed after it     (register prefetch)
//   mode 2: as mode 1 with the loads spread over the FMA loop in 8 slices                 (continuous issue instead of a burst)
// Sweeping `work` shows whether time = memory + arithmetic or max(memory, arithmetic).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/overlapbench.hip -o tools/overlapbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
#ifndef UU
#define UU 8
#endif
constexpr int T = 512, U = UU;           // threads, float4 per thread per tile (UU=16: 128 KiB tiles, like one 16384-sample line)
#ifndef LDSX
#define LDSX 0                            // LDSX=n: n exchange rounds through LDS (write tile, barrier, read permuted, barrier) inside the arithmetic
#endif

__device__ __forceinline__ void burn1(f4 (&d)[U], int iters, float s) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(d[u].x) : "v"(s));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(d[u].y) : "v"(s));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(d[u].z) : "v"(s));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(d[u].w) : "v"(s));
        }
    }
}

__device__ __forceinline__ void burn(f4 (&d)[U], int iters, float s) {
#if LDSX == 0
    burn1(d, iters, s);
#else
    extern __shared__ f4 lds[];
    for (int x = 0; x < LDSX; ++x) {
        burn1(d, iters / LDSX, s);
#pragma unroll
        for (int u = 0; u < U; ++u) lds[u * T + threadIdx.x] = d[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = lds[u * T + (threadIdx.x ^ (1 + x))];
        __syncthreads();
    }
#endif
}

template <int MODE, bool SYNC> __global__ __launch_bounds__(T) void k(const f4* in, f4* out, int n_tiles, int work, float s) {
    const size_t tile_elems = (size_t)T * U;
    f4 d[U], p[U];
    int tile = blockIdx.x;
    auto load = [&](f4 (&r)[U], int t) {
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = __builtin_nontemporal_load(in + (size_t)t * tile_elems + (size_t)u * T + threadIdx.x);
    };
    auto store = [&](f4 (&r)[U], int t) {
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(r[u], out + (size_t)t * tile_elems + (size_t)u * T + threadIdx.x);
    };
    if (MODE == 0) {
        for (; tile < n_tiles; tile += gridDim.x) {
            load(d, tile);
            if (SYNC) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // a tile-wide exchange needs the whole tile: waves in lockstep
            burn(d, work, s);
            if (SYNC) __syncthreads();
            store(d, tile);
        }
    } else {
        if (tile < n_tiles) load(d, tile);
        for (; tile < n_tiles; tile += gridDim.x) {
            const int next = tile + (int)gridDim.x < n_tiles ? tile + (int)gridDim.x : tile;     // last round re-reads its own tile
            if (SYNC) __syncthreads();
            if (MODE == 1) {
                load(p, next);
                burn(d, work, s);
                if (SYNC) __syncthreads();
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    p[u] = __builtin_nontemporal_load(in + (size_t)next * tile_elems + (size_t)u * T + threadIdx.x);
                    burn(d, work / U, s);
                }
                if (SYNC) __syncthreads();
            }
            store(d, tile);
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = p[u];
        }
    }
}

int main() {
    const size_t LDS_BYTES = LDSX ? (size_t)T * U * 16 : 0;
    if (LDSX) {
        CK(hipFuncSetAttribute((const void*)k<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES)); CK(hipFuncSetAttribute((const void*)k<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        CK(hipFuncSetAttribute((const void*)k<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES)); CK(hipFuncSetAttribute((const void*)k<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        CK(hipFuncSetAttribute((const void*)k<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES)); CK(hipFuncSetAttribute((const void*)k<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    }
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const size_t bytes = (size_t)2 << 30;
    const int n_tiles = (int)(bytes / (T * U * 16));
    f4 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("U=%d LDSX=%d: ", U, LDSX); printf("%d tiles of 64 KiB (x U/8), 256 persistent workgroups of 512 threads; ms per pass over 2 GiB in + 2 GiB out\n", n_tiles);
    printf("%8s %12s %12s %12s | %12s %12s %12s  (right: workgroup barriers between the phases)\n", "work", "sequential", "prefetch", "spread", "sequential", "prefetch", "spread");
    for (int work : {0, 32, 64, 96, 128, 192}) {
        float ms[6];
        for (int mode = 0; mode < 6; ++mode) {
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL((k<0, false>), dim3(256), dim3(T), LDS_BYTES, 0, in, out, n_tiles, work, 1.0001f);
                else if (mode == 1) hipLaunchKernelGGL((k<1, false>), dim3(256), dim3(T), LDS_BYTES, 0, in, out, n_tiles, work, 1.0001f);
                else if (mode == 2) hipLaunchKernelGGL((k<2, false>), dim3(256), dim3(T), LDS_BYTES, 0, in, out, n_tiles, work, 1.0001f);
                else if (mode == 3) hipLaunchKernelGGL((k<0, true>), dim3(256), dim3(T), LDS_BYTES, 0, in, out, n_tiles, work, 1.0001f);
                else if (mode == 4) hipLaunchKernelGGL((k<1, true>), dim3(256), dim3(T), LDS_BYTES, 0, in, out, n_tiles, work, 1.0001f);
                else hipLaunchKernelGGL((k<2, true>), dim3(256), dim3(T), LDS_BYTES, 0, in, out, n_tiles, work, 1.0001f);
            };
            launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 3; ++r) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[mode], e0, e1));
            ms[mode] /= 3;
        }
        printf("%8d %12.3f %12.3f %12.3f | %12.3f %12.3f %12.3f\n", work, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5]);
    }
    return 0;
}
