// Access-pattern ceilings on MI355X for the sarx kernels (no arithmetic):
//   flat copy, azimuth-tile copies (step A: strided rows in place; step B: contiguous rows in,
//   strided rows out) at 8 or 16 bytes per lane, and a line copy shaped like the range pass.
// build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o gpurun_out/membench ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void flat_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) out[i] = in[i];
}
typedef float v4f __attribute__((ext_vector_type(4)));
// U float4 in flight per lane (the one-per-iteration loop above leaves the memory system under-subscribed: round-2 verdict):
// a workgroup copies U consecutive blocks of 256 float4 per step, all U loads issued before the first store
template <int U, int NT> __global__ __launch_bounds__(256) void flat_copy_u(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    const size_t step = (size_t)gridDim.x * 256 * U;
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += step) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const v4f* p = reinterpret_cast<const v4f*>(&in[base + (size_t)u * 256]);
            v[u] = (NT & 1) ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v4f* p = reinterpret_cast<v4f*>(&out[base + (size_t)u * 256]);
            if (NT & 2) __builtin_nontemporal_store(v[u], p); else *p = v[u];
        }
    }
}
// NT: 1 = nontemporal loads, 2 = nontemporal stores, 3 = both
template <int NT> __global__ __launch_bounds__(256) void flat_copy_nt(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const v4f* pi = reinterpret_cast<const v4f*>(&in[i]);
        v4f* po = reinterpret_cast<v4f*>(&out[i]);
        v4f v = (NT & 1) ? __builtin_nontemporal_load(pi) : *pi;
        if (NT & 2) __builtin_nontemporal_store(v, po); else *po = v;
    }
}
template <int T, int PTS, int NT>
__global__ __launch_bounds__(T) void line_copy_nt(const float2* __restrict__ in, float2* __restrict__ out, int n) {
    extern __shared__ char smem[];
    const size_t row = blockIdx.x;
    const int t = threadIdx.x;
    v4f v[PTS / 2];
#pragma unroll
    for (int i = 0; i < PTS / 2; ++i) {
        const v4f* p = reinterpret_cast<const v4f*>(&in[row * n + 2 * t + i * 2 * T]);
        v[i] = (NT & 1) ? __builtin_nontemporal_load(p) : *p;
    }
    if (n < 0) smem[t] = 1;
#pragma unroll
    for (int i = 0; i < PTS / 2; ++i) {
        v4f* p = reinterpret_cast<v4f*>(&out[row * n + 2 * t + i * 2 * T]);
        if (NT & 2) __builtin_nontemporal_store(v[i], p); else *p = v[i];
    }
}

// tile: 128 rows x W cols of float2; VEC = float2 per lane per access (1 -> 8 B, 2 -> 16 B)
// thread (c, t), t in [0,8): rows m = t + 8*i, i < 16.
template <int W, int VEC>
__global__ __launch_bounds__(8 * W / VEC) void tile_copy(const float2* __restrict__ in, float2* __restrict__ out, int n_rg,
                                                         int in_q, int in_m, int out_q, int out_m) {
    constexpr int LANES = W / VEC;
    const int c = (threadIdx.x % LANES) * VEC, t = threadIdx.x / LANES;
    const int col = blockIdx.x * W + c, q = blockIdx.y;
    if constexpr (VEC == 1) {
        float2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = in[((size_t)q * in_q + (size_t)(t + 8 * i) * in_m) * n_rg + col];
#pragma unroll
        for (int i = 0; i < 16; ++i) out[((size_t)q * out_q + (size_t)(t + 8 * i) * out_m) * n_rg + col] = v[i];
    } else {
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float4*>(&in[((size_t)q * in_q + (size_t)(t + 8 * i) * in_m) * n_rg + col]);
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(&out[((size_t)q * out_q + (size_t)(t + 8 * i) * out_m) * n_rg + col]) = v[i];
    }
}

// the same tile copy with nontemporal loads (NT & 1) and / or stores (NT & 2), 8 B per lane
template <int W, int NT>
__global__ __launch_bounds__(8 * W) void tile_copy_nt(const float2* __restrict__ in, float2* __restrict__ out, int n_rg,
                                                      int in_q, int in_m, int out_q, int out_m) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int c = threadIdx.x % W, t = threadIdx.x / W;
    const int col = blockIdx.x * W + c, q = blockIdx.y;
    v2f v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const v2f* p = reinterpret_cast<const v2f*>(&in[((size_t)q * in_q + (size_t)(t + 8 * i) * in_m) * n_rg + col]);
        v[i] = (NT & 1) ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v2f* p = reinterpret_cast<v2f*>(&out[((size_t)q * out_q + (size_t)(t + 8 * i) * out_m) * n_rg + col]);
        if (NT & 2) __builtin_nontemporal_store(v[i], p); else *p = v[i];
    }
}

// line copy shaped like the range pass: one 16384-sample line per workgroup of T threads,
// each thread PTS samples at stride T (VEC float2 per access), LDS bytes reserved to set residency
template <int T, int PTS, int VEC>
__global__ __launch_bounds__(T) void line_copy(const float2* __restrict__ in, float2* __restrict__ out, int n) {
    extern __shared__ char smem[];
    const size_t row = blockIdx.x;
    const int t = threadIdx.x;
    if constexpr (VEC == 1) {
        float2 v[PTS];
#pragma unroll
        for (int i = 0; i < PTS; ++i) v[i] = in[row * n + t + i * T];
        if (n < 0) smem[t] = 1;
#pragma unroll
        for (int i = 0; i < PTS; ++i) out[row * n + t + i * T] = v[i];
    } else {
        float4 v[PTS / 2];
#pragma unroll
        for (int i = 0; i < PTS / 2; ++i) v[i] = *reinterpret_cast<const float4*>(&in[row * n + 2 * t + i * 2 * T]);
        if (n < 0) smem[t] = 1;
#pragma unroll
        for (int i = 0; i < PTS / 2; ++i) *reinterpret_cast<float4*>(&out[row * n + 2 * t + i * 2 * T]) = v[i];
    }
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
    const int n = 16384;
    const size_t elems = (size_t)n * n;
    float2 *in, *out;
    CK(hipMalloc(&in, elems * 8)); CK(hipMalloc(&out, elems * 8));
    CK(hipMemset(in, 1, elems * 8)); CK(hipMemset(out, 0, elems * 8));
    const double gb = 16.0 * elems / 1e9;
    auto rep = [&](const char* name, float ms) { printf("%-44s %7.3f ms  %7.1f GB/s\n", name, ms, gb / (ms * 1e-3)); };

    for (int blocks : {2048, 8192, 65536})
        rep(blocks == 2048 ? "flat copy float4 grid 2048" : blocks == 8192 ? "flat copy float4 grid 8192" : "flat copy float4 grid 65536",
            time_ms([&] { hipLaunchKernelGGL(flat_copy, dim3(blocks), dim3(256), 0, 0, (const float4*)in, (float4*)out, elems / 2); }));

    {
        char name[96];
        auto run_u = [&](auto kern, int u, int nt, int blocks) {
            snprintf(name, sizeof name, "flat copy %d x float4 in flight%s grid %d", u, nt == 3 ? " nt" : "", blocks);
            rep(name, time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (const float4*)in, (float4*)out, elems / 2); }));
        };
        for (int blocks : {2048, 4096, 16384}) {
            run_u(flat_copy_u<4, 0>, 4, 0, blocks);
            run_u(flat_copy_u<8, 0>, 8, 0, blocks);
            run_u(flat_copy_u<16, 0>, 16, 0, blocks);
            run_u(flat_copy_u<4, 3>, 4, 3, blocks);
            run_u(flat_copy_u<8, 3>, 8, 3, blocks);
            run_u(flat_copy_u<16, 3>, 16, 3, blocks);
        }
        run_u(flat_copy_u<8, 3>, 8, 3, 131072);      // one step per workgroup
        run_u(flat_copy_u<8, 0>, 8, 0, 131072);
    }
    rep("flat copy nt loads  grid 65536", time_ms([&] { hipLaunchKernelGGL((flat_copy_nt<1>), dim3(65536), dim3(256), 0, 0, (const float4*)in, (float4*)out, elems / 2); }));
    rep("flat copy nt stores grid 65536", time_ms([&] { hipLaunchKernelGGL((flat_copy_nt<2>), dim3(65536), dim3(256), 0, 0, (const float4*)in, (float4*)out, elems / 2); }));
    rep("flat copy nt both   grid 65536", time_ms([&] { hipLaunchKernelGGL((flat_copy_nt<3>), dim3(65536), dim3(256), 0, 0, (const float4*)in, (float4*)out, elems / 2); }));
    rep("flat copy nt both   grid 2048", time_ms([&] { hipLaunchKernelGGL((flat_copy_nt<3>), dim3(2048), dim3(256), 0, 0, (const float4*)in, (float4*)out, elems / 2); }));
    CK(hipFuncSetAttribute((const void*)line_copy_nt<512, 32, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 69632));
    CK(hipFuncSetAttribute((const void*)line_copy_nt<512, 32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 69632));
    CK(hipFuncSetAttribute((const void*)line_copy_nt<512, 32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 69632));
    CK(hipFuncSetAttribute((const void*)line_copy_nt<512, 32, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 69632));
    rep("line copy 512thr 2/CU plain", time_ms([&] { hipLaunchKernelGGL((line_copy_nt<512, 32, 0>), dim3(n), dim3(512), 69632, 0, in, out, n); }));
    rep("line copy 512thr 2/CU nt loads", time_ms([&] { hipLaunchKernelGGL((line_copy_nt<512, 32, 1>), dim3(n), dim3(512), 69632, 0, in, out, n); }));
    rep("line copy 512thr 2/CU nt stores", time_ms([&] { hipLaunchKernelGGL((line_copy_nt<512, 32, 2>), dim3(n), dim3(512), 69632, 0, in, out, n); }));
    rep("line copy 512thr 2/CU nt both", time_ms([&] { hipLaunchKernelGGL((line_copy_nt<512, 32, 3>), dim3(n), dim3(512), 69632, 0, in, out, n); }));
    const int S = 128, RA = 128;
    // step A: rows q + m*S in place (q < S)
    rep("az step A  W=32  8B/lane", time_ms([&] { hipLaunchKernelGGL((tile_copy<32, 1>), dim3(n / 32, S), dim3(256), 0, 0, in, out, n, 1, S, 1, S); }));
    rep("az step A  W=64  8B/lane", time_ms([&] { hipLaunchKernelGGL((tile_copy<64, 1>), dim3(n / 64, S), dim3(512), 0, 0, in, out, n, 1, S, 1, S); }));
    rep("az step A  W=64 16B/lane", time_ms([&] { hipLaunchKernelGGL((tile_copy<64, 2>), dim3(n / 64, S), dim3(256), 0, 0, in, out, n, 1, S, 1, S); }));
    rep("az step A  W=128 16B/lane", time_ms([&] { hipLaunchKernelGGL((tile_copy<128, 2>), dim3(n / 128, S), dim3(512), 0, 0, in, out, n, 1, S, 1, S); }));
    // step B: rows q*S + m in, rows q + m*RA out (q < RA)
    rep("az step B  W=32  8B/lane", time_ms([&] { hipLaunchKernelGGL((tile_copy<32, 1>), dim3(n / 32, RA), dim3(256), 0, 0, in, out, n, S, 1, 1, RA); }));
    rep("az step B  W=64 16B/lane", time_ms([&] { hipLaunchKernelGGL((tile_copy<64, 2>), dim3(n / 64, RA), dim3(256), 0, 0, in, out, n, S, 1, 1, RA); }));
    rep("az step B  W=128 16B/lane", time_ms([&] { hipLaunchKernelGGL((tile_copy<128, 2>), dim3(n / 128, RA), dim3(512), 0, 0, in, out, n, S, 1, 1, RA); }));

    rep("az step A  W=32  8B nt stores", time_ms([&] { hipLaunchKernelGGL((tile_copy_nt<32, 2>), dim3(n / 32, S), dim3(256), 0, 0, in, out, n, 1, S, 1, S); }));
    rep("az step A  W=32  8B nt both", time_ms([&] { hipLaunchKernelGGL((tile_copy_nt<32, 3>), dim3(n / 32, S), dim3(256), 0, 0, in, out, n, 1, S, 1, S); }));
    rep("az step B  W=32  8B nt stores", time_ms([&] { hipLaunchKernelGGL((tile_copy_nt<32, 2>), dim3(n / 32, RA), dim3(256), 0, 0, in, out, n, S, 1, 1, RA); }));
    rep("az step B  W=32  8B nt both", time_ms([&] { hipLaunchKernelGGL((tile_copy_nt<32, 3>), dim3(n / 32, RA), dim3(256), 0, 0, in, out, n, S, 1, 1, RA); }));
    rep("az step B  W=32  8B plain (again)", time_ms([&] { hipLaunchKernelGGL((tile_copy_nt<32, 0>), dim3(n / 32, RA), dim3(256), 0, 0, in, out, n, S, 1, 1, RA); }));
    // range-line copies at different residency
    auto set_lds = [&](const void* k, int bytes) { CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); };
    set_lds((const void*)line_copy<1024, 16, 1>, 139264);
    rep("line copy 1024thr 16pts 8B  136KiB LDS (1/CU)", time_ms([&] { hipLaunchKernelGGL((line_copy<1024, 16, 1>), dim3(n), dim3(1024), 139264, 0, in, out, n); }));
    set_lds((const void*)line_copy<512, 32, 2>, 69632);
    rep("line copy 512thr 32pts 16B  68KiB LDS (2/CU)", time_ms([&] { hipLaunchKernelGGL((line_copy<512, 32, 2>), dim3(n), dim3(512), 69632, 0, in, out, n); }));
    rep("line copy 512thr 32pts 16B  32KiB LDS (4/CU)", time_ms([&] { hipLaunchKernelGGL((line_copy<512, 32, 2>), dim3(n), dim3(512), 32768, 0, in, out, n); }));
    rep("line copy 256thr 64pts 16B  32KiB LDS", time_ms([&] { hipLaunchKernelGGL((line_copy<256, 64, 2>), dim3(n), dim3(256), 32768, 0, in, out, n); }));
    rep("line copy 1024thr 16pts 8B  no LDS", time_ms([&] { hipLaunchKernelGGL((line_copy<1024, 16, 1>), dim3(n), dim3(1024), 0, 0, in, out, n); }));
    return 0;
}
