// What shader clock does the part actually run at while each of our kernels is executing?
// One resident probe wave (own stream, launched first, < 32 VGPRs so it fits beside the 234-VGPR waves of the fused range
// kernel) reads s_memtime (shader-clock cycles) and s_memrealtime (constant 100 MHz) over WINDOWS windows of WINDOW_US each
// while the workload runs on the sarx ctx stream; cycles / time = the clock the power manager granted in that window.
// Motivation: arithmetic alone 0.83 ms, HBM traffic alone 0.79 ms, together 1.10 ms, and hiding latency does
// not bring them closer — is the vector pipe simply clocked lower when HBM is busy?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/clockprobe.hip -o tools/clockprobe.bin \
//         -Lnis-sar-amtigmti-video_amd/sarx -lsarx -Wl,-rpath,$PWD/nis-sar-amtigmti-video_amd/sarx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <chrono>
#include "sarx.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define SK(x) do { int r = (x); if (r) { printf("sarx error %d line %d: %s\n", r, __LINE__, sarx_last_error(ctx)); exit(1); } } while (0)

constexpr int WINDOWS = 24;                               // windows of argv[2] ms each (default 1; s_memrealtime ticks at 100 MHz)

__global__ __launch_bounds__(64) void probe_kernel(unsigned long long* out, unsigned long long WINDOW_TICKS) {
    if (threadIdx.x != 0) return;
    for (int w = 0; w < WINDOWS; ++w) {
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        unsigned long long r1;
        do { __builtin_amdgcn_s_sleep(32); r1 = __builtin_amdgcn_s_memrealtime(); } while (r1 - r0 < WINDOW_TICKS);
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        out[2 * w] = c1 - c0; out[2 * w + 1] = r1 - r0;
    }
}

// every CU busy with independent fp32 FMAs, nothing else (the vector pipe's own power draw)
__global__ __launch_bounds__(512, 2) void burn_kernel(float* out, float s, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    if (r == 12345.678f) out[0] = r;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16384;
    const int win_ms = argc > 2 ? atoi(argv[2]) : 1;
    sarx_ctx* ctx = nullptr; sarx_plan* plan = nullptr;
    SK(sarx_init(0, &ctx));
    sarx_radar_params p;
    memset(&p, 0, sizeof p);
    p.wavelength_m = 0.0310666; p.pulse_width_s = 20e-6; p.chirp_rate_hz_s = 2.5e13; p.sample_rate_hz = 600e6;
    p.prf_hz = 6000; p.platform_speed_mps = 7000; p.range_ref_m = 700e3; p.t_start_fast_s = 2 * 700e3 / 299792458.0 - 11e-6;
    SK(sarx_csa_plan_create(ctx, n, n, &p, SARX_FUSE_RANGE, &plan));
    void *a, *b;
    SK(sarx_malloc(ctx, (size_t)n * n * 8, &a)); SK(sarx_malloc(ctx, (size_t)n * n * 8, &b));
    CK(hipMemset(a, 0, (size_t)n * n * 8));
    hipStream_t ps;
    CK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    unsigned long long* d_out;
    CK(hipMalloc(&d_out, WINDOWS * 2 * sizeof(unsigned long long)));
    float* d_burn;
    CK(hipMalloc(&d_burn, 4));

    struct Load { const char* name; int pass; };
    const Load loads[] = {{"idle", 0}, {"range fused FFT.Phi2.IFFT.Phi3", SARX_PASS_RG_FUSED_23}, {"range FFT+Phi2 (v2)", SARX_PASS_RG_FFT_PHI2},
                          {"azimuth FFT+Phi1 (2 launches)", SARX_PASS_AZ_FFT_PHI1}, {"device copy 2 GiB", -1}, {"fp32 FMA burn, all CUs", -2}};
    for (const Load& l : loads) {
        // warm up the workload once, then start the probe and keep the ctx stream busy for longer than the probe runs
        auto work = [&](int reps) {
            for (int r = 0; r < reps; ++r) {
                if (l.pass > 0) SK(sarx_csa_pass(plan, l.pass, a, b));
                else if (l.pass == -1) CK(hipMemcpyAsync(b, a, (size_t)n * n * 8, hipMemcpyDeviceToDevice, 0));
                else if (l.pass == -2) hipLaunchKernelGGL(burn_kernel, dim3(512), dim3(512), 0, 0, d_burn, 1.0001f, 20000);
            }
        };
        work(2);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, ps, d_out, (unsigned long long)win_ms * 100000ull);
        int reps = l.pass == 0 ? 0 : 40 * win_ms;
        if (l.pass == -2) reps = 30 * win_ms;
        const auto t0 = std::chrono::steady_clock::now();
        work(reps);
        if (l.pass > 0) SK(sarx_sync(ctx)); else CK(hipStreamSynchronize(0));
        const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CK(hipDeviceSynchronize());
        unsigned long long h[WINDOWS * 2];
        CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
        std::vector<double> mhz;
        const int covered = reps ? std::min(WINDOWS, (int)((ms - 1.0f) / win_ms)) : WINDOWS;      // only windows the workload overlapped
        for (int w = 1; w < covered; ++w) mhz.push_back((double)h[2 * w] / ((double)h[2 * w + 1] / 100.0));
        std::sort(mhz.begin(), mhz.end());
        if (mhz.empty()) { printf("%-34s workload too short (%.2f ms)\n", l.name, ms); continue; }
        printf("%-34s shader clock min %6.0f  median %6.0f  max %6.0f MHz over %2zu windows of %d ms", l.name, mhz.front(), mhz[mhz.size() / 2],
               mhz.back(), mhz.size(), win_ms);
        if (reps) printf("   (%.3f ms per repetition)", ms / reps);
        printf("\n");
    }
    return 0;
}
