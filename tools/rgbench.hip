// Stand-alone check + timing of the 16384-sample range kernels (no Python, no plan): the sixteen-wave permuted-spectrum
// kernels of csrc/range_wp.hip against a host fp64 FFT on a few lines, and their launch times beside range_v2.hip and
// range_fused_wl.hip on the same 16384 x 16384 image.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -ffp-contract=on [-DWP_PREFETCH=.. -DWP_HOIST=.. -DWP_NT=..] tools/rgbench.hip -o tools/rgbench.bin
#ifndef WP_WITH_FUSED
#define WP_WITH_FUSED 1
#endif
#include "../nis-sar-amtigmti-video_amd/csrc/range_wp.hip"   // defines the WP_* defaults it was not given
#include "../nis-sar-amtigmti-video_amd/csrc/range_v2.hip"
#include "../nis-sar-amtigmti-video_amd/csrc/range_fused_wl.hip"

#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

using namespace sarx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float2* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed ^ (unsigned)(i >> 32) * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        unsigned g = h * 747796405u + 2891336453u; g ^= g >> 16;
        p[i] = make_float2((float)(int)h * 4.6566e-10f, (float)(int)g * 4.6566e-10f);
    }
}

typedef std::complex<double> cd;
static void fft_host(std::vector<cd>& a, bool inv) {       // iterative radix-2, fp64
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = 2 * M_PI / (double)len * (inv ? 1 : -1);
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const cd w(cos(ang * (double)k), sin(ang * (double)k));
                const cd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v; a[i + k + len / 2] = u - v;
            }
    }
}
static cd cis_rev_h(double p) { p -= nearbyint(p); return cd(cos(2 * M_PI * p), sin(2 * M_PI * p)); }

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

int main(int argc, char** argv) {
    const int N = 16384;
    int n_az = argc > 1 ? atoi(argv[1]) : 16384;
    const size_t elems = (size_t)n_az * N;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    // argv[2]: the spectrum / output images start this many KiB into their allocations (does the relative placement of the read and the
    // write stream matter?)
    const size_t off_elems = (argc > 2 ? (size_t)atol(argv[2]) : 0) * 128;
    float2 *d_in, *d_spec, *d_out;
    CK(hipMalloc(&d_in, elems * 8)); CK(hipMalloc(&d_spec, elems * 8 + (64 << 20))); CK(hipMalloc(&d_out, elems * 8 + (64 << 20)));
    d_spec += off_elems; d_out += 2 * off_elems;
    printf("output offset %zu KiB; d_in %p d_spec %p d_out %p\n", off_elems / 128, (void*)d_in, (void*)d_spec, (void*)d_out);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, d_in, elems, 12345u);
    CK(hipDeviceSynchronize());

    // per-row phase constants as sarx_csa_plan_create builds them (sar_ati_dcpa_sim_csa.py:244-262) for the reference radar
    const double C = 299792458.0, lam = C / 9.65e9, Kr = 500e6 / 20e-6, fs = 600e6, prf = 6000.0, Vr = 7100.0, Rref = 850e3;
    std::vector<double2> c2(n_az), c3(n_az);
    for (int i = 0; i < n_az; ++i) {
        const int ks = i < n_az / 2 ? i : i - n_az;
        const double fa = ks * prf / n_az, u = lam * fa / (2 * Vr);
        double arg = 1 - u * u; if (arg < 0) arg = 1e-9;
        const double D = sqrt(arg), Cs = 1 / D - 1;
        c2[i] = make_double2(0.5 / (Kr * (1 + Cs)), 2 * Rref * Cs / C);
        c3[i] = make_double2(C * D / lam, -0.5 * Kr * Cs * (1 + Cs));
    }
    double2 *d_c2, *d_c3;
    CK(hipMalloc(&d_c2, n_az * sizeof(double2))); CK(hipMalloc(&d_c3, n_az * sizeof(double2)));
    CK(hipMemcpy(d_c2, c2.data(), n_az * sizeof(double2), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_c3, c3.data(), n_az * sizeof(double2), hipMemcpyHostToDevice));
    std::vector<float2> twh(2 * N);
    for (int n2 = 2; n2 <= N; n2 <<= 1) for (int m = 0; m < n2; ++m) twh[n2 + m] = make_float2((float)cos(-2 * M_PI * m / n2), (float)sin(-2 * M_PI * m / n2));
    float2* d_tw; CK(hipMalloc(&d_tw, twh.size() * 8)); CK(hipMemcpy(d_tw, twh.data(), twh.size() * 8, hipMemcpyHostToDevice));

    RangeArgs a{};
    a.tw = d_tw + N; a.c2 = d_c2; a.c3 = d_c3;
    a.dt = 1.0 / fs; a.df = 1.0 / (N * a.dt); a.t_start = 2 * Rref / C - 0.5 * N / fs; a.t0 = 2 * Rref / C; a.inv_n = 1.0f / N; a.n_az = n_az;

    auto run = [&](const char* what, int impl, int mode, const float2* in, float2* out) {
        RangeArgs b = a; b.in = in; b.out = out;
        hipError_t e = impl == 0 ? launch_range_wp(mode, b, cus, 0) : impl == 1 ? launch_range_pass_v2(N, mode, b, cus, 0) : launch_range_fused_wl(b, cus, 0);
        if (e != hipSuccess) { printf("%s: launch failed: %s\n", what, hipGetErrorString(e)); exit(1); }
    };

    // ---- correctness: rows sampled over the image, fp64 host chain ----
    run("wp fft+phi2", 0, RG_FFT_PHI2, d_in, d_spec);
    run("wp ifft+phi3", 0, RG_IFFT_PHI3, d_spec, d_out);
    CK(hipDeviceSynchronize());
    float2* d_fused; CK(hipMalloc(&d_fused, elems * 8));
    run("wp fused", 0, RG_FUSED, d_in, d_fused);
    CK(hipDeviceSynchronize());
    const int rows[] = {0, 1, n_az / 3, n_az / 2 + 5, n_az - 1, 255, 256, 257};
    double worst_spec = 0, worst_out = 0, worst_fused = 0;
    std::vector<float2> hx(N), hs(N), ho(N), hf(N);
    for (int row : rows) {
        if (row >= n_az) continue;
        CK(hipMemcpy(hx.data(), d_in + (size_t)row * N, N * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), d_spec + (size_t)row * N, N * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ho.data(), d_out + (size_t)row * N, N * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hf.data(), d_fused + (size_t)row * N, N * 8, hipMemcpyDeviceToHost));
        std::vector<cd> x(N);
        for (int i = 0; i < N; ++i) x[i] = cd(hx[i].x, hx[i].y);
        fft_host(x, false);
        double num = 0, den = 0;
        for (int k = 0; k < N; ++k) {
            const int ks = k < N / 2 ? k : k - N;
            const double f = ks * a.df;
            x[k] *= cis_rev_h(f * (c2[row].x * f + c2[row].y));
#if WP_LAYOUT == 1
            const float2 g = hs[((k / 16) / 64) * 1024 + (k % 16) * 64 + (k / 16) % 64];
#else
            const float2 g = hs[(k % 16) * 1024 + k / 16];                // the permuted order
#endif
            num += std::norm(cd(g.x, g.y) - x[k]); den += std::norm(x[k]);
        }
        worst_spec = fmax(worst_spec, sqrt(num / den));
        fft_host(x, true);
        double n1 = 0, n2 = 0, d1 = 0;
        for (int j = 0; j < N; ++j) {
            const double tau = a.t_start + j * a.dt, d0 = tau - a.t0;
            const cd y = x[j] / (double)N * cis_rev_h(c3[row].x * tau + c3[row].y * d0 * d0);
            n1 += std::norm(cd(ho[j].x, ho[j].y) - y); n2 += std::norm(cd(hf[j].x, hf[j].y) - y); d1 += std::norm(y);
        }
        worst_out = fmax(worst_out, sqrt(n1 / d1)); worst_fused = fmax(worst_fused, sqrt(n2 / d1));
    }
    printf("relative L2 vs fp64 host chain, worst of the sampled rows: FFT+Phi2 %.2e, then IFFT+Phi3 %.2e, fused %.2e\n", worst_spec, worst_out, worst_fused);
    const bool ok = worst_spec < 1e-5 && worst_out < 1e-5 && worst_fused < 1e-5;
    printf("%s\n", ok ? "PARITY OK" : "PARITY FAILED");
    CK(hipFree(d_fused));

    // ---- timing ----
    const double gb = 16.0 * elems / 1e9;
    auto rep = [&](const char* name, float ms) { printf("%-44s %7.3f ms  %7.1f GB/s  %5.1f %% of 8 TB/s\n", name, ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / 80.0); fflush(stdout); };
    printf("WP_PREFETCH=%d WP_HOIST=%d WP_NT=%d WP_ABL=%d WP_LAYOUT=%d, %d lines\n", WP_PREFETCH, WP_HOIST, WP_NT, WP_ABL, WP_LAYOUT, n_az);
    for (int rep_i = 0; rep_i < 3; ++rep_i) {
        rep("wp  FFT+Phi2 (permuted out)", time_ms([&] { run("", 0, RG_FFT_PHI2, d_in, d_spec); }));
        rep("wp  IFFT+Phi3 (permuted in)", time_ms([&] { run("", 0, RG_IFFT_PHI3, d_spec, d_out); }));
        rep("wp  FFT+Phi2 then IFFT+Phi3 in place (x2)", time_ms([&] { run("", 0, RG_FFT_PHI2, d_out, d_out); run("", 0, RG_IFFT_PHI3, d_out, d_out); }) / 2);
        rep("wp  FFT+Phi2 d_in -> d_out (other offset)", time_ms([&] { run("", 0, RG_FFT_PHI2, d_in, d_out); }));
        rep("wp  fused", time_ms([&] { run("", 0, RG_FUSED, d_in, d_out); }));
        rep("wp  fused in place", time_ms([&] { run("", 0, RG_FUSED, d_out, d_out); }));
        rep("v2  FFT+Phi2", time_ms([&] { run("", 1, RG_FFT_PHI2, d_in, d_spec); }));
        rep("v2  IFFT+Phi3", time_ms([&] { run("", 1, RG_IFFT_PHI3, d_spec, d_out); }));
        rep("wl  fused (range_fused_wl)", time_ms([&] { run("", 2, RG_FUSED, d_in, d_out); }));
        rep("wl  fused in place", time_ms([&] { run("", 2, RG_FUSED, d_out, d_out); }));
    }
    return ok ? 0 : 1;
}
