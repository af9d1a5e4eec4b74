// Round 4, VERDICT item 1: the middle of the CSA chain (azimuth step B -> fused range -> inverse azimuth step A) as ONE persistent
// launch whose row groups stay in the 256 MiB Infinity Cache between the three stages - traffic-only prototype with the kill
// criterion "beats three full-image copies by >= 15 %".
//   image   16384 rows x 128 KiB (2 GiB); group g = rows {g + 128 m} (16 MiB), the row set the three real launches share;
//           SG consecutive groups form one slab (the unit a stage is finished on before the next stage may touch it)
//   stages  0: IN[row] -> W[row]      1: W[row] -> W[row] (in place)      2: W[row] -> OUT[row]     (each adds 1: results checked)
//   jobs    one 128 KiB row per job and 512-thread workgroup (16 x float4 per lane in flight), one persistent workgroup per CU or two
//   order   tickets from one atomic counter; time step t hands out stage 0 of slab t, stage 1 of slab t - D, stage 2 of slab t - 2D;
//           a stage-s job waits until done[s-1][slab] says every row of its slab has passed stage s-1 - all its tickets are older,
//           so the jobs it waits for are running or finished: no co-residency assumption; every spin has a watchdog
//   memory  producer and consumer of a row sit on different XCDs (separate L2s): the producer's last wave writes its L2 back
//           (release fence at agent scope) before it raises the counter, the consumer invalidates (acquire fence) after it has seen it
// baseline: the same row-copy jobs as three separate full-image launches (IN -> W, W -> W, W -> OUT).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mallpipe.hip -o tools/mallpipe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 16384, NG = 128, RPG = ROWS / NG;         // 128 groups of 128 rows
constexpr size_t ROW_BYTES = 128 * 1024, ROW_F4 = ROW_BYTES / 16;

// the same row copy with agent-scope (sc1) accesses on the work image: a store is written through to memory and a load is not
// served from a stale line of this XCD's caches, so no cache-wide write-back / invalidate is needed around the counters
template <bool SC1_IN, bool SC1_OUT> __device__ __forceinline__ void copy_row_sc1(const f4* src, f4* dst) {
    f4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const f4* p = src + threadIdx.x + 512 * k;
        if (SC1_IN) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[k]) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const f4 o = v[k] + 1.0f;
        f4* p = dst + threadIdx.x + 512 * k;
        if (SC1_OUT) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(o) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(o) : "memory");
    }
}

template <bool NT_IN, bool NT_OUT> __device__ __forceinline__ void copy_row(const f4* __restrict__ src, f4* __restrict__ dst) {
    f4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = NT_IN ? __builtin_nontemporal_load(src + threadIdx.x + 512 * k) : src[threadIdx.x + 512 * k];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const f4 o = v[k] + 1.0f;
        if (NT_OUT) __builtin_nontemporal_store(o, dst + threadIdx.x + 512 * k); else dst[threadIdx.x + 512 * k] = o;
    }
}

// one stage as its own launch, persistent over rows (the three-launch baseline)
template <bool NT_IN, bool NT_OUT> __global__ __launch_bounds__(512) void stage_kernel(const f4* in, f4* out) {
    for (int row = blockIdx.x; row < ROWS; row += gridDim.x) copy_row<NT_IN, NT_OUT>(in + (size_t)row * ROW_F4, out + (size_t)row * ROW_F4);
}

struct Ctl { unsigned ticket; unsigned err; unsigned pad[30]; unsigned done[3][NG]; };

template <int MODE> __global__ __launch_bounds__(512) void pipe_kernel(const f4* in, f4* w, f4* out, Ctl* ctl, int sg, int d) {
    __shared__ int job[3];
    const int nslab = NG / sg, jobs_per_entry = sg * RPG;
    const unsigned total = (unsigned)(nslab + 2 * d) * 3u * (unsigned)jobs_per_entry;
    for (;;) {
        if (threadIdx.x == 0) {
            const unsigned k = atomicAdd(&ctl->ticket, 1u);
            int stage = -1, slab = 0, j = 0;
            if (k < total) {
                const unsigned e = k / jobs_per_entry;
                j = (int)(k % jobs_per_entry);
                const int t = (int)(e / 3u);
                stage = (int)(e % 3u);
                slab = t - stage * d;
                if (slab < 0 || slab >= nslab) stage = 3;                       // void ticket at the ends of the pipeline
                else if (stage > 0) {
                    unsigned polls = 0;
                    while (__hip_atomic_load(&ctl->done[stage - 1][slab], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)jobs_per_entry) {
                        __builtin_amdgcn_s_sleep(16);
                        if (++polls > (1u << 16) || __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(&ctl->err, 1u); break; }
                    }
                    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the rows other XCDs wrote are read from memory, not from stale lines
                }
            }
            job[0] = stage; job[1] = slab; job[2] = j;
        }
        __syncthreads();
        const int stage = job[0], slab = job[1], j = job[2];
        __syncthreads();
        if (stage < 0) return;
        if (stage == 3) continue;
        const int g = slab * sg + j / RPG, m = j % RPG;
        const size_t off = (size_t)(g + NG * m) * ROW_F4;
        if (MODE == 0) {
            if (stage == 0) copy_row<true, false>(in + off, w + off);
            else if (stage == 1) copy_row<false, false>(w + off, w + off);
            else copy_row<false, true>(w + off, out + off);
        } else {
            if (stage == 0) copy_row_sc1<false, true>(in + off, w + off);
            else if (stage == 1) copy_row_sc1<true, true>(w + off, w + off);
            else copy_row_sc1<true, false>(w + off, out + off);
        }
        __builtin_amdgcn_s_waitcnt(0);                                           // this wave's stores have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0 && stage < 2) {
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // this XCD's L2 written back before the counter moves
            atomicAdd(&ctl->done[stage][slab], 1u);
        }
    }
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
    const size_t IMG = (size_t)ROWS * ROW_BYTES;
    float *in, *w, *out;
    Ctl* ctl;
    CK(hipMalloc(&in, IMG)); CK(hipMalloc(&w, IMG)); CK(hipMalloc(&out, IMG)); CK(hipMalloc(&ctl, sizeof(Ctl)));
    std::vector<float> h((size_t)1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 1000);
    for (size_t off = 0; off < IMG; off += h.size() * 4) CK(hipMemcpy((char*)in + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    auto check = [&](const char* what) {
        std::vector<float> a(4096), b(4096);
        bool ok = true;
        for (size_t row : {(size_t)0, (size_t)129, (size_t)8191, (size_t)16383}) {
            CK(hipMemcpy(a.data(), (char*)in + row * ROW_BYTES + 8192, 4096 * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), (char*)out + row * ROW_BYTES + 8192, 4096 * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < 4096; ++i) if (b[i] != a[i] + 3.0f) { ok = false; break; }
        }
        printf("   %s: results %s\n", what, ok ? "ok (out = in + 3)" : "WRONG");
        return ok;
    };
    for (int wgs : {256, 512}) {
        CK(hipMemset(out, 0, IMG));
        const float t3 = time_ms([&] {
            hipLaunchKernelGGL((stage_kernel<true, true>), dim3(wgs), dim3(512), 0, 0, (const f4*)in, (f4*)w);
            hipLaunchKernelGGL((stage_kernel<true, true>), dim3(wgs), dim3(512), 0, 0, (const f4*)w, (f4*)w);
            hipLaunchKernelGGL((stage_kernel<true, true>), dim3(wgs), dim3(512), 0, 0, (const f4*)w, (f4*)out);
        }, 5);
        printf("three full-image launches (row copies, nontemporal), %d workgroups      %7.3f ms  (%.2f TB/s per launch)\n", wgs, t3, 3 * 2.0 * IMG / t3 / 1e9);
        check("three launches");
        for (int mode : {0, 1})
        for (int sg : {1, 2, 4, 8})
            for (int d : {1, 2}) {
                CK(hipMemset(out, 0, IMG));
                const float tp = time_ms([&] {
                    (void)hipMemsetAsync(ctl, 0, sizeof(Ctl), 0);
                    if (mode == 0) hipLaunchKernelGGL(pipe_kernel<0>, dim3(wgs), dim3(512), 0, 0, (const f4*)in, (f4*)w, (f4*)out, ctl, sg, d);
                    else hipLaunchKernelGGL(pipe_kernel<1>, dim3(wgs), dim3(512), 0, 0, (const f4*)in, (f4*)w, (f4*)out, ctl, sg, d);
                }, 5);
                Ctl hc;
                CK(hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
                printf("one persistent launch (%s), %d workgroups, slab %3d MiB, stage lag %d slab(s)    %7.3f ms  = %.3f x three launches  watchdog %u\n",
                       mode == 0 ? "release / acquire fences" : "sc1 accesses, no fences ", wgs, sg * 16, d, tp, tp / t3, hc.err);
                check("pipeline");
                fflush(stdout);
            }
    }
    return 0;
}
