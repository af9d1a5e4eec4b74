// Can the two steps of a four-step azimuth transform share an XCD's 4 MiB L2 instead of making two HBM round trips?
// Access-pattern experiment (no FFT arithmetic, only a tile-wide dependency so that results prove the ordering):
//   step A, tile (panel p, q):  rows {q + 128 j}, W columns of panel p     read -> tile function -> write in place
//   step B, tile (panel p, k1): rows {128 k1 + i}, the same W columns      read -> tile function -> write in place
// mode 0: two ordinary launches (what az_tile_kernel does today: four HBM transfers of the image per transform).
// mode 1: one persistent launch.  Every workgroup reads the XCD it really runs on (HW_REG_XCC_ID) and serves that XCD's queue:
//         panel p belongs to XCD p % 8; jobs are handed out in a fixed order in which step A runs LOOKAHEAD panels ahead of step B.  A panel (16384 x W x 8 B = 2 MiB at W = 16) is written by A and
//         re-read by B through the same L2; B overwrites the lines A left dirty.  Ordering uses L2 atomics of the XCD and an L1
//         invalidate only (no L2 write-back / invalidate): correct because producer and consumer sit on the same XCD by construction.
//         Every spin is bounded (watchdog -> error flag -> exit), so the grid always drains.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/l2panel.hip -o tools/l2panel.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int R = 128;                 // rows per tile (both steps), 16384 = 128 * 128
constexpr int NXCD = 8;
constexpr int MAXP = 512;              // panels per XCD at most

struct Queue {                         // one per XCD, 4 KiB apart
    unsigned next_a, next_b, pad[30];
    unsigned done[MAXP];
    unsigned pad2[1024 - 32 - MAXP];
};

template <int W> struct Tile {
    static constexpr int LANES_PER_ROW = W * 8 / 16;            // float4 = two samples per lane
    static constexpr int ROWS_PER_ITER = 256 / LANES_PER_ROW;
    static constexpr int ITERS = R / ROWS_PER_ITER;
};

template <int W, bool NT> __device__ inline void do_tile(const float2* xin, size_t ldin, float2* x, size_t ld, int col0, int row0, int row_stride, float* red) {
    using T = Tile<W>;
    const int lane = threadIdx.x % T::LANES_PER_ROW, rsub = threadIdx.x / T::LANES_PER_ROW;
    float4 v[T::ITERS];
    float s = 0;
#pragma unroll
    for (int i = 0; i < T::ITERS; ++i) {
        const int r = row0 + (i * T::ROWS_PER_ITER + rsub) * row_stride;
        const f4* p = (const f4*)(xin + (size_t)r * ldin + col0) + lane;
        const f4 t = NT ? __builtin_nontemporal_load(p) : *p;
        v[i] = make_float4(t.x, t.y, t.z, t.w);
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    // tile-wide dependency: every output needs every input of the tile
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x % 64 == 0) red[threadIdx.x / 64] = s;
    __syncthreads();
    const float S = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < T::ITERS; ++i) {
        const int r = row0 + (i * T::ROWS_PER_ITER + rsub) * row_stride;
        float4* p = (float4*)(x + (size_t)r * ld + col0) + lane;
        float4 o = make_float4(v[i].x * 0.5f + S * 1e-6f, v[i].y * 0.5f - S * 1e-6f, v[i].z * 0.5f + S * 2e-6f, v[i].w * 0.5f);
        *p = o;
    }
}

struct Bufs { const float2* src; size_t ld_src; float2* work; size_t ld; float2* dst; size_t ld_dst; };   // A: src -> work, B: work -> dst

template <int W, bool NT> __global__ __launch_bounds__(256) void static_kernel(Bufs b, int step) {
    __shared__ float red[4];
    const int tile = blockIdx.x, p = tile / R, t = tile % R;
    if (step == 0) do_tile<W, NT>(b.src, b.ld_src, b.work, b.ld, p * W, t, R, red);
    else do_tile<W, false>(b.work, b.ld, b.dst, b.ld_dst, p * W, t * R, 1, red);
}

// agent-scope relaxed load: global_load ... sc1, served by the XCD's L2 (an add of 0 would be folded into a plain load that may hit a stale L1 line)
__device__ inline unsigned rmw_read(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One ticket stream per XCD, claimed with a single fetch-add per job (no compare-and-swap loops: 128 workgroups retrying on one
// address collapse).  Stream for lookahead L:  A(0) ... A(L), then B(0) A(L+1) B(1) A(L+2) ...; each entry is 128 tiles.  An A job never
// waits; a B(p) job waits until done[p] == 128, and every A(p) ticket precedes every B(p) ticket, so the tiles it waits for are
// already running: no deadlock.  A watchdog sets err and lets every workgroup run out its tickets without waiting.
template <int W, bool NT> __global__ __launch_bounds__(256) void fused_kernel(Bufs b, Queue* queues, int panels_per_xcd, int lookahead,
                                                                          int* err, unsigned* xcd_hist, int chunk) {
    __shared__ float red[4];
    __shared__ int job[3];                       // kind (0 exit, 1 A, 2 B, 3 void), panel index within the XCD, tile
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    Queue* Q = queues + xcc;
    const unsigned P = (unsigned)panels_per_xcd, L = (unsigned)lookahead;
    const unsigned head = (L + 1) * R, total = head + 2 * R * P;
    if (threadIdx.x == 0) atomicAdd(&xcd_hist[xcc * 32], 1u);
    for (;;) {
        if (threadIdx.x == 0) {
            int kind, pi, t;
            const unsigned k = atomicAdd(&Q->next_a, 1u) * (unsigned)chunk;        // one ticket = `chunk` consecutive tiles of one entry
            if (k >= total) { kind = 0; pi = t = 0; }
            else if (k < head) { kind = 1; pi = k / R; t = k % R; }
            else {
                const unsigned kk = k - head, pair = kk / (2 * R), within = kk % (2 * R);
                if (within < R) { kind = 2; pi = pair; t = within; }
                else { kind = pair + L + 1 < P ? 1 : 3; pi = pair + L + 1; t = within - R; }
            }
            if (kind == 1 && pi >= (int)P) kind = 3;
            if (kind == 2) {
                int polls = 0;
                while (rmw_read(&Q->done[pi]) != R) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++polls > (1 << 13) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(err, 1); break; }
                }
            }
            job[0] = kind; job[1] = pi; job[2] = t;
        }
        __syncthreads();
        const int kind = job[0], pi = job[1], t = job[2];
        __syncthreads();
        if (kind == 0) return;
        if (kind == 3) continue;
        const int p = (int)xcc + NXCD * pi;
        if (kind == 1) {
            for (int c = 0; c < chunk; ++c) do_tile<W, NT>(b.src, b.ld_src, b.work, b.ld, p * W, t + c, R, red);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // the tiles' stores have reached L2
            __syncthreads();
            if (threadIdx.x == 0) atomicAdd(&Q->done[pi], (unsigned)chunk);
        } else {
            asm volatile("buffer_inv sc0" ::: "memory");                              // drop this CU's L1 copies of the panel's old contents
            for (int c = 0; c < chunk; ++c) do_tile<W, false>(b.work, b.ld, b.dst, b.ld_dst, p * W, (t + c) * R, 1, red);
        }
    }
}

static int g_mode = 0;                 // 0: in place on the padded work image; 1: A reads a dense source; 2: B writes a dense destination
static float2 *g_dense_src, *g_dense_dst;

template <int W, bool NT> static void run(float2* d, float2* d0, size_t n, size_t ld, int lookahead, int wgs_per_cu, int chunk, Queue* dq, int* derr, unsigned* dhist,
                                          std::vector<float2>& ref, bool first) {
    const size_t bytes = n * ld * sizeof(float2), moved = n * n * sizeof(float2);
    const int panels = (int)(n / W);
    Bufs b{d, ld, d, ld, d, ld};
    if (g_mode == 1) { b.src = g_dense_src; b.ld_src = n; }
    if (g_mode == 2) { b.dst = g_dense_dst; b.ld_dst = n; }
    const float transfers = 4.0f;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms0 = 0, ms1 = 0;
    const int REPS = 3;
    auto sample = [&](std::vector<float2>& v) {
        for (int i = 0; i < 64; ++i) CK(hipMemcpy(v.data() + (size_t)i * n, b.dst + (size_t)(i * 251 + 3) * b.ld_dst, n * sizeof(float2), hipMemcpyDeviceToHost));
    };
    if (first) {
        for (int rep = 0; rep < REPS + 1; ++rep) {
            CK(hipMemcpy(d, d0, bytes, hipMemcpyDeviceToDevice));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((static_kernel<W, NT>), dim3(panels * R), dim3(256), 0, 0, b, 0);
            hipLaunchKernelGGL((static_kernel<W, NT>), dim3(panels * R), dim3(256), 0, 0, b, 1);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) ms0 += ms / REPS;
        }
        sample(ref);
        printf("mode %d W=%2d nt=%d  two launches                                   %.3f ms  (%.2f TB/s over 4 image transfers)\n", g_mode, W, (int)NT, ms0,
               transfers * moved / ms0 * 1e-9);
    }
    unsigned hist[NXCD * 32];
    CK(hipMemset(derr, 0, 4));
    for (int rep = 0; rep < REPS + 1; ++rep) {
        CK(hipMemcpy(d, d0, bytes, hipMemcpyDeviceToDevice));
        CK(hipMemset(dq, 0, sizeof(Queue) * NXCD));
        CK(hipMemset(dhist, 0, sizeof(unsigned) * NXCD * 32));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((fused_kernel<W, NT>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, b, dq, panels / NXCD, lookahead, derr, dhist, chunk);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) ms1 += ms / REPS;
    }
    int err = 0;
    CK(hipMemcpy(&err, derr, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hist, dhist, sizeof hist, hipMemcpyDeviceToHost));
    std::vector<float2> got(64 * n);
    sample(got);
    const bool same = memcmp(got.data(), ref.data(), got.size() * sizeof(float2)) == 0;
    unsigned lo = ~0u, hi = 0;
    for (int i = 0; i < NXCD; ++i) { lo = hist[i * 32] < lo ? hist[i * 32] : lo; hi = hist[i * 32] > hi ? hist[i * 32] : hi; }
    printf("mode %d W=%2d nt=%d  one launch, lookahead %d, %d WG/CU, %2d tiles/ticket   %.3f ms  (%.2f TB/s-equivalent)  results %s  watchdog %d  WGs per XCD %u..%u\n",
           g_mode, W, (int)NT, lookahead, wgs_per_cu, chunk, ms1, transfers * moved / ms1 * 1e-9, same ? "identical" : "DIFFER", err, lo, hi);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const size_t n = argc > 1 ? atoi(argv[1]) : 16384;
    if (n != (size_t)R * R) { printf("n must be %d\n", R * R); return 1; }
    const size_t ld = n + (argc > 2 ? atoi(argv[2]) : 0);          // row pitch in samples: a pad spreads a panel's lines over the L2 sets
    const size_t bytes = n * ld * sizeof(float2);
    printf("row pitch %zu samples (%zu B)\n", ld, ld * 8);
    float2 *d, *d0;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&d0, bytes));
    {
        std::vector<float2> h(n * 64);
        for (size_t i = 0; i < h.size(); ++i) h[i] = make_float2((float)((i * 2654435761u) % 1000) * 1e-3f, (float)((i * 40503u) % 777) * 1e-3f);
        CK(hipMemset(d0, 0, bytes));
        for (size_t r = 0; r < n; ++r) CK(hipMemcpy(d0 + r * ld, h.data() + (r % 64) * n, n * sizeof(float2), hipMemcpyHostToDevice));
    }
    Queue* dq; int* derr; unsigned* dhist;
    CK(hipMalloc(&dq, sizeof(Queue) * NXCD)); CK(hipMalloc(&derr, 4)); CK(hipMalloc(&dhist, sizeof(unsigned) * NXCD * 32));
    CK(hipMemset(derr, 0, 4));
    std::vector<float2> ref(64 * n);
    CK(hipMalloc(&g_dense_src, n * n * sizeof(float2))); CK(hipMalloc(&g_dense_dst, n * n * sizeof(float2)));
    CK(hipMemcpy2D(g_dense_src, n * 8, d0, ld * 8, n * 8, n, hipMemcpyDeviceToDevice));
    bool first = true;
    if (argc > 3 && !strcmp(argv[3], "modes")) {
        for (int mode : {0, 1, 2}) {
            g_mode = mode; first = true;
            for (int wg : {2, 3}) for (int chunk : {4, 8, 16}) for (int la : {1, 2}) { run<32, false>(d, d0, n, ld, la, wg, chunk, dq, derr, dhist, ref, first); first = false; }
        }
        return 0;
    }
    if (argc > 3) {            // fine sweep around the good point
        for (int wg : {1, 2, 3}) for (int chunk : {1, 2, 4, 8}) for (int la : {1, 2}) { run<16, false>(d, d0, n, ld, la, wg, chunk, dq, derr, dhist, ref, first); first = false; }
        for (int chunk : {2, 4}) for (int la : {1, 2}) run<16, true>(d, d0, n, ld, la, 2, chunk, dq, derr, dhist, ref, false);
        first = true;
        for (int wg : {1, 2}) for (int chunk : {1, 2, 4}) for (int la : {0, 1}) { run<32, false>(d, d0, n, ld, la, wg, chunk, dq, derr, dhist, ref, first); first = false; }
        return 0;
    }
    for (int la : {0, 1, 2}) { run<16, false>(d, d0, n, ld, la, 2, 4, dq, derr, dhist, ref, first); first = false; }
    for (int la : {0, 1}) run<16, false>(d, d0, n, ld, la, 4, 16, dq, derr, dhist, ref, false);
    first = true;
    for (int la : {0, 1}) { run<32, false>(d, d0, n, ld, la, 4, 16, dq, derr, dhist, ref, first); first = false; }
    first = true;
    for (int la : {0, 1, 3}) { run<8, false>(d, d0, n, ld, la, 4, 32, dq, derr, dhist, ref, first); first = false; }
    return 0;
}
