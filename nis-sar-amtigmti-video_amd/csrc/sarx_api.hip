// libsarx C ABI (include/sarx.h): context, CSA plan (fp64 migration tables,
// twiddles, scratch), pass orchestration, ATI/DPCA, RCCL all-gather.
#include "../../include/sarx.h"
#include "csa_kernels.h"
#include "general.h"
#include "tdbp.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

using namespace sarx;

static thread_local std::string g_init_error;
static constexpr double C_LIGHT = 299792458.0;     // sar_ati_dcpa_sim_csa.py:211
static constexpr int N_EVENTS = 256;
static constexpr int TW_MAX = 16384;

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    std::string path;      // file the symbols came from
    int version = 0;       // ncclGetVersion of that file
};
static RcclApi g_rccl;
// RCCL must be the build that belongs to the HIP runtime this process runs on: a process that imported torch first runs
// on torch's bundled libamdhip64 (soname libamdhip64.so.7, same as /opt/rocm's, so the loader hands it to libsarx too)
// and must take torch's bundled librccl; a plain C / ctypes process runs on /opt/rocm's runtime and takes /opt/rocm's
// librccl.  So: SARX_RCCL_PATH if set, else librccl from the directory of the loaded HIP runtime, else whatever
// librccl.so.1 is already mapped, else the loader's search path.  The six entry points used are ABI-stable across
// RCCL 2.2x; path and version are reported by sarx_rccl_info so a run states what it gathered with.
static bool load_rccl(std::string& err) {
    if (g_rccl.lib) return true;
    void* h = nullptr;
    std::string tried;
    auto attempt = [&](const std::string& name, int extra) {
        if (h || name.empty()) return;
        h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL | extra);
        if (!h) tried += name + "; ";
    };
    if (const char* e = getenv("SARX_RCCL_PATH")) attempt(e, 0);
    Dl_info di;
    if (!h && dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
        std::string dir(di.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash);
            attempt(dir + "/librccl.so.1", 0);
            attempt(dir + "/librccl.so", 0);
        }
    }
    attempt("librccl.so.1", RTLD_NOLOAD);
    attempt("librccl.so.1", 0);
    attempt("librccl.so", 0);
    if (!h) { err = "dlopen librccl failed (tried " + tried + ")"; return false; }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    g_rccl.GetVersion = (decltype(g_rccl.GetVersion))dlsym(h, "ncclGetVersion");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
        err = "librccl lacks a required symbol";
        dlclose(h);
        return false;
    }
    if (dladdr((void*)g_rccl.GetUniqueId, &di) && di.dli_fname) g_rccl.path = di.dli_fname;
    if (g_rccl.GetVersion) g_rccl.GetVersion(&g_rccl.version);
    g_rccl.lib = h;
    return true;
}

struct sarx_ctx {
    int device = -1;
    int cus = 256;                     // compute units of this device (persistent grids are sized from it)
    hipStream_t stream = nullptr;      // the CURRENT lane's stream: everything is enqueued here
    static constexpr int LANES = 4;
    hipStream_t lane[LANES] = {};      // lane 0 = the stream made by sarx_init; the others on first sarx_select_lane
    hipEvent_t lane_ev[LANES] = {};
    int cur_lane = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev[N_EVENTS] = {};
    bool ev_set[N_EVENTS] = {};
    hipEvent_t comm_fence = nullptr;
    hipEvent_t comm_done = nullptr;
    hipEvent_t comm_mark[4] = {};      // sarx_comm_mark / sarx_comm_wait_mark: "gathers enqueued up to here are finished"
    bool comm_mark_set[4] = {};
    float2* tw_all = nullptr;          // table for size n at offset n: exp(-2 pi i m/n)
    float* ati_part_max_all = nullptr;     // reduction scratch, one set per lane (two frames in flight must not share it)
    double2* ati_part_sum_all = nullptr;
    double* ati_out3_all = nullptr;
    static constexpr int POWER_STRIDE = 2048 + 8;   // per lane: 1024 {sum, max} partials, then the two fp32 noise levels
    double* power_part_all = nullptr;       // sarx_power_stats_dev partials, one set per lane (it used to hipMalloc / hipFree per call)
    float* ati_part_max_() const { return ati_part_max_all + (size_t)cur_lane * 4096; }
    double2* ati_part_sum_() const { return ati_part_sum_all + (size_t)cur_lane * 4096; }
    double* ati_out3_() const { return ati_out3_all + (size_t)cur_lane * 4; }
    // staged host transfers (sarx_memcpy_h2d / _d2h and the *_host entry points, large pageable buffers): COPY_THREADS host
    // threads, each with its own pinned chunk and stream, copy chunk by chunk in parallel with the DMA of the others
    static constexpr int COPY_THREADS = 8;
    static constexpr size_t COPY_CHUNK = (size_t)32 << 20;
    char* pin[COPY_THREADS] = {};
    hipStream_t copy_stream[COPY_THREADS] = {};
    hipEvent_t pin_free[COPY_THREADS] = {};     // "the DMA that last read pinned chunk i has finished"
    int up_streams = COPY_THREADS;     // uploads issue their DMAs on this many of the copy streams (SARX_UP_STREAMS, A/B).  Beside a download in
                                       // flight, 2 GiB each way: 67 ms with eight streams, 75 with two, 76 with one (= one after the other);
                                       // two plain DMAs (page-locked source) 44 ms: the staged upload is bound by the HOST's memory traffic (it
                                       // reads the array, writes the chunk, and the DMA reads the chunk again), not by the stream count
                                       // (profiles/r05_i_duplex.log)
    std::mutex copy_mu;                // the pinned chunks and copy streams are per-ctx state: one staged copy at a time
    // overlapped host transfers (sarx_memcpy_h2d_unordered, sarx_memcpy_d2h_begin / _end): downloads run on their own stream behind an
    // event of the producing lane, uploads into free buffers do not wait for enqueued GPU work - PCIe is full duplex
    static constexpr int DL_SLOTS = 8;
    hipStream_t dl_stream = nullptr, up_stream = nullptr;
    hipEvent_t dl_ready[DL_SLOTS] = {};    // recorded on the producing lane
    hipEvent_t dl_done[DL_SLOTS] = {};     // recorded on dl_stream behind the copy
    bool dl_busy[DL_SLOTS] = {};
    ncclComm_t comm = nullptr;
    int n_ranks = 0, rank = 0;
    int range_cus = 0;                 // > 0: persistent range launches size their grid for this many CUs (sarx_set_range_cus; frames in flight)
    int range_impl = 0;                // SARX_RANGE_IMPL: 0 auto, 1 = 16 pts/thread, 2 = 32 pts/thread split exchange, 3 = fused wave-private, 4 = sixteen-wave permuted-spectrum pair
    std::string err;
};

struct sarx_plan {
    sarx_ctx* ctx = nullptr;
    int n_az = 0, n_rg = 0;
    unsigned flags = 0;
    sarx_radar_params p{};
    int az_s = 0;          // four-step split: n_az = (n_az/az_s) * az_s; az_s == n_az means single step
    int az_w = 32;         // azimuth tile width (range samples)
    int az_w_alone = 0;    // > 0: width of the plain azimuth launches while the focus has the chip to itself (no CU share set): 64 columns =
                           // 512-byte row segments at 16384^2, 1.57-1.58 against 1.62-1.66 ms per two-launch transform; with frames in flight
                           // the 64 KiB tiles share CUs worse with the other lane's range launch (4.03-4.09 against 3.99-4.00 ms per frame),
                           // at 8192^2 and below nothing changes (profiles/r05_p_az_tile_width.log)
    int look = 0;          // > 0: the last azimuth launch also writes row-wise |x|^2 partials and a finish launch turns them into look_slot
    float* look_slot = nullptr;   // caller's [n_az/look x n_rg/look] fp32 slot (device)
    float* look_part = nullptr;   // [n_az x n_rg/look], owned by the plan
    // sarx_csa_plan_set_ati: the last azimuth launch emits the ATI / DPCA products of (ati_s1, the image being written)
    const float2* ati_s1 = nullptr; const float* ati_thr = nullptr; float ati_frac = 0.f; double ati_cal = 0.0;
    float *ati_phase = nullptr, *ati_m1 = nullptr, *ati_dm = nullptr; int ati_keep_image = 0;
    double2* ati_part = nullptr; int ati_nparts = 0;
    int ati_w = 32;                    // tile width of that launch: 64 columns where n_rg allows (256-byte row segments of the fp32 planes)
    float* max_slot = nullptr;         // sarx_csa_plan_set_max_slot: device float that receives max |image| of every focus
    bool az_nt = false;    // azimuth tile launches use nontemporal accesses (images >= 512 MiB; SARX_AZ_NT=0/1 overrides)
    int slab_tiles = 0;    // > 0: slab mode of sarx_csa_focus_dev with this many azimuth tiles per group (SARX_SLAB_MIB)
    double2 *c1 = nullptr, *c2 = nullptr, *c3 = nullptr;
    float2* buf_b = nullptr;           // scratch image
    float2* buf_a = nullptr;           // second scratch (RG_MAJOR only)
    float2 *h_in = nullptr, *h_out = nullptr;   // device staging for the *_host entry point
    // sarx_csa_focus_host_begin / _end: PIPE frames in flight between upload, focus and download
    static constexpr int PIPE = 2;
    float2 *pipe_in[PIPE] = {}, *pipe_out[PIPE] = {};
    int pipe_dl[PIPE] = {-1, -1};           // ctx download slot of the frame in pipeline slot i, -1 = free
    void* pipe_host[PIPE] = {};             // pageable destination of slot i (downloaded by _end), NULL when the DMA already targets it
    int pipe_next = 0;
    uint64_t bytes = 0;
    int mark_start = -1, mark_stop = -1;   // ctx event slots recorded around the range pass(es)
    unsigned long long* stamp = nullptr;   // sarx_csa_plan_stamp_range: {min start, max end} of the fused range launch (s_memrealtime ticks)
    GeneralCsa* gen = nullptr;             // chirp-z path for sizes that are not powers of two in [16, 16384]
};

static int fail(sarx_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_init_error = buf;
    return code;
}
#define HIPCHK(c, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((c), SARX_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// No C++ exception crosses the C ABI: entry points that allocate on the host (new, std::vector, std::string) run their body through
// this guard; std::bad_alloc becomes SARX_ERR_NOMEM, anything else SARX_ERR_DEVICE, the message is set without allocating again.
template <class F> static int guarded(sarx_ctx* c, F&& body) noexcept {
    int code = SARX_ERR_DEVICE;
    const char* what = "unexpected C++ exception inside libsarx";
    try {
        return body();
    } catch (const std::bad_alloc&) {
        code = SARX_ERR_NOMEM; what = "out of host memory";
    } catch (...) {
    }
    try { if (c) c->err.assign(what); else g_init_error.assign(what); } catch (...) {}
    return code;
}

// every lane's stream (sarx_select_lane): host-visible operations are ordered after all of them
static hipError_t sync_all_lanes(sarx_ctx* c) {
    for (int k = 0; k < sarx_ctx::LANES; ++k)
        if (c->lane[k]) { hipError_t e = hipStreamSynchronize(c->lane[k]); if (e != hipSuccess) return e; }
    return hipSuccess;
}

// The runtime calls staged_copy makes, behind function pointers: the sanitizer build (make asan, -DSARX_TESTING) replaces them with host
// stand-ins so that the chunking, the thread / inline-share / join logic and the error paths run under AddressSanitizer on a box
// without a GPU (tests/asan/abi_asan_test.cpp).  The product never changes the table.
struct CopyOps {
    hipError_t (*memcpy_async)(void*, const void*, size_t, hipMemcpyKind, hipStream_t) = hipMemcpyAsync;
    hipError_t (*stream_sync)(hipStream_t) = hipStreamSynchronize;
    hipError_t (*stream_create)(hipStream_t*, unsigned) = hipStreamCreateWithFlags;
    hipError_t (*event_create)(hipEvent_t*, unsigned) = hipEventCreateWithFlags;
    hipError_t (*event_record)(hipEvent_t, hipStream_t) = hipEventRecord;
    hipError_t (*event_sync)(hipEvent_t) = hipEventSynchronize;
    hipError_t (*host_alloc)(void**, size_t, unsigned) = [](void** p, size_t n, unsigned f) { return hipHostMalloc(p, n, f); };
    hipError_t (*set_device)(int) = hipSetDevice;
    bool (*page_locked)(const void*) = [](const void* p) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost) return true;
        (void)hipGetLastError();               // a pointer the runtime does not know is ordinary pageable memory: clear that error
        return false;
    };
    bool (*may_start_thread)(int) = [](int) { return true; };     // false = behave as if std::thread threw for share i
};
static CopyOps g_ops;

// Host <-> device copy of a large pageable buffer.  hipMemcpy from pageable memory runs at 8 GB/s here and into untouched
// memory (a fresh NumPy array) at 13 GB/s (tools/pcibench.hip); eight threads staging 32 MiB chunks through pinned buffers reach
// 51-54 GB/s both ways.  Blocking; ordered after everything on the ctx stream.  Small copies take the plain path.
// narrow: (host -> device only) the host buffer holds complex128 and is rounded to complex64 on the way into the pinned chunk
// (the reference's arrays are complex128; a NumPy astype of 2^28 elements costs more than the whole transfer)
// ordered = false (uploads into a buffer no enqueued work touches, downloads of data already complete): the copy does not wait for
// the lanes and runs on streams of its own, so it overlaps whatever the GPU is doing
// ordered: the copy waits for every lane's enqueued work (the buffer may have been written on any of them); lane_only: it waits for
// the CURRENT lane only and is issued behind it (a table that only this lane's launches read: the other lanes keep running)
static hipError_t staged_copy(sarx_ctx* c, void* dst, const void* src, size_t bytes, bool to_device, bool narrow = false, bool ordered = true,
                              bool lane_only = false) {
    const CopyOps& o = g_ops;
    hipError_t e = hipSuccess;
    if (ordered && lane_only) e = o.stream_sync(c->stream);
    else if (ordered)
        for (int k = 0; k < sarx_ctx::LANES && e == hipSuccess; ++k)
            if (c->lane[k]) e = o.stream_sync(c->lane[k]);
    if (e != hipSuccess) return e;
    hipStream_t direct = c->stream;
    if (!ordered) {
        if (!c->up_stream && (e = o.stream_create(&c->up_stream, hipStreamNonBlocking)) != hipSuccess) return e;
        direct = c->up_stream;
    }
    const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    // a buffer from sarx_host_alloc (page-locked, already faulted in) needs no staging: one DMA at the PCIe rate, no host memcpy,
    // no first touch
    if (!narrow && (bytes < 4 * sarx_ctx::COPY_CHUNK || o.page_locked(to_device ? src : dst))) {
        e = o.memcpy_async(dst, src, bytes, kind, direct);
        return e != hipSuccess ? e : o.stream_sync(direct);
    }
    constexpr int T = sarx_ctx::COPY_THREADS;
    constexpr size_t CH = sarx_ctx::COPY_CHUNK;
    std::lock_guard<std::mutex> lock(c->copy_mu);      // ctypes callers release the GIL: two host threads may arrive on one ctx
    if (!c->pin[0] && (e = o.host_alloc((void**)&c->pin[0], CH, hipHostMallocDefault)) != hipSuccess) return e;
    if (!c->copy_stream[0] && (e = o.stream_create(&c->copy_stream[0], hipStreamNonBlocking)) != hipSuccess) return e;
    if (narrow && bytes <= CH) {      // a small complex128 upload: rounded on the calling thread through one chunk, no thread is started
        const double* in = (const double*)src;
        float* out = (float*)c->pin[0];
        for (size_t k = 0; k < bytes / sizeof(float); ++k) out[k] = (float)in[k];
        e = o.memcpy_async(dst, c->pin[0], bytes, hipMemcpyHostToDevice, c->copy_stream[0]);
        return e != hipSuccess ? e : o.stream_sync(c->copy_stream[0]);
    }
    for (int i = 1; i < T; ++i) {
        if (!c->pin[i] && (e = o.host_alloc((void**)&c->pin[i], CH, hipHostMallocDefault)) != hipSuccess) return e;
        if (!c->copy_stream[i] && (e = o.stream_create(&c->copy_stream[i], hipStreamNonBlocking)) != hipSuccess) return e;
    }
    for (int i = 0; i < T; ++i)
        if (!c->pin_free[i] && (e = o.event_create(&c->pin_free[i], hipEventDisableTiming)) != hipSuccess) return e;
    const int US = to_device ? c->up_streams : T;       // uploads: thread i's DMAs go to copy stream i % US
    hipError_t errs[T];
    for (int i = 0; i < T; ++i) errs[i] = hipSuccess;
    std::thread th[T];                 // fixed storage: nothing here allocates, so nothing but thread creation can throw
    // thread i copies chunks i, i + T, ...; if a thread cannot be started (std::system_error must not cross the C ABI) the
    // calling thread does that share itself after the others
    auto share = [=, &errs](int i) {
            hipError_t r = o.set_device(c->device);
            char* d = (char*)dst;
            const char* s0 = (const char*)src;
            for (size_t off = (size_t)i * CH; r == hipSuccess && off < bytes; off += (size_t)T * CH) {
                const size_t len = bytes - off < CH ? bytes - off : CH;
                if (to_device) {
                    if (off >= (size_t)T * CH) r = o.event_sync(c->pin_free[i]);     // the chunk's previous DMA has left the pinned buffer
                    if (r != hipSuccess) break;
                    if (narrow) {
                        const double* in = (const double*)s0 + off / sizeof(float);      // off counts complex64 bytes: 2 floats <-> 2 doubles
                        float* out = (float*)c->pin[i];
                        for (size_t k = 0; k < len / sizeof(float); ++k) out[k] = (float)in[k];
                    } else {
                        memcpy(c->pin[i], s0 + off, len);
                    }
                    r = o.memcpy_async(d + off, c->pin[i], len, hipMemcpyHostToDevice, c->copy_stream[i % US]);
                    if (r == hipSuccess) r = o.event_record(c->pin_free[i], c->copy_stream[i % US]);
                } else {
                    r = o.memcpy_async(c->pin[i], s0 + off, len, hipMemcpyDeviceToHost, c->copy_stream[i]);
                    if (r == hipSuccess) r = o.stream_sync(c->copy_stream[i]);
                    if (r == hipSuccess) memcpy(d + off, c->pin[i], len);
                }
            }
            if (r == hipSuccess) r = to_device ? o.event_sync(c->pin_free[i]) : o.stream_sync(c->copy_stream[i]);
            errs[i] = r;
        };
    bool inline_share[T] = {};
    for (int i = 0; i < T; ++i) {
        try {
            if (!o.may_start_thread(i)) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
            th[i] = std::thread(share, i);
        } catch (...) { inline_share[i] = true; }        // std::system_error / std::bad_alloc: no exception crosses the C ABI
    }
    for (int i = 0; i < T; ++i) if (inline_share[i]) share(i);
    for (int i = 0; i < T; ++i) if (th[i].joinable()) th[i].join();     // every started thread is joined on the one exit path
    for (int i = 0; i < T; ++i)
        if (errs[i] != hipSuccess) return errs[i];
    return hipSuccess;
}

static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
static int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

extern "C" {

int sarx_version(void) { return SARX_VERSION; }

const char* sarx_last_error(const sarx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int sarx_device_count(int* out_count) {
    if (!out_count) return fail(nullptr, SARX_ERR_INVALID, "out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out_count = 0; return fail(nullptr, SARX_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *out_count = n;
    return SARX_OK;
}

static int sarx_init_impl(int device_id, sarx_ctx** out_ctx) {
    if (!out_ctx) return fail(nullptr, SARX_ERR_INVALID, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, SARX_ERR_DEVICE, "no HIP device available (%s); libsarx has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device_id < 0 || device_id >= n) return fail(nullptr, SARX_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, n);
    HIPCHK(nullptr, hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(nullptr, hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, SARX_ERR_UNSUPPORTED, "device %d is %s; libsarx is built for gfx950 only", device_id, prop.gcnArchName);
    sarx_ctx* c = new sarx_ctx();
    c->device = device_id;
    if (prop.multiProcessorCount > 0) c->cus = prop.multiProcessorCount;
    if (const char* e2 = getenv("SARX_RANGE_IMPL")) c->range_impl = (e2[0] == 'v') ? atoi(e2 + 1) : atoi(e2);
    if (const char* e2 = getenv("SARX_RANGE_CUS")) c->range_cus = atoi(e2);
    if (const char* e2 = getenv("SARX_UP_STREAMS")) { const int v = atoi(e2); if (v >= 1 && v <= sarx_ctx::COPY_THREADS) c->up_streams = v; }
    HIPCHK(nullptr, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->lane[0] = c->stream;
    HIPCHK(nullptr, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    for (int i = 0; i < N_EVENTS; ++i) HIPCHK(nullptr, hipEventCreate(&c->ev[i]));
    HIPCHK(nullptr, hipEventCreateWithFlags(&c->comm_fence, hipEventDisableTiming));
    HIPCHK(nullptr, hipEventCreateWithFlags(&c->comm_done, hipEventDisableTiming));
    for (int i = 0; i < 4; ++i) HIPCHK(nullptr, hipEventCreateWithFlags(&c->comm_mark[i], hipEventDisableTiming));
    // twiddle tables for every power of two up to TW_MAX, fp64-evaluated
    std::vector<float2> tw(2 * TW_MAX);
    tw[0] = tw[1] = make_float2(1.f, 0.f);
    for (int n2 = 2; n2 <= TW_MAX; n2 <<= 1)
        for (int m = 0; m < n2; ++m) {
            const double ang = -2.0 * M_PI * (double)m / (double)n2;
            tw[n2 + m] = make_float2((float)cos(ang), (float)sin(ang));
        }
    HIPCHK(nullptr, hipMalloc(&c->tw_all, tw.size() * sizeof(float2)));
    HIPCHK(nullptr, hipMemcpy(c->tw_all, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    HIPCHK(nullptr, hipMalloc(&c->ati_part_max_all, sarx_ctx::LANES * 4096 * sizeof(float)));
    HIPCHK(nullptr, hipMalloc(&c->ati_part_sum_all, sarx_ctx::LANES * 4096 * sizeof(double2)));
    HIPCHK(nullptr, hipMalloc(&c->ati_out3_all, sarx_ctx::LANES * 4 * sizeof(double)));
    HIPCHK(nullptr, hipMalloc(&c->power_part_all, sarx_ctx::LANES * sarx_ctx::POWER_STRIDE * sizeof(double)));
    *out_ctx = c;
    return SARX_OK;
}
int sarx_init(int device_id, sarx_ctx** out_ctx) {
    return guarded(nullptr, [&] { return sarx_init_impl(device_id, out_ctx); });
}

int sarx_destroy(sarx_ctx* c) {
    if (!c) return SARX_OK;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    hipFree(c->tw_all); hipFree(c->ati_part_max_all); hipFree(c->ati_part_sum_all); hipFree(c->ati_out3_all); hipFree(c->power_part_all);
    for (int i = 0; i < N_EVENTS; ++i) hipEventDestroy(c->ev[i]);
    hipEventDestroy(c->comm_fence);
    hipEventDestroy(c->comm_done);
    for (int i = 0; i < 4; ++i) hipEventDestroy(c->comm_mark[i]);
    for (int i = 0; i < sarx_ctx::COPY_THREADS; ++i) {
        if (c->pin[i]) hipHostFree(c->pin[i]);
        if (c->copy_stream[i]) hipStreamDestroy(c->copy_stream[i]);
        if (c->pin_free[i]) hipEventDestroy(c->pin_free[i]);
    }
    for (int i = 0; i < sarx_ctx::DL_SLOTS; ++i) {
        if (c->dl_ready[i]) hipEventDestroy(c->dl_ready[i]);
        if (c->dl_done[i]) hipEventDestroy(c->dl_done[i]);
    }
    if (c->dl_stream) hipStreamDestroy(c->dl_stream);
    if (c->up_stream) hipStreamDestroy(c->up_stream);
    for (int k = 0; k < sarx_ctx::LANES; ++k) {
        if (c->lane_ev[k]) hipEventDestroy(c->lane_ev[k]);
        if (k > 0 && c->lane[k]) hipStreamDestroy(c->lane[k]);
    }
    hipStreamDestroy(c->lane[0]);
    hipStreamDestroy(c->comm_stream);
    delete c;
    return SARX_OK;
}

int sarx_persistent_grid(int wgs_per_cu, int cus, int work_items) { return persistent_grid(wgs_per_cu, cus, work_items); }

int sarx_device_info(sarx_ctx* c, char* name, size_t name_len, int* cus, uint64_t* hbm, char* arch, size_t arch_len) {
    if (!c) return fail(nullptr, SARX_ERR_INVALID, "ctx is NULL");
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    if (name && name_len) snprintf(name, name_len, "%s", prop.name);
    if (arch && arch_len) snprintf(arch, arch_len, "%s", prop.gcnArchName);
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm) *hbm = (uint64_t)prop.totalGlobalMem;
    return SARX_OK;
}

// ---- memory / timing ----------------------------------------------------------
#define NEED_CTX(c) do { if (!(c)) return fail(nullptr, SARX_ERR_INVALID, "ctx is NULL"); hipSetDevice((c)->device); } while (0)

int sarx_malloc(sarx_ctx* c, size_t bytes, void** out) {
    NEED_CTX(c);
    if (!out) return fail(c, SARX_ERR_INVALID, "out_dptr is NULL");
    *out = nullptr;
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(c, SARX_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return SARX_OK;
}
int sarx_free(sarx_ctx* c, void* p) { NEED_CTX(c); HIPCHK(c, hipFree(p)); return SARX_OK; }
int sarx_host_alloc(sarx_ctx* c, size_t bytes, void** out) {
    NEED_CTX(c);
    if (!out) return fail(c, SARX_ERR_INVALID, "out_hptr is NULL");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(c, SARX_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return SARX_OK;
}
int sarx_host_free(sarx_ctx* c, void* p) { NEED_CTX(c); if (p) HIPCHK(c, hipHostFree(p)); return SARX_OK; }
int sarx_memcpy_h2d(sarx_ctx* c, void* d, const void* s, size_t n) {
    NEED_CTX(c);
    HIPCHK(c, staged_copy(c, d, s, n, true));
    return SARX_OK;
}
int sarx_memcpy_d2h(sarx_ctx* c, void* d, const void* s, size_t n) {
    NEED_CTX(c);
    HIPCHK(c, staged_copy(c, d, s, n, false));
    return SARX_OK;
}
int sarx_memcpy_d2d(sarx_ctx* c, void* d, const void* s, size_t n) {
    NEED_CTX(c);
    HIPCHK(c, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, c->stream));
    return SARX_OK;
}
int sarx_memcpy_h2d_lane(sarx_ctx* c, void* d, const void* s, size_t n) {
    NEED_CTX(c);
    if (!d || !s) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    HIPCHK(c, staged_copy(c, d, s, n, true, false, /*ordered=*/true, /*lane_only=*/true));
    return SARX_OK;
}
int sarx_memcpy_h2d_unordered(sarx_ctx* c, void* d, const void* s, size_t n) {
    NEED_CTX(c);
    if (!d || !s) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    HIPCHK(c, staged_copy(c, d, s, n, true, false, /*ordered=*/false));
    return SARX_OK;
}
static bool is_page_locked(const void* p) { return g_ops.page_locked(p); }
int sarx_memcpy_d2h_begin(sarx_ctx* c, void* h, const void* d, size_t n, int* out_slot) {
    NEED_CTX(c);
    if (!h || !d || !out_slot) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    *out_slot = -1;
    if (!is_page_locked(h))
        return fail(c, SARX_ERR_INVALID, "sarx_memcpy_d2h_begin needs a page-locked destination (sarx_host_alloc): a pageable one cannot be "
                                         "written by an asynchronous DMA (use sarx_memcpy_d2h)");
    int slot = -1;
    for (int i = 0; i < sarx_ctx::DL_SLOTS; ++i) if (!c->dl_busy[i]) { slot = i; break; }
    if (slot < 0) return fail(c, SARX_ERR_INVALID, "all %d download slots are in flight: call sarx_memcpy_d2h_end first", sarx_ctx::DL_SLOTS);
    if (!c->dl_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->dl_stream, hipStreamNonBlocking));
    if (!c->dl_ready[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->dl_ready[slot], hipEventDisableTiming));
    if (!c->dl_done[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->dl_done[slot], hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->dl_ready[slot], c->stream));              // everything enqueued on the current lane so far
    HIPCHK(c, hipStreamWaitEvent(c->dl_stream, c->dl_ready[slot], 0));
    {   // in pieces (SARX_DL_CHUNK_MIB, A/B): does one 2 GiB download hold up the upload's 32 MiB DMAs more than many small ones?
        // beside a staged upload 2 GiB as one DMA took 65.9 ms for both, in 32 MiB pieces 62.9, in 256 MiB pieces 59.0 (profiles/r05_i_duplex.log)
        static const size_t piece = [] { const char* e = getenv("SARX_DL_CHUNK_MIB"); return (size_t)(e ? atoi(e) : 256) << 20; }();
        const size_t step = piece ? piece : n;
        for (size_t off = 0; off < n; off += step)
            HIPCHK(c, hipMemcpyAsync((char*)h + off, (const char*)d + off, n - off < step ? n - off : step, hipMemcpyDeviceToHost, c->dl_stream));
    }
    HIPCHK(c, hipEventRecord(c->dl_done[slot], c->dl_stream));
    c->dl_busy[slot] = true;
    *out_slot = slot;
    return SARX_OK;
}
int sarx_memcpy_d2h_end(sarx_ctx* c, int slot) {
    NEED_CTX(c);
    if (slot < 0 || slot >= sarx_ctx::DL_SLOTS || !c->dl_busy[slot]) return fail(c, SARX_ERR_INVALID, "download slot %d is not in flight", slot);
    c->dl_busy[slot] = false;                                            // released whatever the wait returns
    HIPCHK(c, hipEventSynchronize(c->dl_done[slot]));
    return SARX_OK;
}
int sarx_memcpy2d_d2h(sarx_ctx* c, void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height) {
    NEED_CTX(c);
    if (!d || !s || width > dpitch || width > spitch) return fail(c, SARX_ERR_INVALID, "bad 2-D copy arguments");
    if (!width || !height) return SARX_OK;
    HIPCHK(c, sync_all_lanes(c));
    HIPCHK(c, hipMemcpy2DAsync(d, dpitch, s, spitch, width, height, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SARX_OK;
}
int sarx_memcpy2d_h2d(sarx_ctx* c, void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height) {
    NEED_CTX(c);
    if (!d || !s || width > dpitch || width > spitch) return fail(c, SARX_ERR_INVALID, "bad 2-D copy arguments");
    if (!width || !height) return SARX_OK;
    HIPCHK(c, sync_all_lanes(c));
    HIPCHK(c, hipMemcpy2DAsync(d, dpitch, s, spitch, width, height, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SARX_OK;
}
int sarx_memset(sarx_ctx* c, void* d, int v, size_t n) { NEED_CTX(c); HIPCHK(c, hipMemsetAsync(d, v, n, c->stream)); return SARX_OK; }
int sarx_select_lane(sarx_ctx* c, int lane) {
    NEED_CTX(c);
    if (lane < 0 || lane >= sarx_ctx::LANES) return fail(c, SARX_ERR_INVALID, "lane %d out of range [0,%d)", lane, sarx_ctx::LANES);
    if (!c->lane[lane]) HIPCHK(c, hipStreamCreateWithFlags(&c->lane[lane], hipStreamNonBlocking));
    c->cur_lane = lane;
    c->stream = c->lane[lane];
    return SARX_OK;
}
// How concurrently do two lanes run?  Two small launches (64 one-wave workgroups spinning ~`us` microseconds each) on lanes a and b,
// timed from the host: *ratio = time of both together / time of one alone - 1.0 when the lanes' hardware queues run side by side,
// 2.0 when they take turns.
int sarx_probe_lanes(sarx_ctx* c, int a, int b, int us, double* ratio) {
    NEED_CTX(c);
    if (a < 0 || b < 0 || a >= sarx_ctx::LANES || b >= sarx_ctx::LANES || a == b || !ratio || us <= 0)
        return fail(c, SARX_ERR_INVALID, "bad lane probe arguments");
    for (int l : {a, b})
        if (!c->lane[l]) HIPCHK(c, hipStreamCreateWithFlags(&c->lane[l], hipStreamNonBlocking));
    unsigned* sink = reinterpret_cast<unsigned*>(c->power_part_all);          // never written (the kernel's condition is never true)
    const unsigned long long cycles = (unsigned long long)us * 100ull;         // s_memrealtime counts at 100 MHz
    auto wall = [&](bool both, double& ms) -> hipError_t {
        hipError_t e = sync_all_lanes(c);
        if (e != hipSuccess) return e;
        const auto t0 = std::chrono::steady_clock::now();
        e = launch_spin(64, cycles, sink, c->lane[a]);
        if (e == hipSuccess && both) e = launch_spin(64, cycles, sink, c->lane[b]);
        if (e == hipSuccess) e = hipStreamSynchronize(c->lane[a]);
        if (e == hipSuccess && both) e = hipStreamSynchronize(c->lane[b]);
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return e;
    };
    double one = 1e30, two = 1e30, ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        HIPCHK(c, wall(false, ms)); if (ms < one) one = ms;
        HIPCHK(c, wall(true, ms)); if (ms < two) two = ms;
    }
    *ratio = two / one;
    return SARX_OK;
}
int sarx_set_range_cus(sarx_ctx* c, int cus) {
    NEED_CTX(c);
    if (cus < 0) return fail(c, SARX_ERR_INVALID, "cus must be >= 0 (0 = all)");
    c->range_cus = cus;
    return SARX_OK;
}
int sarx_lanes_join(sarx_ctx* c) {
    NEED_CTX(c);
    for (int k = 0; k < sarx_ctx::LANES; ++k) {
        if (!c->lane[k]) continue;
        if (!c->lane_ev[k]) HIPCHK(c, hipEventCreateWithFlags(&c->lane_ev[k], hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->lane_ev[k], c->lane[k]));
    }
    for (int k = 0; k < sarx_ctx::LANES; ++k)
        for (int j = 0; j < sarx_ctx::LANES; ++j)
            if (j != k && c->lane[k] && c->lane[j]) HIPCHK(c, hipStreamWaitEvent(c->lane[k], c->lane_ev[j], 0));
    return SARX_OK;
}
int sarx_sync(sarx_ctx* c) {
    NEED_CTX(c);
    HIPCHK(c, sync_all_lanes(c));
    HIPCHK(c, hipStreamSynchronize(c->comm_stream));
    if (c->dl_stream) HIPCHK(c, hipStreamSynchronize(c->dl_stream));
    return SARX_OK;
}
int sarx_event_record(sarx_ctx* c, int slot) {
    NEED_CTX(c);
    if (slot < 0 || slot >= N_EVENTS) return fail(c, SARX_ERR_INVALID, "event slot %d out of range", slot);
    HIPCHK(c, hipEventRecord(c->ev[slot], c->stream));
    c->ev_set[slot] = true;
    return SARX_OK;
}
int sarx_event_elapsed_ms(sarx_ctx* c, int a, int b, float* ms) {
    NEED_CTX(c);
    if (a < 0 || a >= N_EVENTS || b < 0 || b >= N_EVENTS || !ms) return fail(c, SARX_ERR_INVALID, "bad event slots");
    if (!c->ev_set[a] || !c->ev_set[b]) return fail(c, SARX_ERR_INVALID, "event slot not recorded");
    HIPCHK(c, hipEventSynchronize(c->ev[b]));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev[a], c->ev[b]));
    return SARX_OK;
}

// ---- CSA plan -----------------------------------------------------------------
static int sarx_csa_plan_create_impl(sarx_ctx* c, int n_az, int n_rg, const sarx_radar_params* prm, unsigned flags, sarx_plan** out) {
    NEED_CTX(c);
    if (!out || !prm) return fail(c, SARX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_az < 2 || n_rg < 2 || n_az > 2 * TW_MAX || n_rg > 2 * TW_MAX)
        return fail(c, SARX_ERR_UNSUPPORTED, "n_az=%d n_rg=%d: sizes must be in [2, %d]", n_az, n_rg, 2 * TW_MAX);
    if (flags & ~(SARX_OUT_RG_MAJOR | SARX_FUSE_RANGE)) return fail(c, SARX_ERR_INVALID, "unknown plan flags 0x%x", flags);
    const bool general = !is_pow2(n_az) || !is_pow2(n_rg) || n_az < 16 || n_rg < 16 || n_az > TW_MAX || n_rg > TW_MAX;
    if (!(prm->sample_rate_hz > 0) || !(prm->prf_hz > 0) || !(prm->platform_speed_mps > 0) ||
        !(prm->wavelength_m > 0) || prm->chirp_rate_hz_s == 0.0)
        return fail(c, SARX_ERR_INVALID, "radar parameters must be positive (chirp rate non-zero)");
    sarx_plan* p = new sarx_plan();
    p->ctx = c; p->n_az = n_az; p->n_rg = n_rg; p->flags = flags; p->p = *prm;
    if (general) {       // any other size: chirp-z transforms over the power-of-two kernels (general.hip)
        std::string err;
        p->gen = general_csa_create(n_az, n_rg, prm, c->tw_all, err, true, c->cus);
        if (!p->gen) { delete p; return fail(c, SARX_ERR_UNSUPPORTED, "n_az=%d n_rg=%d: %s", n_az, n_rg, err.c_str()); }
        p->bytes = general_csa_bytes(p->gen);
        if (flags & SARX_OUT_RG_MAJOR) {
            hipError_t e2 = hipMalloc(&p->buf_a, (size_t)n_az * n_rg * sizeof(float2));
            if (e2 != hipSuccess) { int rc = fail(c, SARX_ERR_NOMEM, "hipMalloc: %s", hipGetErrorString(e2)); sarx_csa_plan_destroy(p); return rc; }
            p->bytes += (size_t)n_az * n_rg * sizeof(float2);
        }
        *out = p;
        return SARX_OK;
    }
    p->az_s = (n_az <= 128) ? n_az : (1 << (ilog2(n_az) / 2));
    p->az_w = (n_rg % 32 == 0) ? 32 : 16;
    p->az_nt = (size_t)n_az * n_rg * sizeof(float2) >= ((size_t)1 << 29);     // 8192^2 and up (measured: +2 % / +3.8 % at 8192^2 / 16384^2, -3 % at 4096^2)
    if (const char* e = getenv("SARX_AZ_NT")) p->az_nt = atoi(e) != 0;
    if (const char* e = getenv("SARX_SLAB_MIB")) {     // rows of one group of tiles, in MiB (0 = off)
        const double mib = atof(e);
        const double tile_mib = (double)p->az_s * n_rg * sizeof(float2) / (1024.0 * 1024.0);
        if (mib > 0 && p->az_s != n_az) {
            int q = (int)(mib / tile_mib);
            if (q < 1) q = 1;
            if (q > n_az / p->az_s) q = n_az / p->az_s;
            p->slab_tiles = q;
        }
    }
    if (const char* e = getenv("SARX_AZ_W")) { const int w = atoi(e); if ((w == 16 || w == 32 || w == 64) && n_rg % w == 0) p->az_w = w; }
    else if (n_rg % 64 == 0 && (size_t)n_az * n_rg * sizeof(float2) >= ((size_t)1 << 31)) p->az_w_alone = 64;

    // migration factors, natural fftfreq order (sar_ati_dcpa_sim_csa.py:225,244-249,262)
    const double lam = prm->wavelength_m, Kr = prm->chirp_rate_hz_s, Vr = prm->platform_speed_mps, Rref = prm->range_ref_m;
    const double fa_step = 1.0 / ((double)n_az * (1.0 / prm->prf_hz));
    std::vector<double2> c1(n_az), c2(n_az), c3(n_az);
    for (int i = 0; i < n_az; ++i) {
        const int ks = (i < n_az / 2) ? i : i - n_az;
        const double fa = (double)ks * fa_step;
        const double u = lam * fa / (2.0 * Vr);
        double arg = 1.0 - u * u;
        if (arg < 0) arg = 1e-9;                           // :246 sets, does not clamp to 0
        const double D = sqrt(arg);
        const double Cs = 1.0 / D - 1.0;
        const double tau_ref = 2.0 * Rref / (C_LIGHT * D);
        c1[i] = make_double2(-0.5 * Kr * Cs, tau_ref);
        c2[i] = make_double2(0.5 / (Kr * (1.0 + Cs)), 2.0 * Rref * Cs / C_LIGHT);
        c3[i] = make_double2(C_LIGHT * D / lam, -0.5 * Kr * Cs * (1.0 + Cs));
    }
    const size_t tb = (size_t)n_az * sizeof(double2), img = (size_t)n_az * n_rg * sizeof(float2);
    auto bail = [&](hipError_t e, const char* what) {
        int rc = fail(c, e == hipErrorOutOfMemory ? SARX_ERR_NOMEM : SARX_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
        sarx_csa_plan_destroy(p);
        return rc;
    };
    hipError_t e;
    if ((e = hipMalloc(&p->c1, tb)) != hipSuccess) return bail(e, "hipMalloc c1");
    if ((e = hipMalloc(&p->c2, tb)) != hipSuccess) return bail(e, "hipMalloc c2");
    if ((e = hipMalloc(&p->c3, tb)) != hipSuccess) return bail(e, "hipMalloc c3");
    if ((e = hipMemcpy(p->c1, c1.data(), tb, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "upload c1");
    if ((e = hipMemcpy(p->c2, c2.data(), tb, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "upload c2");
    if ((e = hipMemcpy(p->c3, c3.data(), tb, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "upload c3");
    if ((e = hipMalloc(&p->buf_b, img)) != hipSuccess) return bail(e, "hipMalloc scratch image");
    p->bytes = 3 * tb + img;
    if (flags & SARX_OUT_RG_MAJOR) {
        if ((e = hipMalloc(&p->buf_a, img)) != hipSuccess) return bail(e, "hipMalloc second scratch image");
        p->bytes += img;
    }
    *out = p;
    return SARX_OK;
}
int sarx_csa_plan_create(sarx_ctx* c, int n_az, int n_rg, const sarx_radar_params* prm, unsigned flags, sarx_plan** out) {
    return guarded(c, [&] { return sarx_csa_plan_create_impl(c, n_az, n_rg, prm, flags, out); });
}

int sarx_csa_plan_destroy(sarx_plan* p) {
    if (!p) return SARX_OK;
    hipSetDevice(p->ctx->device);
    sync_all_lanes(p->ctx);
    general_csa_destroy(p->gen);
    hipFree(p->c1); hipFree(p->c2); hipFree(p->c3);
    hipFree(p->ati_part);
    hipFree(p->buf_a); hipFree(p->buf_b); hipFree(p->h_in); hipFree(p->h_out); hipFree(p->look_part);
    for (int i = 0; i < sarx_plan::PIPE; ++i)             // frames still in the pipeline: their download slots go back to the ctx
        if (p->pipe_dl[i] >= 0 && p->pipe_dl[i] < sarx_ctx::DL_SLOTS) { sarx_memcpy_d2h_end(p->ctx, p->pipe_dl[i]); p->pipe_dl[i] = -1; }
    if (p->ctx->dl_stream) hipStreamSynchronize(p->ctx->dl_stream);
    for (int i = 0; i < sarx_plan::PIPE; ++i) { hipFree(p->pipe_in[i]); hipFree(p->pipe_out[i]); }
    delete p;
    return SARX_OK;
}

int sarx_csa_plan_mark_range(sarx_plan* p, int slot_start, int slot_stop) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    if (slot_start >= N_EVENTS || slot_stop >= N_EVENTS) return fail(p->ctx, SARX_ERR_INVALID, "event slot out of range");
    p->mark_start = slot_start; p->mark_stop = slot_stop;
    return SARX_OK;
}

int sarx_csa_plan_stamp_range(sarx_plan* p, uint64_t* d_pair) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    if (((uintptr_t)d_pair) & 7) return fail(p->ctx, SARX_ERR_INVALID, "the stamp pair must be 8-byte aligned");
    p->stamp = reinterpret_cast<unsigned long long*>(d_pair);
    return SARX_OK;
}

int sarx_csa_plan_set_look_slot(sarx_plan* p, int looks, float* d_slot) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (!d_slot) { p->look_slot = nullptr; return SARX_OK; }            // switch off; the partials buffer is kept
    if (p->gen) return fail(c, SARX_ERR_UNSUPPORTED, "the fused stack slot exists for power-of-two plans only (use sarx_multilook_dev)");
    if (looks < 1 || (looks & (looks - 1)) || looks > p->az_w || p->n_az % looks || p->n_rg % looks)
        return fail(c, SARX_ERR_UNSUPPORTED, "looks=%d must be a power of two <= %d dividing n_az=%d and n_rg=%d", looks, p->az_w, p->n_az, p->n_rg);
    if (p->look_part && p->look != looks) { hipStreamSynchronize(c->stream); hipFree(p->look_part); p->look_part = nullptr; }
    if (!p->look_part) {
        const size_t bytes = (size_t)p->n_az * (p->n_rg / looks) * sizeof(float);
        hipError_t e = hipMalloc(&p->look_part, bytes);
        if (e != hipSuccess) return fail(c, SARX_ERR_NOMEM, "hipMalloc look partials: %s", hipGetErrorString(e));
        p->bytes += bytes;
    }
    p->look = looks; p->look_slot = d_slot;
    return SARX_OK;
}

int sarx_csa_plan_set_max_slot(sarx_plan* p, float* d_max) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    if (p->gen && !general_csa_set_max_slot(p->gen, reinterpret_cast<unsigned*>(d_max)))
        return fail(p->ctx, SARX_ERR_UNSUPPORTED, "the fused maximum exists for power-of-two plans and 7199 x 13200 (sarx_ati_dpca_dev reduces it otherwise)");
    p->max_slot = d_max;
    return SARX_OK;
}

int sarx_csa_plan_set_ati(sarx_plan* p, const void* d_slc1, const float* d_max, float mask_frac, double cal_phase,
                          float* d_ati_phase_masked, float* d_slc1_mag, float* d_dpca_mag, int keep_image) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (!d_slc1) {
        p->ati_s1 = nullptr;
        if (p->gen) general_csa_set_ati(p->gen, nullptr);
        return SARX_OK;
    }
    if (!d_max || !d_ati_phase_masked || !d_slc1_mag || !d_dpca_mag) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (p->gen) {        // the native 7199 x 13200: the inverse DFT-23 launch of the prime-factor route has the same epilogue
        const int parts = general_csa_ati_parts(p->gen);
        if (parts < 0 || (p->flags & SARX_OUT_RG_MAJOR))
            return fail(c, SARX_ERR_UNSUPPORTED, "the fused ATI products exist for power-of-two plans and 7199 x 13200 in the default image "
                                                 "layout (sarx_ati_dpca_dev otherwise)");
        if (!p->ati_part) {
            hipError_t e = hipMalloc(&p->ati_part, ((size_t)parts + 128) * sizeof(double2));
            if (e != hipSuccess) return fail(c, SARX_ERR_NOMEM, "hipMalloc ATI partial sums: %s", hipGetErrorString(e));
            p->bytes += (size_t)parts * sizeof(double2);
        }
        p->ati_nparts = parts;
        AtiFuse f{};
        f.s1 = (const float2*)d_slc1; f.phase = d_ati_phase_masked; f.m1 = d_slc1_mag; f.dm = d_dpca_mag; f.part = p->ati_part;
        f.thr = d_max; f.cc = (float)cos(cal_phase); f.cs = (float)sin(cal_phase); f.frac = mask_frac; f.keep_image = keep_image != 0;
        general_csa_set_ati(p->gen, &f);
        p->ati_s1 = f.s1; p->ati_thr = d_max;
        return SARX_OK;
    }
    if (p->n_rg % 32 || (p->flags & SARX_OUT_RG_MAJOR) || p->slab_tiles > 0)
        return fail(c, SARX_ERR_UNSUPPORTED, "the fused ATI products exist for power-of-two plans in the default image layout (sarx_ati_dpca_dev otherwise)");
    int w = (p->n_rg % 64 == 0) ? 64 : 32;          // 64 columns where n_rg allows: 256-byte row segments of the fp32 planes
    if (const char* ev = getenv("SARX_ATI_W")) { const int e = atoi(ev); if ((e == 32 || e == 64) && p->n_rg % e == 0) w = e; }
    const int tpt = (p->az_s >= 16 ? p->az_s / 16 : 1) * w;        // threads per tile of the last azimuth launch (rows az_s)
    if (tpt % 64) return fail(c, SARX_ERR_UNSUPPORTED, "n_az=%d n_rg=%d: the last azimuth launch's tiles have %d threads, the fused ATI "
                                                        "products need whole waves (sarx_ati_dpca_dev otherwise)", p->n_az, p->n_rg, tpt);
    p->ati_w = w;
    const int tiles = (p->az_s == p->n_az ? 1 : p->n_az / p->az_s) * (p->n_rg / w);
    const int waves = tpt / 64;
    if (!p->ati_part) {
        hipError_t e = hipMalloc(&p->ati_part, ((size_t)tiles * waves + 128) * sizeof(double2));     // + the finish's first-level results
        if (e != hipSuccess) return fail(c, SARX_ERR_NOMEM, "hipMalloc ATI partial sums: %s", hipGetErrorString(e));
        p->bytes += (size_t)tiles * waves * sizeof(double2);
    }
    p->ati_nparts = tiles * waves;
    p->ati_s1 = (const float2*)d_slc1; p->ati_thr = d_max; p->ati_frac = mask_frac; p->ati_cal = cal_phase;
    p->ati_phase = d_ati_phase_masked; p->ati_m1 = d_slc1_mag; p->ati_dm = d_dpca_mag; p->ati_keep_image = keep_image != 0;
    return SARX_OK;
}

int sarx_csa_plan_bytes(const sarx_plan* p, uint64_t* out) {
    if (!p || !out) return fail(p ? p->ctx : nullptr, SARX_ERR_INVALID, "NULL argument");
    *out = p->bytes;
    return SARX_OK;
}

int sarx_csa_axes(const sarx_plan* p, double* range_axis, double* cross_range_axis) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    const double dt = 1.0 / p->p.sample_rate_hz;
    if (range_axis)
        for (int j = 0; j < p->n_rg; ++j) range_axis[j] = C_LIGHT * (p->p.t_start_fast_s + (double)j * dt) / 2.0;   // :219,346
    if (cross_range_axis) {
        // t_slow = arange/prf; t_slow -= mean; * Vr   (:392-394), pairwise mean like NumPy is not
        // needed: the reference result is reproduced to 1e-13 relative, stated in the test
        double mean = 0.0;
        for (int i = 0; i < p->n_az; ++i) mean += (double)i / p->p.prf_hz;
        mean /= (double)p->n_az;
        for (int i = 0; i < p->n_az; ++i) cross_range_axis[i] = ((double)i / p->p.prf_hz - mean) * p->p.platform_speed_mps;
    }
    return SARX_OK;
}

static RangeArgs range_args(const sarx_plan* p, const void* in, void* out) {
    RangeArgs a{};
    a.in = (const float2*)in; a.out = (float2*)out;
    a.tw = p->ctx->tw_all + p->n_rg;
    a.c2 = p->c2; a.c3 = p->c3;
    a.dt = 1.0 / p->p.sample_rate_hz;
    a.df = 1.0 / ((double)p->n_rg * a.dt);           // numpy.fft.fftfreq step
    a.t_start = p->p.t_start_fast_s;
    a.t0 = 2.0 * p->p.range_ref_m / C_LIGHT;
    a.inv_n = 1.0f / (float)p->n_rg;
    a.n_az = p->n_az;
    return a;
}

static hipError_t run_range(const sarx_plan* p, int mode, const RangeArgs& a) {
    const sarx_ctx* c = p->ctx;
    // measured on MI355X (profiles/): 32 pts/thread wins for one FFT per launch at n_rg >= 8192,
    // 16 pts/thread wins for the fused FFT+IFFT launch (the 32-pt form spills there)
    const bool v2 = range_v2_supported(p->n_rg, mode) &&
                    (c->range_impl == 2 || (c->range_impl == 0 && p->n_rg >= 16384 && mode != RG_FUSED));
    // impl 3 (default for the fused launch at 16384): wave-private sub-transforms
    if (mode == RG_FUSED && range_fused_wl_supported(p->n_rg) && (c->range_impl == 3 || c->range_impl == 0))
        return launch_range_fused_wl(a, (c->range_cus > 0 && c->range_cus < c->cus) ? c->range_cus : c->cus, c->stream,
                                     /*alone=*/!(c->range_cus > 0 && c->range_cus < c->cus));
    return v2 ? launch_range_pass_v2(p->n_rg, mode, a, c->cus, c->stream) : launch_range_pass(p->n_rg, mode, a, c->stream);
}

// One step of the two-step (four-step) azimuth transform n_az = RA * S over the tiles [q0, q0 + nq):
//   step A: tile q in [0,S):  rows q + m*S, (I)FFT over m (length RA), twiddle W_n^(-+q*m'), same rows of `out`
//   step B: tile q in [0,RA): rows q*S + m, (I)FFT over m (length S), rows q + m'*RA of `out` (natural bin order), epilogue
static void ati_args(const sarx_plan* p, AzArgs& a) {
    a.ati_s1 = p->ati_s1; a.ati_thr = p->ati_thr; a.ati_frac = p->ati_frac;
    a.ati_cc = (float)cos(p->ati_cal); a.ati_cs = (float)sin(p->ati_cal);
    a.ati_phase = p->ati_phase; a.ati_m1 = p->ati_m1; a.ati_dm = p->ati_dm;
    a.ati_part = p->ati_part; a.ati_keep_image = p->ati_keep_image;
}
// the fixed-order finish of the fused ATI products' phase-balance sum, after the last azimuth launch of a focus
static int ati_finish(sarx_plan* p) {
    if (!p->ati_s1) return SARX_OK;
    sarx_ctx* c = p->ctx;
    HIPCHK(c, launch_ati_finish_sums(p->ati_part, p->ati_nparts, p->ati_thr, p->ati_part + p->ati_nparts, c->ati_out3_(), c->stream));
    return SARX_OK;
}
static int az_step(sarx_plan* p, bool inv, bool step_b, int S, const void* in, void* out, int q0, int nq) {
    sarx_ctx* c = p->ctx;
    const int n = p->n_az, RA = n / S;
    AzArgs a{};
    a.tw_n = c->tw_all + n;
    a.c1 = p->c1;
    a.dt = 1.0 / p->p.sample_rate_hz;
    a.t_start = p->p.t_start_fast_s;
    a.scale = 1.0f / (float)n;
    a.n_rg = p->n_rg;
    a.in = (const float2*)in; a.out = (float2*)out;
    a.q0 = q0;
    a.nt = p->az_nt;
    const bool alone = !(c->range_cus > 0 && c->range_cus < c->cus);
    const int w_plain = (alone && p->az_w_alone) ? p->az_w_alone : p->az_w;      // the same columns' arithmetic either way: bit-identical images
    if (!step_b) {
        a.tw_r = c->tw_all + RA;
        a.in_q_stride = 1; a.in_m_stride = S; a.out_q_stride = 1; a.out_m_stride = S;
        HIPCHK(c, launch_az_tile(RA, w_plain, inv, AZ_EPI_TWIDDLE, a, nq, c->stream));
    } else {
        a.tw_r = c->tw_all + S;
        a.in_q_stride = S; a.in_m_stride = 1; a.out_q_stride = 1; a.out_m_stride = RA;
        const bool look = inv && p->look_slot;
        if (look) { a.look_part = p->look_part; a.look = p->look; }
        const bool ati = inv && p->ati_s1;
        if (inv && !ati) a.max_out = reinterpret_cast<unsigned*>(p->max_slot);      // an armed ATI epilogue reads the slot (ati_thr): never reduce into it then
        if (ati) ati_args(p, a);
        HIPCHK(c, launch_az_tile(S, ati ? p->ati_w : look ? p->az_w : w_plain, inv, inv ? (ati ? AZ_EPI_SCALE_ATI : look ? AZ_EPI_SCALE_LOOK : AZ_EPI_SCALE) : AZ_EPI_PHI1, a, nq, c->stream));
    }
    return SARX_OK;
}
// the finish half of the fused multilook, after the last azimuth launch of a focus
static int look_finish(sarx_plan* p) {
    if (!p->look_slot || p->ati_s1) return SARX_OK;      // the ATI epilogue takes precedence: no look partials were written
    sarx_ctx* c = p->ctx;
    HIPCHK(c, launch_look_finish(p->look_part, p->look_slot, p->n_az / p->look, p->n_rg / p->look, p->look, c->stream));
    return SARX_OK;
}

// azimuth FFT (+epilogue) in -> out via tmp (tmp unused for single-step sizes); in is not modified
static int az_pass(sarx_plan* p, bool inv, const void* in, void* tmp, void* out) {
    sarx_ctx* c = p->ctx;
    const int n = p->n_az, S = p->az_s;
    if (S == n) {          // one tile spans the whole azimuth extent
        AzArgs a{};
        a.tw_n = c->tw_all + n;
        a.c1 = p->c1;
        a.dt = 1.0 / p->p.sample_rate_hz;
        a.t_start = p->p.t_start_fast_s;
        a.scale = 1.0f / (float)n;
        a.n_rg = p->n_rg;
        a.in = (const float2*)in; a.out = (float2*)out;
        a.nt = p->az_nt;
        a.tw_r = c->tw_all + n;
        a.in_q_stride = 0; a.in_m_stride = 1; a.out_q_stride = 0; a.out_m_stride = 1;
        const bool look = inv && p->look_slot;
        if (look) { a.look_part = p->look_part; a.look = p->look; }
        const bool ati = inv && p->ati_s1;
        if (inv && !ati) a.max_out = reinterpret_cast<unsigned*>(p->max_slot);
        if (ati) ati_args(p, a);
        HIPCHK(c, launch_az_tile(n, ati ? p->ati_w : p->az_w, inv, inv ? (ati ? AZ_EPI_SCALE_ATI : look ? AZ_EPI_SCALE_LOOK : AZ_EPI_SCALE) : AZ_EPI_PHI1, a, 1, c->stream));
        return SARX_OK;
    }
    int rc;
    if ((rc = az_step(p, inv, false, S, in, tmp, 0, S)) != SARX_OK) return rc;
    return az_step(p, inv, true, S, tmp, out, 0, n / S);
}

int sarx_csa_pass(sarx_plan* p, int pass_id, const void* d_in, void* d_out) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (!d_in || !d_out) return fail(c, SARX_ERR_INVALID, "NULL image pointer");
    int rc;
    if (p->gen) {      // any-size plans: the range passes of a direct mixed-radix line length (13200) only
        int mode = -1;
        switch (pass_id) {
            case SARX_PASS_RG_FFT_PHI2: mode = RG_FFT_PHI2; break;
            case SARX_PASS_RG_IFFT_PHI3: mode = RG_IFFT_PHI3; break;
            case SARX_PASS_RG_FUSED_23: mode = RG_FUSED; break;
            case 100: mode = RG_FFT; break;
            case 101: mode = RG_IFFT; break;
        }
        hipError_t e = hipErrorNotSupported;
        if (mode >= 0) e = general_csa_range_pass(p->gen, mode, (const float2*)d_in, (float2*)d_out, c->stream);
        else if (pass_id == SARX_PASS_AZ_FFT_PHI1 || pass_id == SARX_PASS_AZ_IFFT) {
            if (d_in == d_out) return fail(c, SARX_ERR_INVALID, "azimuth passes are out-of-place");
            if (pass_id == SARX_PASS_AZ_IFFT && (p->max_slot || p->ati_s1))
                return fail(c, SARX_ERR_UNSUPPORTED, "the per-pass azimuth IFFT of a 7199 x 13200 plan has no max-slot / ATI epilogue: switch them off or use sarx_csa_focus_dev");
            e = general_csa_az_pass(p->gen, pass_id == SARX_PASS_AZ_IFFT, (const float2*)d_in, (float2*)d_out, c->stream);
        }
        if (e == hipErrorNotSupported)
            return fail(c, SARX_ERR_UNSUPPORTED, "per-pass entry points exist for power-of-two plans, for the range passes of n_rg = 13200 "
                                                 "and for the azimuth passes of 7199 x 13200");
        HIPCHK(c, e);
        return SARX_OK;
    }
    switch (pass_id) {
        case SARX_PASS_AZ_FFT_PHI1:
        case SARX_PASS_AZ_IFFT:
            if (d_in == d_out || d_in == p->buf_b || d_out == p->buf_b)
                return fail(c, SARX_ERR_INVALID, "azimuth passes are out-of-place");
            if (p->max_slot && !p->ati_s1 && pass_id == SARX_PASS_AZ_IFFT) HIPCHK(c, hipMemsetAsync(p->max_slot, 0, MAX_SLOT_BYTES, c->stream));
            if ((rc = az_pass(p, pass_id == SARX_PASS_AZ_IFFT, d_in, p->buf_b, d_out)) != SARX_OK) return rc;
            if (pass_id == SARX_PASS_AZ_IFFT) {      // the armed epilogues of the last azimuth launch need their finish launches here too
                if ((rc = look_finish(p)) != SARX_OK) return rc;
                if ((rc = ati_finish(p)) != SARX_OK) return rc;
            }
            return SARX_OK;
        case SARX_PASS_RG_FFT_PHI2: { RangeArgs a = range_args(p, d_in, d_out); HIPCHK(c, run_range(p, RG_FFT_PHI2, a)); return SARX_OK; }
        case SARX_PASS_RG_IFFT_PHI3: { RangeArgs a = range_args(p, d_in, d_out); HIPCHK(c, run_range(p, RG_IFFT_PHI3, a)); return SARX_OK; }
        case SARX_PASS_RG_FUSED_23: { RangeArgs a = range_args(p, d_in, d_out); HIPCHK(c, run_range(p, RG_FUSED, a)); return SARX_OK; }
        case SARX_PASS_RG_FFT_PHI2_PERM:
        case SARX_PASS_RG_IFFT_PHI3_PERM: {
            if (!range_wp_supported(p->n_rg)) return fail(c, SARX_ERR_UNSUPPORTED, "the permuted-spectrum range passes exist for n_rg = 16384 only");
            RangeArgs a = range_args(p, d_in, d_out);
            HIPCHK(c, launch_range_wp(pass_id == SARX_PASS_RG_FFT_PHI2_PERM ? RG_FFT_PHI2 : RG_IFFT_PHI3, a, c->cus, c->stream));
            return SARX_OK;
        }
        case 100: { RangeArgs a = range_args(p, d_in, d_out); HIPCHK(c, run_range(p, RG_FFT, a)); return SARX_OK; }   // plain FFT (tests)
        case 101: { RangeArgs a = range_args(p, d_in, d_out); HIPCHK(c, run_range(p, RG_IFFT, a)); return SARX_OK; }  // plain IFFT (tests)
    }
    return fail(c, SARX_ERR_INVALID, "unknown pass id %d", pass_id);
}

int sarx_csa_focus_dev(sarx_plan* p, const void* d_phist, void* d_image) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (!d_phist || !d_image || d_phist == d_image) return fail(c, SARX_ERR_INVALID, "image pointers NULL or aliased");
    if (p->ati_s1 && (d_image == (const void*)p->ati_s1 || d_phist == (const void*)p->ati_s1))
        return fail(c, SARX_ERR_INVALID, "the first channel's image (sarx_csa_plan_set_ati) must not be this focus's input or output: the output buffer is scratch");
    const bool rg_major = p->flags & SARX_OUT_RG_MAJOR;
    int rc;
    // the slot is cleared and re-reduced by every focus EXCEPT one with the ATI epilogue armed: that focus is the second
    // channel's and reads the first channel's maximum from it (normally the same buffer) - clearing it there made the
    // threshold 0 and the mask pass every pixel
    if (p->max_slot && !p->ati_s1) HIPCHK(c, hipMemsetAsync(p->max_slot, 0, MAX_SLOT_BYTES, c->stream));
    if (p->gen) {
        float2* dst = rg_major ? p->buf_a : (float2*)d_image;
        HIPCHK(c, general_csa_focus(p->gen, (const float2*)d_phist, dst, c->stream));
        if ((rc = ati_finish(p)) != SARX_OK) return rc;
        if (rg_major) HIPCHK(c, launch_corner_turn(p->buf_a, (float2*)d_image, p->n_az, p->n_rg, c->stream));
        return SARX_OK;
    }
    if (p->slab_tiles > 0 && p->az_s != p->n_az && (p->flags & SARX_FUSE_RANGE)) {
        // Slab mode.  The forward transform's second step, the fused range pass and the inverse transform's first step all
        // work on the same row set when the inverse is split the other way round (its stride = the forward's tile count):
        // forward tile q writes rows q + m'*RA (m' < S), the range pass needs whole rows, inverse tile q reads rows
        // q + m*RA.  Running the three launches group of tiles by group of tiles keeps a group's rows (slab_tiles * S rows)
        // in the 256 MiB Infinity Cache between them: the image makes three HBM round trips instead of five.
        const int S = p->az_s, RA = p->n_az / S, Q = p->slab_tiles;
        float2* last = rg_major ? p->buf_a : (float2*)d_image;
        if ((rc = az_step(p, false, false, S, d_phist, d_image, 0, S)) != SARX_OK) return rc;       // forward step A, whole image
        bool marked = false;
        for (int q0 = 0; q0 < RA; q0 += Q) {
            const int nq = (q0 + Q <= RA) ? Q : RA - q0;
            if ((rc = az_step(p, false, true, S, d_image, p->buf_b, q0, nq)) != SARX_OK) return rc;
            RangeArgs a = range_args(p, p->buf_b, p->buf_b);
            a.n_az = nq * S; a.row0 = q0; a.row_inner = nq; a.row_stride = RA;
            const bool mark = !marked && p->mark_start >= 0;          // the first group's launch is the one that is timed
            if (mark) { HIPCHK(c, hipEventRecord(c->ev[p->mark_start], c->stream)); c->ev_set[p->mark_start] = true; }
            HIPCHK(c, run_range(p, RG_FUSED, a));
            if (mark && p->mark_stop >= 0) { HIPCHK(c, hipEventRecord(c->ev[p->mark_stop], c->stream)); c->ev_set[p->mark_stop] = true; marked = true; }
            if ((rc = az_step(p, true, false, RA, p->buf_b, p->buf_b, q0, nq)) != SARX_OK) return rc;   // inverse step A, stride RA
        }
        if ((rc = az_step(p, true, true, RA, p->buf_b, last, 0, S)) != SARX_OK) return rc;           // inverse step B, whole image
        if ((rc = look_finish(p)) != SARX_OK) return rc;
        if (rg_major) HIPCHK(c, launch_corner_turn(p->buf_a, (float2*)d_image, p->n_az, p->n_rg, c->stream));
        return SARX_OK;
    }
    // pass 1: azimuth FFT + Phi_1: phist -> (image as step-A scratch) -> buf_b
    if ((rc = az_pass(p, false, d_phist, d_image, p->buf_b)) != SARX_OK) return rc;
    // passes 2, 3 in place on buf_b
    if (p->mark_start >= 0) { HIPCHK(c, hipEventRecord(c->ev[p->mark_start], c->stream)); c->ev_set[p->mark_start] = true; }
    if (p->flags & SARX_FUSE_RANGE) {
        RangeArgs a = range_args(p, p->buf_b, p->buf_b);
        a.stamp = p->stamp;
        HIPCHK(c, run_range(p, RG_FUSED, a));
    } else if (range_wp_supported(p->n_rg) && (c->range_impl == 0 || c->range_impl == 4)) {
        // two launches with the spectrum in permuted order between them (range_wp.hip): one workgroup-wide exchange each
        RangeArgs a = range_args(p, p->buf_b, p->buf_b);
        HIPCHK(c, launch_range_wp(RG_FFT_PHI2, a, c->cus, c->stream));
        HIPCHK(c, launch_range_wp(RG_IFFT_PHI3, a, c->cus, c->stream));
    } else {
        RangeArgs a = range_args(p, p->buf_b, p->buf_b);
        HIPCHK(c, run_range(p, RG_FFT_PHI2, a));
        HIPCHK(c, run_range(p, RG_IFFT_PHI3, a));
    }
    if (p->mark_stop >= 0) { HIPCHK(c, hipEventRecord(c->ev[p->mark_stop], c->stream)); c->ev_set[p->mark_stop] = true; }
    // pass 4: azimuth IFFT; step A in place on buf_b, step B out to the image (or buf_a before the corner turn)
    float2* last = rg_major ? p->buf_a : (float2*)d_image;
    if (p->az_s == p->n_az) {
        if ((rc = az_pass(p, true, p->buf_b, nullptr, last)) != SARX_OK) return rc;
    } else {
        if ((rc = az_pass(p, true, p->buf_b, p->buf_b, last)) != SARX_OK) return rc;
    }
    if ((rc = look_finish(p)) != SARX_OK) return rc;
    if ((rc = ati_finish(p)) != SARX_OK) return rc;
    if (rg_major) HIPCHK(c, launch_corner_turn(p->buf_a, (float2*)d_image, p->n_az, p->n_rg, c->stream));
    return SARX_OK;
}


int sarx_csa_focus_host_c128(sarx_plan* p, const void* phist_host, void* image_host) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (!phist_host || !image_host) return fail(c, SARX_ERR_INVALID, "NULL host pointer");
    const size_t img = (size_t)p->n_az * p->n_rg * sizeof(float2);
    if (!p->h_in) { hipError_t e = hipMalloc(&p->h_in, img); if (e != hipSuccess) return fail(c, SARX_ERR_NOMEM, "hipMalloc staging: %s", hipGetErrorString(e)); }
    if (!p->h_out) { hipError_t e = hipMalloc(&p->h_out, img); if (e != hipSuccess) return fail(c, SARX_ERR_NOMEM, "hipMalloc staging: %s", hipGetErrorString(e)); }
    HIPCHK(c, staged_copy(c, p->h_in, phist_host, img, true, true));
    int rc = sarx_csa_focus_dev(p, p->h_in, p->h_out);
    if (rc != SARX_OK) return rc;
    HIPCHK(c, staged_copy(c, image_host, p->h_out, img, false));
    return SARX_OK;
}

int sarx_csa_focus_host(sarx_plan* p, const void* phist_host, void* image_host) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (!phist_host || !image_host) return fail(c, SARX_ERR_INVALID, "NULL host pointer");
    const size_t img = (size_t)p->n_az * p->n_rg * sizeof(float2);
    if (!p->h_in) { hipError_t e = hipMalloc(&p->h_in, img); if (e != hipSuccess) return fail(c, SARX_ERR_NOMEM, "hipMalloc staging: %s", hipGetErrorString(e)); }
    if (!p->h_out) { hipError_t e = hipMalloc(&p->h_out, img); if (e != hipSuccess) return fail(c, SARX_ERR_NOMEM, "hipMalloc staging: %s", hipGetErrorString(e)); }
    HIPCHK(c, staged_copy(c, p->h_in, phist_host, img, true));
    int rc = sarx_csa_focus_dev(p, p->h_in, p->h_out);
    if (rc != SARX_OK) return rc;
    HIPCHK(c, staged_copy(c, image_host, p->h_out, img, false));
    return SARX_OK;
}

// The host-array call as a pipeline (the loop of sar_batch_sim.py:303-331, the two back-to-back calls of sar_ati_dcpa_sim_csa.py:410-411):
// _begin uploads this frame while the previous frame focuses and downloads; _end waits for a frame's image.
int sarx_csa_focus_host_begin(sarx_plan* p, const void* phist_host, void* image_host, int* out_ticket) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (!phist_host || !image_host || !out_ticket) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    *out_ticket = -1;
    const int s = p->pipe_next;
    if (p->pipe_dl[s] != -1) return fail(c, SARX_ERR_INVALID, "%d frames are in flight on this plan: call sarx_csa_focus_host_end first", sarx_plan::PIPE);
    const size_t img = (size_t)p->n_az * p->n_rg * sizeof(float2);
    for (float2** b : {&p->pipe_in[s], &p->pipe_out[s]})
        if (!*b) { hipError_t e = hipMalloc(b, img); if (e != hipSuccess) return fail(c, SARX_ERR_NOMEM, "hipMalloc pipeline buffer: %s", hipGetErrorString(e)); }
    // slot s last held frame i - PIPE, whose _end has returned: nothing enqueued touches these two buffers, the upload need not wait
    // for the frame that is focusing or downloading right now
    HIPCHK(c, staged_copy(c, p->pipe_in[s], phist_host, img, true, false, /*ordered=*/false));
    int rc = sarx_csa_focus_dev(p, p->pipe_in[s], p->pipe_out[s]);
    if (rc != SARX_OK) return rc;
    if (is_page_locked(image_host)) {
        int slot = -1;
        if ((rc = sarx_memcpy_d2h_begin(c, image_host, p->pipe_out[s], img, &slot)) != SARX_OK) {
            hipStreamSynchronize(c->stream);        // the focus is enqueued but the slot stays free: nothing may still touch its buffers
            return rc;
        }
        p->pipe_dl[s] = slot; p->pipe_host[s] = nullptr;
    } else {                       // a pageable result cannot be the target of an asynchronous DMA: _end downloads it (staged, blocking)
        p->pipe_dl[s] = sarx_ctx::DL_SLOTS; p->pipe_host[s] = image_host;
    }
    p->pipe_next = (s + 1) % sarx_plan::PIPE;
    *out_ticket = s;
    return SARX_OK;
}
int sarx_csa_focus_host_end(sarx_plan* p, int ticket) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (ticket < 0 || ticket >= sarx_plan::PIPE || p->pipe_dl[ticket] == -1) return fail(c, SARX_ERR_INVALID, "ticket %d is not in flight", ticket);
    const int slot = p->pipe_dl[ticket];
    p->pipe_dl[ticket] = -1;
    if (slot == sarx_ctx::DL_SLOTS) {
        const size_t img = (size_t)p->n_az * p->n_rg * sizeof(float2);
        HIPCHK(c, staged_copy(c, p->pipe_host[ticket], p->pipe_out[ticket], img, false));      // ordered: waits for the focus
        return SARX_OK;
    }
    return sarx_memcpy_d2h_end(c, slot);
}

// ---- Range-Doppler focus ---------------------------------------------------------------
struct sarx_rda_plan {
    sarx_ctx* ctx = nullptr;
    Rda* r = nullptr;
    int n_r = 0, n_p = 0;
    float2* d_in = nullptr;
};

static int sarx_rda_plan_create_impl(sarx_ctx* c, int n_ranges, int n_pulses, const sarx_radar_params* prm, sarx_rda_plan** out) {
    NEED_CTX(c);
    if (!out || !prm) return fail(c, SARX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_ranges < 2 || n_pulses < 2 || n_ranges > 2 * TW_MAX || n_pulses > 2 * TW_MAX)
        return fail(c, SARX_ERR_UNSUPPORTED, "n_ranges=%d n_pulses=%d: sizes must be in [2, %d]", n_ranges, n_pulses, 2 * TW_MAX);
    if (!(prm->sample_rate_hz > 0) || !(prm->prf_hz > 0) || !(prm->platform_speed_mps > 0) || !(prm->wavelength_m > 0) ||
        !(prm->pulse_width_s > 0))
        return fail(c, SARX_ERR_INVALID, "radar parameters must be positive");
    std::string err;
    Rda* r = rda_create(n_ranges, n_pulses, prm, c->tw_all, err, c->cus);
    if (!r) return fail(c, SARX_ERR_UNSUPPORTED, "n_ranges=%d n_pulses=%d: %s", n_ranges, n_pulses, err.c_str());
    sarx_rda_plan* p = new sarx_rda_plan();
    p->ctx = c; p->r = r; p->n_r = n_ranges; p->n_p = n_pulses;
    hipError_t e = hipMalloc(&p->d_in, (size_t)n_ranges * n_pulses * sizeof(float2));
    if (e != hipSuccess) { rda_destroy(r); delete p; return fail(c, SARX_ERR_NOMEM, "hipMalloc: %s", hipGetErrorString(e)); }
    *out = p;
    return SARX_OK;
}
int sarx_rda_plan_create(sarx_ctx* c, int n_ranges, int n_pulses, const sarx_radar_params* prm, sarx_rda_plan** out) {
    return guarded(c, [&] { return sarx_rda_plan_create_impl(c, n_ranges, n_pulses, prm, out); });
}
int sarx_rda_plan_destroy(sarx_rda_plan* p) {
    if (!p) return SARX_OK;
    hipSetDevice(p->ctx->device);
    sync_all_lanes(p->ctx);
    rda_destroy(p->r);
    hipFree(p->d_in);
    delete p;
    return SARX_OK;
}
int sarx_rda_focus_host2(sarx_rda_plan* p, const void* phist, float* mag, void* pc, void* rd, void* rc, void* ac) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (!phist || !mag) return fail(c, SARX_ERR_INVALID, "NULL host pointer");
    const size_t px = (size_t)p->n_r * p->n_p;
    HIPCHK(c, staged_copy(c, p->d_in, phist, px * sizeof(float2), true));
    HIPCHK(c, rda_focus(p->r, p->d_in, c->stream, nullptr, rc != nullptr, ac != nullptr));
    HIPCHK(c, staged_copy(c, mag, rda_mag(p->r), px * sizeof(float), false));
    void* outs[4] = {pc, rd, rc, ac};
    for (int i = 0; i < 4; ++i)
        if (outs[i]) HIPCHK(c, staged_copy(c, outs[i], rda_stage(p->r, i), px * sizeof(float2), false));
    return SARX_OK;
}
int sarx_rda_focus_host(sarx_rda_plan* p, const void* phist, float* mag, void* pc, void* rd, void* rc) {
    return sarx_rda_focus_host2(p, phist, mag, pc, rd, rc, nullptr);
}
int sarx_rda_focus_dev2(sarx_rda_plan* p, const void* d_phist, float* d_mag, void* d_pc, void* d_rd, void* d_rc, void* d_ac) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (!d_phist || !d_mag) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    const size_t px = (size_t)p->n_r * p->n_p;
    HIPCHK(c, rda_focus(p->r, (const float2*)d_phist, c->stream, d_mag, d_rc != nullptr, d_ac != nullptr));   // magnitude written in place by the last launch
    void* outs[4] = {d_pc, d_rd, d_rc, d_ac};
    for (int i = 0; i < 4; ++i)
        if (outs[i]) HIPCHK(c, hipMemcpyAsync(outs[i], rda_stage(p->r, i), px * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
    return SARX_OK;
}
int sarx_rda_focus_dev(sarx_rda_plan* p, const void* d_phist, float* d_mag, void* d_pc, void* d_rd, void* d_rc) {
    return sarx_rda_focus_dev2(p, d_phist, d_mag, d_pc, d_rd, d_rc, nullptr);
}
int sarx_rda_axes(const sarx_rda_plan* p, double* range_centered, double* cross_range, double* doppler) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    rda_axes(p->r, range_centered, cross_range, doppler);
    return SARX_OK;
}

// ---- ATI / DPCA ------------------------------------------------------------------
static int ati_dpca_impl(sarx_ctx* c, const void* s1, const void* s2, size_t n, double cal_phase, const sarx_ati_outputs* o,
                         double* max_mag, double* sum2, const float* d_max, float mask_frac);
int sarx_ati_dpca_dev(sarx_ctx* c, const void* s1, const void* s2, size_t n, double cal_phase,
                      const sarx_ati_outputs* o, double* max_mag, double* sum2) {
    return ati_dpca_impl(c, s1, s2, n, cal_phase, o, max_mag, sum2, nullptr, 0.f);
}
int sarx_ati_dpca_masked_dev(sarx_ctx* c, const void* s1, const void* s2, size_t n, double cal_phase, const float* d_max,
                             float mask_frac, const sarx_ati_outputs* o) {
    NEED_CTX(c);
    if (!d_max) return fail(c, SARX_ERR_INVALID, "d_max is NULL (sarx_csa_plan_set_max_slot provides it)");
    return ati_dpca_impl(c, s1, s2, n, cal_phase, o, nullptr, nullptr, d_max, mask_frac);
}
static int ati_dpca_impl(sarx_ctx* c, const void* s1, const void* s2, size_t n, double cal_phase, const sarx_ati_outputs* o,
                         double* max_mag, double* sum2, const float* d_max, float mask_frac) {
    NEED_CTX(c);
    if (!s1 || !s2 || !o || !o->ati_phase || !o->slc1_mag || !o->dpca_mag) return fail(c, SARX_ERR_INVALID, "NULL required pointer");
    if (n == 0) { if (max_mag) *max_mag = 0; if (sum2) sum2[0] = sum2[1] = 0; return SARX_OK; }
    {   // the kernel moves 16 bytes per lane and plane
        const void* ptrs[] = {s1, s2, o->ati_phase, o->slc1_mag, o->dpca_mag, o->ati_interf, o->dpca_diff, o->slc2_mag,
                              o->slc1_phase, o->slc2_phase, o->dpca_phase};
        for (const void* q : ptrs)
            if (((uintptr_t)q) & 15) return fail(c, SARX_ERR_INVALID, "ATI/DPCA buffers must be 16-byte aligned");
    }
    AtiArgs a{};
    a.s1 = (const float2*)s1; a.s2 = (const float2*)s2; a.n = n;
    a.cal_c = (float)cos(cal_phase); a.cal_s = (float)sin(cal_phase);
    a.ati_phase = o->ati_phase; a.mag1 = o->slc1_mag; a.dpca_mag = o->dpca_mag;
    a.interf = (float2*)o->ati_interf; a.diff = (float2*)o->dpca_diff;
    a.mag2 = o->slc2_mag; a.ph1 = o->slc1_phase; a.ph2 = o->slc2_phase; a.dpca_phase = o->dpca_phase;
    a.part_max = c->ati_part_max_(); a.part_sum = c->ati_part_sum_();
    a.thr_max = d_max; a.mask_frac = mask_frac;
    // the two images are read for the last time here: nontemporal loads once they are too large to still be cached
    // (0.341 -> 0.329 ms at 8192^2); SARX_ATI_NT=0/1 overrides
    { static const int nt = [] { const char* e = getenv("SARX_ATI_NT"); return e ? atoi(e) : -1; }(); a.nt = nt < 0 ? n >= ((size_t)1 << 25) : nt != 0; }
    HIPCHK(c, launch_ati_dpca(a, c->stream));
    HIPCHK(c, launch_ati_finish(c->ati_part_max_(), c->ati_part_sum_(), ati_blocks(n), c->ati_out3_(), c->stream));
    if (max_mag || sum2) {
        double h[3];
        HIPCHK(c, hipMemcpyAsync(h, c->ati_out3_(), sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (max_mag) *max_mag = h[0];
        if (sum2) { sum2[0] = h[1]; sum2[1] = h[2]; }
    }
    return SARX_OK;
}

int sarx_ati_stats(sarx_ctx* c, double* max_mag, double* sum2) {
    NEED_CTX(c);
    double h[3];
    HIPCHK(c, hipMemcpyAsync(h, c->ati_out3_(), sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (max_mag) *max_mag = h[0];
    if (sum2) { sum2[0] = h[1]; sum2[1] = h[2]; }
    return SARX_OK;
}

int sarx_mask_phase_frac_dev(sarx_ctx* c, const float* phase, const float* mag, size_t n, float frac, float* out) {
    NEED_CTX(c);
    if (!phase || !mag || !out) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (n == 0) return SARX_OK;
    HIPCHK(c, launch_mask_phase_frac(phase, mag, n, frac, c->ati_out3_(), out, c->stream));
    return SARX_OK;
}

int sarx_magnitude_dev(sarx_ctx* c, const void* in, float* out, size_t n) {
    NEED_CTX(c);
    if (!in || !out) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (n) HIPCHK(c, launch_magnitude((const float2*)in, out, n, c->stream));
    return SARX_OK;
}

int sarx_max_abs_f32_dev(sarx_ctx* c, const float* x, size_t n, float* d_max) {
    NEED_CTX(c);
    if (!x || !d_max) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (n) HIPCHK(c, launch_max_abs_f32(x, n, d_max, c->stream));
    return SARX_OK;
}

int sarx_mask_phase_dev(sarx_ctx* c, const float* phase, const float* mag, size_t n, float thr, float* out) {
    NEED_CTX(c);
    if (!phase || !mag || !out) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (n == 0) return SARX_OK;
    HIPCHK(c, launch_mask_phase(phase, mag, n, thr, out, c->stream));
    return SARX_OK;
}

int sarx_corner_turn_dev(sarx_ctx* c, const void* in, void* out, int rows, int cols) {
    NEED_CTX(c);
    if (!in || !out || in == out || rows <= 0 || cols <= 0) return fail(c, SARX_ERR_INVALID, "bad corner-turn arguments");
    HIPCHK(c, launch_corner_turn((const float2*)in, (float2*)out, rows, cols, c->stream));
    return SARX_OK;
}

int sarx_multilook_dev(sarx_ctx* c, const void* in, float* out, int rows, int cols, int looks) {
    NEED_CTX(c);
    if (!in || !out) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (looks < 1 || looks > 512 || (looks & (looks - 1)) || rows % looks || cols % looks || (cols & 1))
        return fail(c, SARX_ERR_UNSUPPORTED, "looks=%d must be a power of two <= 512 dividing rows=%d and cols=%d", looks, rows, cols);
    HIPCHK(c, launch_multilook((const float2*)in, out, rows, cols, looks, c->stream));
    return SARX_OK;
}

int sarx_echo_synth_dev(sarx_ctx* c, const double* tau_pb, const float* amp, const double* t_fast, int n_pulses,
                        int n_targets, int n_samples, double kr, double t_p, void* raw, int accumulate) {
    NEED_CTX(c);
    if (!tau_pb || !amp || !t_fast || !raw) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (n_pulses <= 0 || n_targets <= 0 || n_samples <= 0 || n_pulses > 65535)
        return fail(c, SARX_ERR_INVALID, "echo sizes must be positive (n_pulses <= 65535 per call)");
    EchoArgs a{};
    a.tau_pb = (const double2*)tau_pb; a.amp = amp; a.t_fast = t_fast; a.out = (float2*)raw;
    a.kr = kr; a.t_p = t_p; a.u_off = 0.5 * t_p; a.n_pulses = n_pulses; a.n_targets = n_targets; a.n_samples = n_samples;
    a.accumulate = accumulate != 0;
    HIPCHK(c, launch_echo_synth(a, c->stream));
    return SARX_OK;
}
int sarx_echo_geometry_dev(sarx_ctx* c, int model, int n_pulses, int n_targets, const double* tgt_pos, const double* tgt_vel,
                           const double* t_pulse, const double* tx_pos, const double* aux, const double* rcs, double c_light,
                           double fc, double l_ant, double wavelength, double* tau_pb, float* amp_pt) {
    NEED_CTX(c);
    if (model < 0 || model > 2) return fail(c, SARX_ERR_INVALID, "echo model must be 0, 1 or 2");
    if (n_pulses <= 0 || n_targets <= 0 || n_pulses > 65535) return fail(c, SARX_ERR_INVALID, "echo sizes must be positive (n_pulses <= 65535 per call)");
    if (!tgt_pos || !tx_pos || !tau_pb) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (model != 0 && (!tgt_vel || !t_pulse || !aux)) return fail(c, SARX_ERR_INVALID, "models 1 and 2 need target velocity, pulse times and aux");
    if (model == 0 && tgt_vel && !t_pulse) return fail(c, SARX_ERR_INVALID, "moving targets need the pulse times");
    if (model == 2 && (!rcs || !amp_pt || !(wavelength > 0))) return fail(c, SARX_ERR_INVALID, "model 2 needs rcs, amp_pt and the wavelength");
    if (!(c_light > 0) || !(fc > 0)) return fail(c, SARX_ERR_INVALID, "C and FC must be positive");
    EchoGeoArgs a{};
    a.model = model; a.n_pulses = n_pulses; a.n_targets = n_targets;
    a.tgt_pos = tgt_pos; a.tgt_vel = tgt_vel; a.t_pulse = t_pulse; a.tx_pos = tx_pos; a.aux = aux; a.rcs = rcs;
    a.c = c_light; a.fc = fc; a.l_ant = l_ant; a.lambda = wavelength;
    a.tau_pb = (double2*)tau_pb; a.amp_pt = amp_pt;
    HIPCHK(c, launch_echo_geometry(a, c->stream));
    return SARX_OK;
}
int sarx_echo_spotlight_dev(sarx_ctx* c, const double* tau_pb, const float* amp_pt, const double* t_fast, int n_pulses,
                            int n_targets, int n_samples, double kr, double t_p, void* raw) {
    NEED_CTX(c);
    if (!tau_pb || !amp_pt || !t_fast || !raw) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (n_pulses <= 0 || n_targets <= 0 || n_samples <= 0 || n_pulses > 65535)
        return fail(c, SARX_ERR_INVALID, "echo sizes must be positive (n_pulses <= 65535 per call)");
    EchoArgs a{};
    a.tau_pb = (const double2*)tau_pb; a.amp_pt = amp_pt; a.t_fast = t_fast; a.out = (float2*)raw;
    a.kr = kr; a.t_p = t_p; a.u_off = 0.0; a.n_pulses = n_pulses; a.n_targets = n_targets; a.n_samples = n_samples;
    HIPCHK(c, launch_echo_synth(a, c->stream));
    return SARX_OK;
}

// ---- time-domain back-projection ---------------------------------------------------------
struct sarx_tdbp_plan {
    sarx_ctx* ctx = nullptr;
    Tdbp* t = nullptr;
    int n_p = 0, n_s = 0, nx = 0, ny = 0;
    float2* d_raw = nullptr;       // staging for the host entry point
};

static int sarx_tdbp_plan_create_impl(sarx_ctx* c, int n_pulses, int num_samples, int nx, int ny, const sarx_tdbp_params* k,
                          sarx_tdbp_plan** out) {
    NEED_CTX(c);
    if (!out || !k) return fail(c, SARX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_pulses < 1 || num_samples < 2 || nx < 1 || ny < 1 || nx > 65536 || ny > 65536)
        return fail(c, SARX_ERR_INVALID, "n_pulses=%d num_samples=%d nx=%d ny=%d: sizes must be positive", n_pulses, num_samples, nx, ny);
    if (!(k->c > 0) || !(k->fc > 0) || !(k->fs > 0) || !(k->t_p > 0) || !(k->k_rate != 0))
        return fail(c, SARX_ERR_INVALID, "TDBP constants must be positive");
    std::string err;
    Tdbp* t = tdbp_create(n_pulses, num_samples, nx, ny, k, c->tw_all, err);
    if (!t) return fail(c, SARX_ERR_UNSUPPORTED, "tdbp plan: %s", err.c_str());
    sarx_tdbp_plan* p = new sarx_tdbp_plan();
    p->ctx = c; p->t = t; p->n_p = n_pulses; p->n_s = num_samples; p->nx = nx; p->ny = ny;
    *out = p;
    return SARX_OK;
}
int sarx_tdbp_plan_create(sarx_ctx* c, int n_pulses, int num_samples, int nx, int ny, const sarx_tdbp_params* k,
                          sarx_tdbp_plan** out) {
    return guarded(c, [&] { return sarx_tdbp_plan_create_impl(c, n_pulses, num_samples, nx, ny, k, out); });
}
int sarx_tdbp_plan_destroy(sarx_tdbp_plan* p) {
    if (!p) return SARX_OK;
    hipSetDevice(p->ctx->device);
    sync_all_lanes(p->ctx);
    tdbp_destroy(p->t);
    hipFree(p->d_raw);
    delete p;
    return SARX_OK;
}
int sarx_tdbp_focus_dev(sarx_tdbp_plan* p, const void* d_raw, const double* pos, const double* vel, const double* t_pulses,
                        double t_start, const double* vel_focus, double scene_size, void* d_image) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (!d_raw || !pos || !vel || !t_pulses || !vel_focus || !d_image) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (!(scene_size > 0)) return fail(c, SARX_ERR_INVALID, "scene_size must be positive");
    HIPCHK(c, tdbp_focus(p->t, (const float2*)d_raw, pos, vel, t_pulses, t_start, vel_focus, scene_size, false, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_image, tdbp_image(p->t), (size_t)p->nx * p->ny * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    return SARX_OK;
}
int sarx_tdbp_last_window(const sarx_tdbp_plan* p, int* lo, int* hi) {
    if (!p || !lo || !hi) return fail(nullptr, SARX_ERR_INVALID, "NULL argument");
    tdbp_window(p->t, lo, hi);
    return SARX_OK;
}
int sarx_tdbp_focus_host(sarx_tdbp_plan* p, const void* raw, const double* pos, const double* vel, const double* t_pulses,
                         double t_start, const double* vel_focus, double scene_size, void* image, void* range_compressed) {
    if (!p) return fail(nullptr, SARX_ERR_INVALID, "plan is NULL");
    sarx_ctx* c = p->ctx;
    hipSetDevice(c->device);
    if (!raw || !pos || !vel || !t_pulses || !vel_focus || !image) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (!(scene_size > 0)) return fail(c, SARX_ERR_INVALID, "scene_size must be positive");
    const size_t n = (size_t)p->n_p * p->n_s;
    if (!p->d_raw) HIPCHK(c, hipMalloc(&p->d_raw, n * sizeof(float2)));
    HIPCHK(c, staged_copy(c, p->d_raw, raw, n * sizeof(float2), true));
    HIPCHK(c, tdbp_focus(p->t, p->d_raw, pos, vel, t_pulses, t_start, vel_focus, scene_size, range_compressed != nullptr, c->stream));
    HIPCHK(c, hipMemcpyAsync(image, tdbp_image(p->t), (size_t)p->nx * p->ny * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
    if (range_compressed) HIPCHK(c, hipMemcpyAsync(range_compressed, tdbp_rc(p->t), n * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SARX_OK;
}

int sarx_fill_noise_c64(sarx_ctx* c, void* buf, size_t n, uint64_t seed) {
    NEED_CTX(c);
    if (!buf) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (n) HIPCHK(c, launch_fill_noise((float2*)buf, n, seed, c->stream));
    return SARX_OK;
}

int sarx_add_ocean_noise_dev(sarx_ctx* c, void* buf, size_t n, double noise_std, double clutter_power, double k_nu,
                             uint64_t seed) {
    NEED_CTX(c);
    if (!buf) return fail(c, SARX_ERR_INVALID, "NULL pointer");
    if (!(noise_std >= 0) || !(clutter_power >= 0) || (clutter_power > 0 && !(k_nu > 0)))
        return fail(c, SARX_ERR_INVALID, "noise_std, clutter_power must be >= 0 and k_nu > 0");
    if (n) HIPCHK(c, launch_ocean_noise((float2*)buf, n, (float)noise_std, (float)clutter_power, (float)k_nu, seed, c->stream));
    return SARX_OK;
}
int sarx_add_ocean_noise_rel_dev(sarx_ctx* c, void* buf, size_t n, int ref_is_max, double snr_lin, double scr_lin, double k_nu,
                                 uint64_t seed) {
    NEED_CTX(c);
    if (!buf || !n) return fail(c, SARX_ERR_INVALID, "empty buffer");
    if (!(snr_lin > 0) || !(scr_lin >= 0) || (scr_lin > 0 && !(k_nu > 0)))
        return fail(c, SARX_ERR_INVALID, "snr_lin must be > 0, scr_lin >= 0 (0 = thermal noise only) and k_nu > 0");
    double* d_part = c->power_part_all + (size_t)c->cur_lane * sarx_ctx::POWER_STRIDE;
    float* d_levels = reinterpret_cast<float*>(d_part + 2048);
    HIPCHK(c, launch_power_stats((const float2*)buf, n, d_part, 1024, c->stream));
    HIPCHK(c, launch_noise_levels(d_part, 1024, n, ref_is_max != 0, snr_lin, scr_lin, d_levels, c->stream));
    HIPCHK(c, launch_ocean_noise((float2*)buf, n, 0.f, 0.f, (float)k_nu, seed, c->stream, d_levels));
    return SARX_OK;
}
static int sarx_power_stats_dev_impl(sarx_ctx* c, const void* buf, size_t n, double* max_abs2, double* mean_abs2) {
    NEED_CTX(c);
    if (!buf || !n) return fail(c, SARX_ERR_INVALID, "empty buffer");
    const int blocks = 1024;
    double* d_part = c->power_part_all + (size_t)c->cur_lane * sarx_ctx::POWER_STRIDE;
    hipError_t e = launch_power_stats((const float2*)buf, n, d_part, blocks, c->stream);
    std::vector<double> part(2 * blocks);
    if (e == hipSuccess) e = hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    HIPCHK(c, e);
    double sum = 0.0, mx = 0.0;
    for (int b = 0; b < blocks; ++b) { sum += part[2 * b]; if (part[2 * b + 1] > mx) mx = part[2 * b + 1]; }
    if (max_abs2) *max_abs2 = mx;
    if (mean_abs2) *mean_abs2 = sum / (double)n;
    return SARX_OK;
}
int sarx_power_stats_dev(sarx_ctx* c, const void* buf, size_t n, double* max_abs2, double* mean_abs2) {
    return guarded(c, [&] { return sarx_power_stats_dev_impl(c, buf, n, max_abs2, mean_abs2); });
}

// ---- RCCL ------------------------------------------------------------------------
int sarx_comm_unique_id(void* id_out) {
    if (!id_out) return fail(nullptr, SARX_ERR_INVALID, "id_out is NULL");
    std::string err;
    if (!load_rccl(err)) return fail(nullptr, SARX_ERR_COMM, "%s", err.c_str());
    static_assert(sizeof(ncclUniqueId) == SARX_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, SARX_ERR_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    memcpy(id_out, &id, sizeof id);
    return SARX_OK;
}
int sarx_rccl_info(char* path, size_t path_len, int* version, int* header_version) {
    std::string err;
    if (!load_rccl(err)) return fail(nullptr, SARX_ERR_COMM, "%s", err.c_str());
    if (path && path_len) snprintf(path, path_len, "%s", g_rccl.path.c_str());
    if (version) *version = g_rccl.version;
    if (header_version) *header_version = NCCL_VERSION_CODE;
    return SARX_OK;
}
int sarx_comm_init(sarx_ctx* c, const void* id, int n_ranks, int rank) {
    NEED_CTX(c);
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, SARX_ERR_INVALID, "bad comm arguments");
    std::string err;
    if (!load_rccl(err)) return fail(c, SARX_ERR_COMM, "%s", err.c_str());
    if (c->comm) return fail(c, SARX_ERR_COMM, "communicator already initialised");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, n_ranks, uid, rank);
    if (r != ncclSuccess) { c->comm = nullptr; return fail(c, SARX_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"); }
    c->n_ranks = n_ranks; c->rank = rank;
    return SARX_OK;
}
int sarx_allgather_dev(sarx_ctx* c, const void* send, void* recv, size_t bytes_per_rank) {
    NEED_CTX(c);
    if (!c->comm) return fail(c, SARX_ERR_COMM, "communicator not initialised");
    if (!send || !recv || (bytes_per_rank & 3)) return fail(c, SARX_ERR_INVALID, "bad all-gather arguments");
    HIPCHK(c, hipEventRecord(c->comm_fence, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->comm_fence, 0));
    ncclResult_t r = g_rccl.AllGather(send, recv, bytes_per_rank / 4, ncclFloat32, c->comm, c->comm_stream);
    if (r != ncclSuccess) return fail(c, SARX_ERR_COMM, "ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return SARX_OK;
}
int sarx_allreduce_max_dev(sarx_ctx* c, float* d_buf, size_t count) {
    NEED_CTX(c);
    if (!c->comm) return fail(c, SARX_ERR_COMM, "communicator not initialised");
    if (!d_buf || !count) return fail(c, SARX_ERR_INVALID, "bad all-reduce arguments");
    HIPCHK(c, hipEventRecord(c->comm_fence, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->comm_fence, 0));
    ncclResult_t r = g_rccl.AllReduce(d_buf, d_buf, count, ncclFloat32, ncclMax, c->comm, c->comm_stream);
    if (r != ncclSuccess) return fail(c, SARX_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return SARX_OK;
}
int sarx_comm_sync(sarx_ctx* c) { NEED_CTX(c); HIPCHK(c, hipStreamSynchronize(c->comm_stream)); return SARX_OK; }
int sarx_comm_fence_compute(sarx_ctx* c) {
    NEED_CTX(c);
    HIPCHK(c, hipEventRecord(c->comm_done, c->comm_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->comm_done, 0));
    return SARX_OK;
}
int sarx_comm_mark(sarx_ctx* c, int slot) {
    NEED_CTX(c);
    if (slot < 0 || slot >= 4) return fail(c, SARX_ERR_INVALID, "comm mark slot %d out of range [0,4)", slot);
    HIPCHK(c, hipEventRecord(c->comm_mark[slot], c->comm_stream));
    c->comm_mark_set[slot] = true;
    return SARX_OK;
}
int sarx_comm_wait_mark(sarx_ctx* c, int slot) {
    NEED_CTX(c);
    if (slot < 0 || slot >= 4) return fail(c, SARX_ERR_INVALID, "comm mark slot %d out of range [0,4)", slot);
    if (c->comm_mark_set[slot]) HIPCHK(c, hipStreamWaitEvent(c->stream, c->comm_mark[slot], 0));
    return SARX_OK;
}
int sarx_comm_destroy(sarx_ctx* c) {
    NEED_CTX(c);
    if (c->comm) { hipStreamSynchronize(c->comm_stream); g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    return SARX_OK;
}


#ifdef SARX_TESTING
// ---- sanitizer-build hook (make asan; never part of libsarx.so) -----------------------------------------------------------------
// staged_copy on a stand-in context with host stand-ins for the runtime: "device" memory is host memory, a DMA is a memcpy, streams
// and events are opaque tokens.  no_thread_mask: bit i = share i's thread "cannot be started" (the inline-share path);
// fail_at >= 0: the fail_at-th copy call returns an error (every thread must still be joined, the error returned);
// page_locked != 0: the host side counts as page-locked (the one-DMA path).  Returns staged_copy's hipError_t as an int.
static std::atomic<int> t_copy_calls{0};
static int t_fail_at = -1;
static unsigned t_no_thread = 0;
static int t_locked = 0;
static std::atomic<int> t_tokens{0};
int sarx_test_staged_copy(void* dst, const void* src, size_t bytes, int to_device, int narrow, int ordered, unsigned no_thread_mask,
                          int fail_at, int page_locked, int up_streams, int* threads_inline) {
    CopyOps saved = g_ops;
    t_copy_calls = 0; t_fail_at = fail_at; t_no_thread = no_thread_mask; t_locked = page_locked;
    g_ops.memcpy_async = [](void* d, const void* s2, size_t n, hipMemcpyKind, hipStream_t st) {
        if (!st) return hipErrorInvalidHandle;
        if (t_copy_calls.fetch_add(1) == t_fail_at) return hipErrorUnknown;
        memcpy(d, s2, n);
        return hipSuccess;
    };
    g_ops.stream_sync = [](hipStream_t st) { return st ? hipSuccess : hipErrorInvalidHandle; };
    g_ops.stream_create = [](hipStream_t* st, unsigned) { *st = (hipStream_t)(uintptr_t)(0x1000 + 16 * t_tokens.fetch_add(1)); return hipSuccess; };
    g_ops.event_create = [](hipEvent_t* ev, unsigned) { *ev = (hipEvent_t)(uintptr_t)(0x100000 + 16 * t_tokens.fetch_add(1)); return hipSuccess; };
    g_ops.event_record = [](hipEvent_t ev, hipStream_t st) { return (ev && st) ? hipSuccess : hipErrorInvalidHandle; };
    g_ops.event_sync = [](hipEvent_t ev) { return ev ? hipSuccess : hipErrorInvalidHandle; };
    g_ops.host_alloc = [](void** p2, size_t n, unsigned) { *p2 = malloc(n); return *p2 ? hipSuccess : hipErrorOutOfMemory; };
    g_ops.set_device = [](int) { return hipSuccess; };
    g_ops.page_locked = [](const void*) { return t_locked != 0; };
    g_ops.may_start_thread = [](int i) { return !((t_no_thread >> i) & 1u); };
    int rc;
    {
        sarx_ctx c;
        c.device = 0;
        c.stream = (hipStream_t)(uintptr_t)0x10;
        c.lane[0] = c.stream;
        if (up_streams >= 1 && up_streams <= sarx_ctx::COPY_THREADS) c.up_streams = up_streams;
        rc = (int)staged_copy(&c, dst, src, bytes, to_device != 0, narrow != 0, ordered != 0);
        if (threads_inline) { int k = 0; for (int i = 0; i < sarx_ctx::COPY_THREADS; ++i) k += (no_thread_mask >> i) & 1u; *threads_inline = k; }
        for (int i = 0; i < sarx_ctx::COPY_THREADS; ++i) free(c.pin[i]);
    }
    g_ops = saved;
    return rc;
}
size_t sarx_test_copy_chunk(void) { return sarx_ctx::COPY_CHUNK; }
int sarx_test_copy_threads(void) { return sarx_ctx::COPY_THREADS; }
int sarx_test_guard(int what) {      // the exception guard of the allocating entry points: 0 ok, 1 bad_alloc, 2 any other exception
    return guarded(nullptr, [&]() -> int {
        if (what == 1) throw std::bad_alloc();
        if (what == 2) throw std::runtime_error("x");
        return SARX_OK;
    });
}
#endif

}  // extern "C"
