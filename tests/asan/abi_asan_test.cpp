// Host-side sanitizer test of the C ABI (sanitizers on the CPU build only; `make asan` in csrc/).
//
// Linked against libsarx_asan.so = every translation unit of libsarx built with -fsanitize=address,undefined for the HOST pass
// (-fno-gpu-sanitize: device code unchanged) and -DSARX_TESTING.  Runs where there is no GPU:
//   1. every entry point include/sarx.h declares is called with the arguments a careless caller would pass (NULL context, NULL
//      plan, NULL pointers): each must return an error code - never crash - and sarx_last_error must name it;
//   2. the no-device path: sarx_init fails with a message that says there is no CPU fallback;
//   3. staged_copy (the eight-thread staged host transfer behind sarx_memcpy_* and the *_host entry points) on a stand-in context
//      with host stand-ins for the runtime: chunking at every boundary, both directions, the complex128 -> complex64 narrowing,
//      threads that "cannot be started" (inline shares), a copy that fails in the middle (all threads joined, error returned),
//      the page-locked one-DMA path, one / two / eight upload streams;
//   4. the exception guard of the allocating entry points.
// Exit code 0 and no sanitizer report = pass (tests/test_asan.py).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/sarx.h"

extern "C" {
int sarx_test_staged_copy(void* dst, const void* src, size_t bytes, int to_device, int narrow, int ordered, unsigned no_thread_mask,
                          int fail_at, int page_locked, int up_streams, int* threads_inline);
size_t sarx_test_copy_chunk(void);
int sarx_test_copy_threads(void);
int sarx_test_guard(int what);
}

static int failures = 0;
#define CHECK(cond)                                                                 \
    do {                                                                            \
        if (!(cond)) { ++failures; fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static void argument_checking() {
    int n = -1;
    CHECK(sarx_version() == SARX_VERSION);
    CHECK(sarx_device_count(nullptr) == SARX_ERR_INVALID);
    CHECK(strstr(sarx_last_error(nullptr), "NULL") != nullptr);
    int rc = sarx_device_count(&n);
    CHECK(rc == SARX_OK || rc == SARX_ERR_DEVICE);
    sarx_ctx* ctx = (sarx_ctx*)(uintptr_t)0x1;
    CHECK(sarx_init(0, nullptr) == SARX_ERR_INVALID);
    rc = sarx_init(0, &ctx);
    if (rc != SARX_OK) {                          // this box has no GPU: the message must say so, and there is no fallback
        CHECK(ctx == nullptr);
        CHECK(rc == SARX_ERR_DEVICE || rc == SARX_ERR_UNSUPPORTED);
        CHECK(strlen(sarx_last_error(nullptr)) > 10);
        CHECK(rc != SARX_ERR_DEVICE || strstr(sarx_last_error(nullptr), "no CPU fallback") != nullptr);
    } else {
        CHECK(sarx_destroy(ctx) == SARX_OK);      // (a GPU box: the rest still runs with ctx = NULL)
    }
    CHECK(sarx_destroy(nullptr) == SARX_OK);
    // every call below gets a NULL context or a NULL plan: an error code, a message, no crash
    void* p = nullptr;
    char buf[64];
    float f = 0;
    double d[4] = {0, 0, 0, 0};
    int i = 0;
    uint64_t u = 0;
    sarx_radar_params prm{};
    sarx_plan* plan = nullptr;
    sarx_rda_plan* rplan = nullptr;
    sarx_tdbp_plan* tplan = nullptr;
    sarx_ati_outputs outs{};
    sarx_tdbp_params tk{};
    CHECK(sarx_device_info(nullptr, buf, sizeof buf, &i, &u, buf, sizeof buf) != SARX_OK);
    CHECK(sarx_malloc(nullptr, 16, &p) != SARX_OK);
    CHECK(sarx_free(nullptr, nullptr) != SARX_OK);
    CHECK(sarx_host_alloc(nullptr, 16, &p) != SARX_OK);
    CHECK(sarx_host_free(nullptr, nullptr) != SARX_OK);
    CHECK(sarx_memcpy_h2d(nullptr, buf, buf, 1) != SARX_OK);
    CHECK(sarx_memcpy_d2h(nullptr, buf, buf, 1) != SARX_OK);
    CHECK(sarx_memcpy_d2d(nullptr, buf, buf, 1) != SARX_OK);
    CHECK(sarx_memcpy_h2d_unordered(nullptr, buf, buf, 1) != SARX_OK);
    CHECK(sarx_memcpy_h2d_lane(nullptr, buf, buf, 1) != SARX_OK);
    CHECK(sarx_memcpy_d2h_begin(nullptr, buf, buf, 1, &i) != SARX_OK);
    CHECK(sarx_memcpy_d2h_end(nullptr, 0) != SARX_OK);
    CHECK(sarx_memcpy2d_d2h(nullptr, buf, 8, buf, 8, 8, 1) != SARX_OK);
    CHECK(sarx_memcpy2d_h2d(nullptr, buf, 8, buf, 8, 8, 1) != SARX_OK);
    CHECK(sarx_memset(nullptr, buf, 0, 1) != SARX_OK);
    CHECK(sarx_sync(nullptr) != SARX_OK);
    CHECK(sarx_select_lane(nullptr, 0) != SARX_OK);
    CHECK(sarx_lanes_join(nullptr) != SARX_OK);
    CHECK(sarx_set_range_cus(nullptr, 0) != SARX_OK);
    CHECK(sarx_probe_lanes(nullptr, 0, 1, 100, d) != SARX_OK);
    CHECK(sarx_event_record(nullptr, 0) != SARX_OK);
    CHECK(sarx_event_elapsed_ms(nullptr, 0, 1, &f) != SARX_OK);
    CHECK(sarx_csa_plan_create(nullptr, 64, 64, &prm, 0, &plan) != SARX_OK && plan == nullptr);
    CHECK(sarx_csa_plan_destroy(nullptr) == SARX_OK);
    CHECK(sarx_csa_plan_bytes(nullptr, &u) != SARX_OK);
    CHECK(sarx_csa_axes(nullptr, d, d) != SARX_OK);
    CHECK(sarx_csa_focus_host(nullptr, buf, buf) != SARX_OK);
    CHECK(sarx_csa_focus_host_c128(nullptr, buf, buf) != SARX_OK);
    CHECK(sarx_csa_focus_host_begin(nullptr, buf, buf, &i) != SARX_OK);
    CHECK(sarx_csa_focus_host_end(nullptr, 0) != SARX_OK);
    CHECK(sarx_csa_focus_dev(nullptr, buf, buf) != SARX_OK);
    CHECK(sarx_csa_pass(nullptr, 1, buf, buf) != SARX_OK);
    CHECK(sarx_csa_plan_mark_range(nullptr, 0, 1) != SARX_OK);
    CHECK(sarx_csa_plan_stamp_range(nullptr, nullptr) != SARX_OK);
    CHECK(sarx_csa_plan_set_look_slot(nullptr, 16, &f) != SARX_OK);
    CHECK(sarx_csa_plan_set_max_slot(nullptr, &f) != SARX_OK);
    CHECK(sarx_csa_plan_set_ati(nullptr, buf, &f, 0.05f, 0.0, &f, &f, &f, 0) != SARX_OK);
    CHECK(sarx_rda_plan_create(nullptr, 64, 64, &prm, &rplan) != SARX_OK && rplan == nullptr);
    CHECK(sarx_rda_plan_destroy(nullptr) == SARX_OK);
    CHECK(sarx_rda_focus_host(nullptr, buf, &f, nullptr, nullptr, nullptr) != SARX_OK);
    CHECK(sarx_rda_focus_dev(nullptr, buf, &f, nullptr, nullptr, nullptr) != SARX_OK);
    CHECK(sarx_rda_focus_host2(nullptr, buf, &f, nullptr, nullptr, nullptr, nullptr) != SARX_OK);
    CHECK(sarx_rda_focus_dev2(nullptr, buf, &f, nullptr, nullptr, nullptr, nullptr) != SARX_OK);
    CHECK(sarx_rda_axes(nullptr, d, d, d) != SARX_OK);
    CHECK(sarx_ati_dpca_dev(nullptr, buf, buf, 4, 0.0, &outs, d, d) != SARX_OK);
    CHECK(sarx_ati_dpca_masked_dev(nullptr, buf, buf, 4, 0.0, &f, 0.05f, &outs) != SARX_OK);
    CHECK(sarx_ati_stats(nullptr, d, d) != SARX_OK);
    CHECK(sarx_mask_phase_dev(nullptr, &f, &f, 1, 0.f, &f) != SARX_OK);
    CHECK(sarx_mask_phase_frac_dev(nullptr, &f, &f, 1, 0.05f, &f) != SARX_OK);
    CHECK(sarx_magnitude_dev(nullptr, buf, &f, 1) != SARX_OK);
    CHECK(sarx_max_abs_f32_dev(nullptr, &f, 1, &f) != SARX_OK);
    CHECK(sarx_corner_turn_dev(nullptr, buf, buf, 2, 2) != SARX_OK);
    CHECK(sarx_multilook_dev(nullptr, buf, &f, 16, 16, 4) != SARX_OK);
    CHECK(sarx_echo_synth_dev(nullptr, d, &f, d, 1, 1, 1, 1.0, 1.0, buf, 0) != SARX_OK);
    CHECK(sarx_echo_geometry_dev(nullptr, 0, 1, 1, d, d, d, d, d, d, 3e8, 1e9, 1.0, 0.03, d, &f) != SARX_OK);
    CHECK(sarx_echo_spotlight_dev(nullptr, d, &f, d, 1, 1, 1, 1.0, 1.0, buf) != SARX_OK);
    CHECK(sarx_tdbp_plan_create(nullptr, 1, 2, 1, 1, &tk, &tplan) != SARX_OK && tplan == nullptr);
    CHECK(sarx_tdbp_plan_destroy(nullptr) == SARX_OK);
    CHECK(sarx_tdbp_focus_dev(nullptr, buf, d, d, d, 0.0, d, 1.0, buf) != SARX_OK);
    CHECK(sarx_tdbp_focus_host(nullptr, buf, d, d, d, 0.0, d, 1.0, buf, nullptr) != SARX_OK);
    CHECK(sarx_tdbp_last_window(nullptr, &i, &i) != SARX_OK);
    CHECK(sarx_fill_noise_c64(nullptr, buf, 1, 1) != SARX_OK);
    CHECK(sarx_add_ocean_noise_dev(nullptr, buf, 1, 1.0, 1.0, 1.0, 1) != SARX_OK);
    CHECK(sarx_add_ocean_noise_rel_dev(nullptr, buf, 1, 1, 10.0, 10.0, 1.0, 1) != SARX_OK);
    CHECK(sarx_power_stats_dev(nullptr, buf, 1, d, d) != SARX_OK);
    CHECK(sarx_comm_unique_id(nullptr) != SARX_OK);
    rc = sarx_rccl_info(buf, sizeof buf, &i, &i);                          // librccl present or not: a code and, on failure, a message
    CHECK(rc == SARX_OK || (rc == SARX_ERR_COMM && strlen(sarx_last_error(nullptr)) > 0));
    CHECK(sarx_comm_init(nullptr, buf, 1, 0) != SARX_OK);
    CHECK(sarx_allgather_dev(nullptr, buf, buf, 4) != SARX_OK);
    CHECK(sarx_allreduce_max_dev(nullptr, &f, 1) != SARX_OK);
    CHECK(sarx_comm_sync(nullptr) != SARX_OK);
    CHECK(sarx_comm_fence_compute(nullptr) != SARX_OK);
    CHECK(sarx_comm_mark(nullptr, 0) != SARX_OK);
    CHECK(sarx_comm_wait_mark(nullptr, 0) != SARX_OK);
    CHECK(sarx_comm_destroy(nullptr) != SARX_OK);
    CHECK(strstr(sarx_last_error(nullptr), "NULL") != nullptr);          // the last of them left its message
    CHECK(sarx_persistent_grid(2, 256, 7199) == 512 && sarx_persistent_grid(0, 0, 0) == 1 && sarx_persistent_grid(1, 256, 100) == 100);
}

static void fill(std::vector<unsigned char>& v, unsigned seed) {
    unsigned x = seed * 2654435761u + 12345u;
    for (auto& b : v) { x = x * 1664525u + 1013904223u; b = (unsigned char)(x >> 24); }
}

static void staged_copy_logic() {
    const size_t CH = sarx_test_copy_chunk();
    const int T = sarx_test_copy_threads();
    CHECK(CH >= (1u << 20) && T >= 2 && T <= 32);
    // sizes around every boundary of the chunking: below the staging threshold (4 chunks), exactly on it, one byte past a chunk, fewer
    // chunks than threads, exactly one round of chunks, a ragged second round
    const size_t sizes[] = {1, 4096, 4 * CH - 8, 4 * CH, 4 * CH + 8, 5 * CH + 24, (size_t)T * CH, (size_t)T * CH + CH / 2 + 16, (size_t)(2 * T + 1) * CH + 8};
    for (size_t bytes : sizes) {
        std::vector<unsigned char> src(bytes), dst(bytes + 64, 0xAB);
        fill(src, (unsigned)bytes);
        for (int to_device = 0; to_device < 2; ++to_device)
            for (int ordered = 0; ordered < 2; ++ordered) {
                memset(dst.data(), 0xAB, dst.size());
                int inl = -1;
                const int us = to_device ? (1 + (int)(bytes % 3)) : T;
                CHECK(sarx_test_staged_copy(dst.data(), src.data(), bytes, to_device, 0, ordered, 0u, -1, 0, us == 3 ? T : us, &inl) == 0);
                CHECK(memcmp(dst.data(), src.data(), bytes) == 0);
                for (size_t k = bytes; k < dst.size(); ++k) CHECK(dst[k] == 0xAB);              // nothing past the end
            }
    }
    // threads that cannot be started: their shares run inline on the caller, the result is the same
    {
        const size_t bytes = (size_t)(T + 3) * CH + 40;
        std::vector<unsigned char> src(bytes), dst(bytes);
        fill(src, 7);
        const unsigned masks[] = {1u, 1u << (T - 1), 0x5u, (1u << T) - 1u};
        for (unsigned m : masks)
            for (int to_device = 0; to_device < 2; ++to_device) {
                memset(dst.data(), 0, bytes);
                int inl = 0;
                CHECK(sarx_test_staged_copy(dst.data(), src.data(), bytes, to_device, 0, 1, m, -1, 0, 1, &inl) == 0);
                CHECK(inl == __builtin_popcount(m));
                CHECK(memcmp(dst.data(), src.data(), bytes) == 0);
            }
        // a copy that fails somewhere in the middle: the error comes back (not success, not a crash), every thread has been joined
        for (int fail_at : {0, 1, T - 1, T, T + 2})
            for (int to_device = 0; to_device < 2; ++to_device)
                CHECK(sarx_test_staged_copy(dst.data(), src.data(), bytes, to_device, 0, 1, (unsigned)fail_at & 3u, fail_at, 0, 2, nullptr) != 0);
        // page-locked host memory: one DMA, no staging
        memset(dst.data(), 0, bytes);
        CHECK(sarx_test_staged_copy(dst.data(), src.data(), bytes, 1, 0, 0, 0u, -1, 1, 1, nullptr) == 0);
        CHECK(memcmp(dst.data(), src.data(), bytes) == 0);
        CHECK(sarx_test_staged_copy(dst.data(), src.data(), bytes, 1, 0, 0, 0u, 0, 1, 1, nullptr) != 0);
    }
    // complex128 -> complex64 narrowing (upload only): bytes counts the complex64 side; small (one chunk, no thread) and staged
    for (size_t n_floats : {(size_t)10, CH / 4, CH / 4 + 2, (size_t)(T + 1) * (CH / 4) + 6}) {
        std::vector<double> src(n_floats);
        std::vector<float> dst(n_floats + 4, -1.0f);
        for (size_t k = 0; k < n_floats; ++k) src[k] = 0.1 * (double)(k % 1000) - 37.0 + 1e-9 * (double)k;
        CHECK(sarx_test_staged_copy(dst.data(), src.data(), n_floats * sizeof(float), 1, 1, 1, n_floats % 2 ? 2u : 0u, -1, 0, 1, nullptr) == 0);
        size_t bad = 0;
        for (size_t k = 0; k < n_floats; ++k) bad += dst[k] != (float)src[k];
        CHECK(bad == 0);
        for (size_t k = n_floats; k < dst.size(); ++k) CHECK(dst[k] == -1.0f);
    }
}

int main() {
    argument_checking();
    staged_copy_logic();
    CHECK(sarx_test_guard(0) == SARX_OK);
    CHECK(sarx_test_guard(1) == SARX_ERR_NOMEM && strstr(sarx_last_error(nullptr), "memory") != nullptr);
    CHECK(sarx_test_guard(2) == SARX_ERR_DEVICE);
    if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    printf("abi_asan_test: all checks passed\n");
    return 0;
}
