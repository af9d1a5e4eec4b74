/* sarx.h - C ABI of libsarx.so, the MI355X (gfx950) backend for the
 * raw-echo -> Chirp-Scaling focus -> two-channel ATI/DPCA hot path of
 * noiseinspacechannel/NIS-SAR-AMTIGMTI-Video.
 *
 * The reference has no FFI layer: its boundary is plain Python functions
 * (SURVEY.md 8b).  Each entry point below names the reference code it
 * replaces (file:line under the reference tree); INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C, no C++/torch types; complex64 is two packed floats (re, im);
 *  - every call returns 0 on success or a negative sarx_status; the message is
 *    available from sarx_last_error(); nothing throws across the boundary;
 *  - the caller owns every host buffer; device buffers come from sarx_malloc
 *    (or are any valid HIP device pointer of this process) and are owned by
 *    whoever allocated them; a plan owns its tables and scratch;
 *  - one sarx_ctx per GPU; a ctx is not thread-safe; work is issued on the
 *    ctx's own HIP stream; *_host calls and sarx_sync block, *_dev calls
 *    only enqueue;
 *  - there is no CPU fallback: without a usable gfx950 device sarx_init fails.
 */
#ifndef SARX_H
#define SARX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SARX_VERSION 206   /* 206: sarx_rda_focus_host2 / _dev2 (the airborne script's eighth output, the azimuth-compressed map), sarx_memcpy_h2d_lane; 205: sarx_csa_plan_stamp_range (execution span of the fused range launch from in-kernel clock stamps), overlapped host transfers (sarx_memcpy_h2d_unordered, sarx_memcpy_d2h_begin / _end, sarx_csa_focus_host_begin / _end); 204: sarx_max_abs_f32_dev, sarx_allreduce_max_dev (global normalisation of a frame stack), sarx_host_alloc / _free, lanes (sarx_select_lane, sarx_lanes_join), sarx_add_ocean_noise_rel_dev, sarx_probe_lanes; 203: permuted-spectrum pass ids 12 / 13; a focus with sarx_csa_plan_set_ati armed no longer clears the max slot */

typedef struct sarx_ctx sarx_ctx;
typedef struct sarx_plan sarx_plan;
typedef struct sarx_rda_plan sarx_rda_plan;
typedef struct sarx_tdbp_plan sarx_tdbp_plan;

typedef enum {
    SARX_OK = 0,
    SARX_ERR_INVALID = -1,      /* bad argument (size, null pointer, flag) */
    SARX_ERR_UNSUPPORTED = -2,  /* size not a supported power of two, etc. */
    SARX_ERR_DEVICE = -3,       /* HIP runtime error, text in sarx_last_error */
    SARX_ERR_NOMEM = -4,
    SARX_ERR_COMM = -5          /* RCCL error / not initialised */
} sarx_status;

/* Positional arguments of sar_focus_csa after `phist`
 * (sar_ati_dcpa_sim_csa.py:202).  pulse_width_s is carried and unused, as in
 * the reference. */
typedef struct {
    double wavelength_m;       /* center_wavelength_m  */
    double pulse_width_s;      /* pulse_width_sec (unused by the algorithm) */
    double chirp_rate_hz_s;    /* chirp_rate_hzpsec  Kr */
    double sample_rate_hz;     /* sample_rate_hz     fs */
    double prf_hz;             /* prf_hz */
    double platform_speed_mps; /* platform_speed_mps Vr */
    double range_ref_m;        /* range_ref_m        R_ref */
    double t_start_fast_s;     /* t_start_fast */
} sarx_radar_params;

/* plan flags */
#define SARX_OUT_AZ_MAJOR 0u      /* image left as [n_az x n_rg]; img.T is a view of it (what NumPy returns) */
#define SARX_OUT_RG_MAJOR 1u      /* image corner-turned to row-major [n_rg x n_az] */
#define SARX_FUSE_RANGE 2u        /* passes 2+3 (range FFT, Phi2, range IFFT, Phi3) in one launch */

/* pass identifiers for sarx_csa_pass (per-pass parity tests and profiling) */
#define SARX_PASS_AZ_FFT_PHI1 1   /* sar_ati_dcpa_sim_csa.py:233-274 */
#define SARX_PASS_RG_FFT_PHI2 2   /* :278-326  (the roofline pass) */
#define SARX_PASS_RG_IFFT_PHI3 3  /* :331-382 */
#define SARX_PASS_AZ_IFFT 4       /* :385 */
#define SARX_PASS_RG_FUSED_23 23  /* passes 2 and 3 in one launch */
/* Passes 2 and 3 with the range spectrum in the PERMUTED order the unfused focus keeps it in between its two range
 * launches (n_rg = 16384 only; SARX_ERR_UNSUPPORTED otherwise):  with k the natural (numpy.fft.fftfreq) bin index, q = k mod 16 and
 * k2 = k div 16,  P[(k2 div 64) * 1024 + q * 64 + (k2 mod 64)] = X[k].  The spectrum exists only between these two launches (:278-382 never hands it out), so its
 * storage order is free, and this one lets each launch run with a single workgroup-wide exchange (csrc/range_wp.hip). */
#define SARX_PASS_RG_FFT_PHI2_PERM 12   /* natural-order line in, permuted spectrum out */
#define SARX_PASS_RG_IFFT_PHI3_PERM 13  /* permuted spectrum in, natural-order line out */

/* ---- context ------------------------------------------------------------- */
int sarx_init(int device_id, sarx_ctx** out_ctx);
int sarx_destroy(sarx_ctx* ctx);
/* last error text of this ctx (or of the failed sarx_init when ctx is NULL) */
const char* sarx_last_error(const sarx_ctx* ctx);
int sarx_version(void);
int sarx_device_count(int* out_count);
int sarx_device_info(sarx_ctx* ctx, char* name, size_t name_len, int* compute_units,
                     uint64_t* hbm_bytes, char* arch, size_t arch_len);
/* grid size the persistent range kernels use: min(wgs_per_cu * cus, work_items), at least 1; cus is the compute-unit
 * count of the ctx's own device (pure host arithmetic, no device needed) */
int sarx_persistent_grid(int wgs_per_cu, int cus, int work_items);

/* ---- device memory and timing (so the Python host needs no torch) -------- */
int sarx_malloc(sarx_ctx* ctx, size_t bytes, void** out_dptr);
int sarx_free(sarx_ctx* ctx, void* dptr);
/* Page-locked host memory for the results (or inputs) of the *_host entry points and sarx_memcpy_*: such a buffer is copied
 * with one DMA at the PCIe rate - no staging through pinned chunks, no first touch of fresh pages (a new 2 GiB NumPy result
 * array costs ~90 ms of page faults at 16384^2).  Allocation itself is slow (~0.25 s per GiB): allocate once, reuse.  The
 * Python facade keeps a small pool of these for the arrays sar_focus_csa returns (the reference allocates its result per
 * call, sar_ati_dcpa_sim_csa.py:385-396; callee-allocates stays the contract). */
int sarx_host_alloc(sarx_ctx* ctx, size_t bytes, void** out_hptr);
int sarx_host_free(sarx_ctx* ctx, void* hptr);
int sarx_memcpy_h2d(sarx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int sarx_memcpy_d2h(sarx_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* Overlapped host transfers: PCIe is full duplex and both copies can run beside a focus, so a frame loop that calls
 * sar_focus_csa on host arrays (sar_batch_sim.py:303-331; the two back-to-back channel calls of sar_ati_dcpa_sim_csa.py:410-411) need
 * not pay upload + focus + download one after the other.
 * sarx_memcpy_h2d_unordered: like sarx_memcpy_h2d (blocking for the caller, staged through pinned chunks by the copy threads), but
 *   it does NOT wait for work already enqueued on the lanes: the caller guarantees that nothing enqueued reads or writes dst_dev.
 * sarx_memcpy_d2h_begin: asynchronous download into PAGE-LOCKED host memory (sarx_host_alloc; SARX_ERR_INVALID for pageable memory)
 *   on the ctx's download stream, ordered after everything enqueued so far on the current lane; *out_slot identifies it (eight may be
 *   in flight).  sarx_memcpy_d2h_end(slot) blocks until that copy has landed.  sarx_sync waits for all of them too. */
int sarx_memcpy_h2d_unordered(sarx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
/* sarx_memcpy_h2d_lane: sarx_memcpy_h2d that waits for the CURRENT lane's enqueued work only (sarx_memcpy_h2d waits for every lane):
 * for a table only this lane's launches read - the per-frame tables of a frame loop with frames in flight on other lanes
 * (sar_batch_sim.py:303-331), which a wait for every lane would drain at every upload. */
int sarx_memcpy_h2d_lane(sarx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int sarx_memcpy_d2h_begin(sarx_ctx* ctx, void* dst_host_pinned, const void* src_dev, size_t bytes, int* out_slot);
int sarx_memcpy_d2h_end(sarx_ctx* ctx, int slot);
int sarx_memcpy_d2d(sarx_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
/* strided copies (rows of `width` bytes, pitches in bytes), blocking: sample columns / blocks of a device image */
int sarx_memcpy2d_d2h(sarx_ctx* ctx, void* dst_host, size_t dst_pitch, const void* src_dev, size_t src_pitch,
                      size_t width, size_t height);
int sarx_memcpy2d_h2d(sarx_ctx* ctx, void* dst_dev, size_t dst_pitch, const void* src_host, size_t src_pitch,
                      size_t width, size_t height);
int sarx_memset(sarx_ctx* ctx, void* dst_dev, int value, size_t bytes);
int sarx_sync(sarx_ctx* ctx);      /* waits for every lane and the comm stream */
/* Lanes: a ctx owns up to four compute streams; lane 0 exists from sarx_init.  sarx_select_lane makes every LATER enqueue of
 * this ctx (focus, passes, products, memset, events) go to that lane's stream.  The frames of a batch are independent
 * (sar_batch_sim.py:303-331), so consecutive frames may be enqueued on alternating lanes - each lane with its OWN plan and its
 * own buffers - and kernels of neighbouring frames then share the GPU: the issue-bound range launch of one frame beside the
 * bandwidth-bound azimuth launches of the next (16384^2: 4.25-4.37 -> 3.91-4.26 ms per frame with two frames in flight, box by box).  Results are
 * bit-identical to one lane.  sarx_lanes_join: on the device, every lane waits for everything enqueued so far on every lane.
 * Blocking host copies and sarx_sync wait for all lanes. */
int sarx_select_lane(sarx_ctx* ctx, int lane);
int sarx_lanes_join(sarx_ctx* ctx);
/* With frames in flight the persistent fused range launch (16384-sample lines: one 136 KiB-LDS workgroup per CU, which no azimuth
 * tile can share a CU with) should leave part of the chip to the other lane's azimuth launches: its grid is sized for `cus`
 * compute units instead of all of them (0 = all, the default; 192 of 256 measured best with two lanes: 3.91-4.26 ms per 16384^2
 * frame over thirteen boxes, against 3.93-4.34 ms when both lanes ask for the whole chip: profiles/r04_boxes_inflight2_vs_1.log).  Results do not depend on it. */
int sarx_set_range_cus(sarx_ctx* ctx, int cus);
/* Do lanes a and b run side by side?  HIP maps streams onto a few hardware queues; two lanes that share one take turns and frames in
 * flight on them gain nothing (4.6 against 4.0 ms per 16384^2 frame, profiles/r04_ai_lane_probe_boxes.log).  Two small launches (64
 * one-wave workgroups spinning `us` microseconds) are timed on lane a alone and on a and b together: *ratio = together / alone, 1.0 =
 * side by side, 2.0 = taking turns.  Blocking, a few milliseconds; creates the lanes if needed. */
int sarx_probe_lanes(sarx_ctx* ctx, int lane_a, int lane_b, int us, double* ratio);
/* HIP events on the ctx stream: record `slot` (0..255); elapsed ms between two recorded slots */
int sarx_event_record(sarx_ctx* ctx, int slot);
int sarx_event_elapsed_ms(sarx_ctx* ctx, int slot_start, int slot_stop, float* out_ms);

/* ---- CSA focus: replaces sar_focus_csa (sar_ati_dcpa_sim_csa.py:202-396) -- */
/* n_az, n_rg in [2, 32768].  Powers of two >= 16 run the tuned kernels; the reference's native 7199 x 13200 (:47,111,402)
 * runs direct transforms (mixed-radix 24 x 22 x 25 range lines, prime-factor 23 x 313 azimuth transforms); any other size runs
 * chirp-z transforms over the power-of-two kernels (a non-power-of-two n_az must be <= 16384; range lines beyond 16384 samples
 * run as 128 x 256 / 256 x 256 split transforms).  Unsupported sizes fail with SARX_ERR_UNSUPPORTED, never silently.
 * Environment switches read at plan creation, for A/B measurements only: SARX_RANGE_MIXED=0, SARX_AZ_PFA=0 (chirp-z routes at
 * the native size), SARX_SLAB_MIB=<MiB> (tile-grouped middle section of the power-of-two focus; measured slower, off by default). */
int sarx_csa_plan_create(sarx_ctx* ctx, int n_az, int n_rg, const sarx_radar_params* params,
                         unsigned flags, sarx_plan** out_plan);
int sarx_csa_plan_destroy(sarx_plan* plan);
/* range_axis[n_rg] = c*tau/2 (:346,388); cross_range_axis[n_az] = (i/prf - mean)*Vr (:392-394) */
int sarx_csa_axes(const sarx_plan* plan, double* range_axis, double* cross_range_axis);
/* host in / host out, blocking: upload, four passes, download.
 * phist: [n_az x n_rg] complex64 row-major.  image: n_az*n_rg complex64 in the plan's layout. */
int sarx_csa_focus_host(sarx_plan* plan, const void* phist_host, void* image_host);
/* the same with phist as complex128 (the dtype the reference's arrays have, sar_ati_dcpa_sim_csa.py:135): rounded to complex64
 * while it is staged for the transfer, by the copy threads; the image comes back as complex64 */
int sarx_csa_focus_host_c128(sarx_plan* plan, const void* phist_c128_host, void* image_host);
/* sarx_csa_focus_host as a two-deep pipeline.  _begin uploads the frame (blocking for the caller, but without waiting for the frame
 * that is still focusing or downloading), enqueues its focus and - when image_host is page-locked (sarx_host_alloc) - its download on
 * the download stream, and returns a ticket; _end(ticket) blocks until that frame's image is complete in image_host (a pageable
 * image_host is downloaded there, staged and blocking: no overlap for that half).  Called as begin(i+1), end(i), begin(i+2), ... frame
 * i+1 uploads while frame i focuses and downloads: 2 GiB each way at 16384^2 in ~41-46 ms per frame instead of 82-88.  Both host
 * buffers must stay valid and untouched until _end; at most two frames per plan are in flight (a third _begin returns
 * SARX_ERR_INVALID); results are bit-identical to sarx_csa_focus_host.  The plan owns two more image pairs on the device for it. */
int sarx_csa_focus_host_begin(sarx_plan* plan, const void* phist_host, void* image_host, int* out_ticket);
int sarx_csa_focus_host_end(sarx_plan* plan, int ticket);
/* device in / device out, asynchronous on the ctx stream.  d_phist is not modified.
 * d_image must not alias d_phist. */
int sarx_csa_focus_dev(sarx_plan* plan, const void* d_phist, void* d_image);
/* one pass, device to device (d_out may equal d_in only for the range passes).  Power-of-two plans: every pass;
 * 7199 x 13200 plans: every pass; other any-size plans with n_rg = 13200: the range passes; otherwise SARX_ERR_UNSUPPORTED */
int sarx_csa_pass(sarx_plan* plan, int pass_id, const void* d_in, void* d_out);
/* profiling hook: sarx_csa_focus_dev records ctx event slots around its range pass(es)
 * (the roofline kernel); pass -1, -1 to switch off */
int sarx_csa_plan_mark_range(sarx_plan* plan, int slot_start, int slot_stop);
/* profiling hook for launches that share the GPU (frames in flight on several lanes): while d_pair is set, the fused range launch of
 * every sarx_csa_focus_dev of this plan (16384-sample plans: range_fused_wl_kernel; ignored otherwise) reduces {start of its first
 * workgroup, end of its last workgroup} into d_pair[0] (atomic min) / d_pair[1] (atomic max), in ticks of the device's 100 MHz
 * constant clock (s_memrealtime) - the launch's execution span, which an event pair on the stream cannot separate from the time the
 * launch queues behind another lane's kernels.  The caller initialises d_pair to {UINT64_MAX, 0} and reads it after a sync; one
 * pair per launch to be measured.  d_pair = NULL switches it off (the kernel then executes no stamp instruction). */
int sarx_csa_plan_stamp_range(sarx_plan* plan, uint64_t* d_pair);
/* VideoSAR stack slot fused into the focus: while d_slot is set, every sarx_csa_focus_dev of this plan also writes
 * d_slot[(n_az/looks) x (n_rg/looks)] fp32 = mean of |image|^2 over looks x looks blocks (what sarx_multilook_dev computes from the
 * finished image, sar_batch_sim.py:322's display stack) - the last azimuth launch emits row-wise partial sums and a small launch
 * finishes them, so the 2 GiB image is not read again.  looks: a power of two <= 32 dividing both extents; power-of-two plans
 * only; the slot is [n_az/looks x n_rg/looks] whatever the image layout.  d_slot = NULL switches it off.  Bitwise reproducible. */
int sarx_csa_plan_set_look_slot(sarx_plan* plan, int looks, float* d_slot);
#define SARX_MAX_SLOT_BYTES 32768  /* 256 partial maxima, one per 128-byte line: atomics on one address would serialise */
/* max |image| fused into the focus: while d_max is set, every sarx_csa_focus_dev of this plan also leaves max |image| (fp32, the
 * hypotf the ATI launch computes) in d_max[SARX_MAX_SLOT_BYTES] as 256 partial maxima (float k*32, the rest zero; the maximum is
 * their maximum) - the last azimuth launch reduces them while it writes the image - so that
 * sarx_ati_dpca_masked_dev can apply the 5 % mask (sar_ati_dcpa_sim_csa.py:447-449) in the ATI pass itself instead of a further
 * pass over two planes.  Power-of-two plans and the native 7199 x 13200; d_max = NULL switches it off.
 * A focus that has sarx_csa_plan_set_ati armed neither clears nor reduces into the slot: it is the second channel's focus
 * and READS the first channel's maximum from it, so one plan may keep both settings across the two focuses of a frame. */
int sarx_csa_plan_set_max_slot(sarx_plan* plan, float* d_max);
/* ATI / DPCA products fused into the focus of the SECOND channel: while d_slc1 is set, the last azimuth launch of every
 * sarx_csa_focus_dev of this plan reads slc1 = d_slc1 [n_az x n_rg] beside the samples of slc2 it is about to write and emits
 * ati_phase (masked: 0 where |slc1| <= mask_frac * max|slc1|, :447-449), |slc1| and |slc1 - slc2 e^(i cal)| (:414-419) as
 * [n_az x n_rg] fp32 planes - bit for bit what sarx_ati_dpca_masked_dev computes from the two finished images, without writing
 * slc2 and reading both images again (keep_image != 0 also writes slc2 to d_image as usual; d_image is needed as scratch either
 * way).  d_max: the SARX_MAX_SLOT_BYTES slot of the plan that focused slc1.  sarx_ati_stats afterwards returns max|slc1| and the
 * phase-balance sum (fixed-order reduction: reproducible; not bit-identical to the separate launch's order of additions).
 * Power-of-two plans (n_rg a multiple of 64, or the last azimuth launch's tiles whole waves) and the native 7199 x 13200, default
 * image layout; takes precedence over a look slot (while armed no look slot is written) and over the max slot (left as it is).
 * d_slc1 = NULL switches it off. */
int sarx_csa_plan_set_ati(sarx_plan* plan, const void* d_slc1, const float* d_max, float mask_frac, double cal_phase,
                          float* d_ati_phase_masked, float* d_slc1_mag, float* d_dpca_mag, int keep_image);
/* bytes of HBM scratch the plan holds (two ping-pong images + tables) */
int sarx_csa_plan_bytes(const sarx_plan* plan, uint64_t* out_bytes);

/* ---- Range-Doppler focus: replaces sar_focus_rda (sar_satellite_sim.py:356-448, and its copies at
 *      sar_satellite_moving_sim.py:208, sar_vehicle_sim.py:182) --------------------------------------
 * params carries the function's positional arguments (range_ref_m = range_grp_m, t_start_fast unused).
 * Images here are [n_pulses x n_ranges] row-major: the memory of the reference's [n_ranges x n_pulses]
 * argument when it is raw.T, and of the returned sar_image_mag.T.  n_ranges + matched-filter taps - 1
 * must be <= 65536; n_pulses <= 32768 (a non-power-of-two n_pulses <= 16384). */
int sarx_rda_plan_create(sarx_ctx* ctx, int n_ranges, int n_pulses, const sarx_radar_params* params,
                         sarx_rda_plan** out_plan);
int sarx_rda_plan_destroy(sarx_rda_plan* plan);
/* host in / host out, blocking.  Optional complex64 outputs (NULL = skip): range-compressed data (:392),
 * range-Doppler map (:399), map after RCMC (:427), each [n_pulses x n_ranges]. */
int sarx_rda_focus_host(sarx_rda_plan* plan, const void* phist_pulse_major_host, float* image_mag_host,
                        void* range_compressed_host, void* range_doppler_host, void* range_doppler_rcmc_host);
/* device in / device out, asynchronous on the ctx stream: d_image_mag [n_pulses x n_ranges] fp32 is written by the last
 * launch itself; the optional complex64 maps (NULL = skip) are device-to-device copies of the plan's buffers.  Nothing
 * is downloaded and nothing blocks. */
int sarx_rda_focus_dev(sarx_rda_plan* plan, const void* d_phist_pulse_major, float* d_image_mag,
                       void* d_range_compressed, void* d_range_doppler, void* d_range_doppler_rcmc);
/* The same two calls with the map sar_vehicle_sim.py:182-273 returns in addition (its eighth output, :268
 * range_doppler_filtered = the range-Doppler map after RCMC and azimuth compression, Doppler order, [n_pulses x n_ranges]
 * complex64; NULL = skip, and then these ARE the calls above).  sar_satellite_moving_sim.py:208-285 returns only the first
 * three outputs: every optional pointer NULL. */
int sarx_rda_focus_host2(sarx_rda_plan* plan, const void* phist_pulse_major_host, float* image_mag_host,
                         void* range_compressed_host, void* range_doppler_host, void* range_doppler_rcmc_host,
                         void* range_doppler_filtered_host);
int sarx_rda_focus_dev2(sarx_rda_plan* plan, const void* d_phist_pulse_major, float* d_image_mag,
                        void* d_range_compressed, void* d_range_doppler, void* d_range_doppler_rcmc,
                        void* d_range_doppler_filtered);
/* range_axis_centered[n_ranges] (:443-444), cross_range_m[n_pulses] (:442), doppler_freq[n_pulses] (:402-405) */
int sarx_rda_axes(const sarx_rda_plan* plan, double* range_axis_centered, double* cross_range_m, double* doppler_freq);

/* ---- ATI / DPCA: replaces the inline expressions of
 *      sar_ati_dcpa_sim_csa.py:414-419,447-449 and
 *      sar_ati_dcpa_viewer_csa.py:42-52,245-253 ------------------------------ */
typedef struct {
    /* required outputs, n floats each */
    float* ati_phase;   /* angle(slc1 * conj(slc2_cal))   :414-415 */
    float* slc1_mag;    /* |slc1|                          :416 */
    float* dpca_mag;    /* |slc1 - slc2_cal|               :418-419 */
    /* optional outputs (NULL = skip) */
    void* ati_interf;   /* complex64 slc1*conj(slc2_cal)   :414 */
    void* dpca_diff;    /* complex64 slc1 - slc2_cal       :418 */
    float* slc2_mag;    /* viewer 'Ch2 Magnitude' */
    float* slc1_phase;  /* viewer 'Ch1 Phase' */
    float* slc2_phase;  /* viewer 'Ch2 Phase' */
    float* dpca_phase;  /* viewer 'DPCA Phase' */
} sarx_ati_outputs;      /* all device pointers */

/* slc2_cal = slc2 * exp(i*cal_phase) (viewer :43).  Also reduces
 * max|slc1| and sum(slc1*conj(slc2)) (uncalibrated: the phase-balance input,
 * viewer :249).  Results land in host doubles after an internal sync. */
int sarx_ati_dpca_dev(sarx_ctx* ctx, const void* d_slc1, const void* d_slc2, size_t n, double cal_phase,
                      const sarx_ati_outputs* d_out, double* max_mag, double* sum_interf_re_im /*[2]*/);
/* All ATI/DPCA buffers must be 16-byte aligned.  With max_mag and sum_interf_re_im both NULL the call only enqueues;
 * sarx_ati_stats fetches the two reductions of the most recent launch on this ctx later (blocking). */
int sarx_ati_stats(sarx_ctx* ctx, double* max_mag, double* sum_interf_re_im /*[2]*/);
/* The same launch with the magnitude mask applied on the way out: d_out->ati_phase receives ati_phase where
 * |slc1| > mask_frac * max|slc1| and 0 elsewhere (:447-449); d_max = the SARX_MAX_SLOT_BYTES slot sarx_csa_plan_set_max_slot of
 * the plan that focused slc1 filled (or 256 floats, 32 apart, the caller put there).  Only enqueues; sarx_ati_stats works as after sarx_ati_dpca_dev. */
int sarx_ati_dpca_masked_dev(sarx_ctx* ctx, const void* d_slc1, const void* d_slc2, size_t n, double cal_phase, const float* d_max,
                             float mask_frac, const sarx_ati_outputs* d_out);
/* ati_phase[~(mag > mask_frac * max|slc1|)] = 0 with the maximum taken on the device from the most recent
 * sarx_ati_dpca_dev launch of this ctx (sar_ati_dcpa_sim_csa.py:447-449 without a host round trip) */
int sarx_mask_phase_frac_dev(sarx_ctx* ctx, const float* d_phase, const float* d_mag, size_t n, float mask_frac,
                             float* d_out);
/* out[i] = |in[i]|, complex64 in, fp32 out (sar_ati_dcpa_sim_csa.py:416 on its own: full-resolution stack slot) */
int sarx_magnitude_dev(sarx_ctx* ctx, const void* d_in, float* d_out, size_t n);
/* *d_max = max(*d_max, max_i |d_x[i]|) over an fp32 device buffer (clear *d_max with sarx_memset first): a rank's share of
 * g_max = max([np.max(np.abs(fr)) for fr in loaded_frames]) (sar_batch_sim.py:337); asynchronous on the ctx stream */
int sarx_max_abs_f32_dev(sarx_ctx* ctx, const float* d_x, size_t n, float* d_max);
/* ati_phase[~(mag > thr)] = 0  (sar_ati_dcpa_sim_csa.py:447-449); d_out may alias d_phase */
int sarx_mask_phase_dev(sarx_ctx* ctx, const float* d_phase, const float* d_mag, size_t n, float threshold,
                        float* d_out);

/* ---- corner turn (range <-> azimuth transpose through LDS tiles) --------- */
/* out[c][r] = in[r][c], complex64, rows/cols multiples of 32 */
int sarx_corner_turn_dev(sarx_ctx* ctx, const void* d_in, void* d_out, int rows, int cols);

/* ---- VideoSAR stack products --------------------------------------------- */
/* out[R/L x C/L] = mean over LxL blocks of |in|^2 (multilooked intensity), complex64 in, fp32 out */
int sarx_multilook_dev(sarx_ctx* ctx, const void* d_in, float* d_out, int rows, int cols, int looks);

/* ---- synthetic input on device ------------------------------------------- */
/* counter-based N(0,1)+iN(0,1) complex64 noise, reproducible per (seed, index) */
int sarx_fill_noise_c64(sarx_ctx* ctx, void* d_buf, size_t n, uint64_t seed);

/* thermal noise + K-distributed sea clutter added in place to a complex64 buffer: add_ocean_noise
 * (sar_satellite_sim.py:331-344) and generate_noise_tensor (sar_batch_sim.py:66-82):
 * x += noise_std (N + jN) + sqrt(clutter_power * G * E) exp(j 2 pi U), G ~ Gamma(k_nu, 1/k_nu), E ~ Exp(1).
 * Counter-based (seed, index); clutter_power = 0 skips the clutter. */
int sarx_add_ocean_noise_dev(sarx_ctx* ctx, void* d_buf, size_t n, double noise_std, double clutter_power, double k_nu,
                             uint64_t seed);
/* The same with the levels taken on the device, relative to the buffer's own power, so that a frame loop needs no host round trip
 * between the echo and its noise: reference power = max |x|^2 (ref_is_max, sar_batch_sim.py:313-314) or mean |x|^2
 * (sar_satellite_sim.py:333), sigma = sqrt(ref / snr_lin / 2), clutter power = ref / scr_lin (scr_lin = 0: thermal noise only), with
 * snr_lin = 10^(snr_db / 10), scr_lin = 10^(scr_db / 10).  Same partial sums in the same order and the same arithmetic as
 * sarx_power_stats_dev + sarx_add_ocean_noise_dev with the levels computed on the host: bit-identical samples.  Asynchronous. */
int sarx_add_ocean_noise_rel_dev(sarx_ctx* ctx, void* d_buf, size_t n, int ref_is_max, double snr_lin, double scr_lin, double k_nu,
                                 uint64_t seed);
/* max and mean of |x|^2 over a complex64 device buffer (blocking): the reference powers of
 * sar_batch_sim.py:316 (max) and sar_satellite_sim.py:333 (mean).  Either output may be NULL. */
int sarx_power_stats_dev(sarx_ctx* ctx, const void* d_buf, size_t n, double* max_abs2, double* mean_abs2);

/* ---- point-target echo synthesis: the sample loops of run_physics_engine
 *      (sar_satellite_sim.py:264-302) and run_bistatic_physics_gpu
 *      (sar_ati_dcpa_sim_csa.py:137-178) ---------------------------------------
 * raw[i][j] = sum_b amp[b] * [|u|<=Tp/2] * exp(2*pi*i*(pb[i][b] + 0.5*kr*u^2)),  u = t_fast[j]-tau[i][b]-Tp/2
 * d_tau_pb: [n_pulses][n_targets] pairs of doubles {tau seconds, carrier phase in revolutions};
 * d_amp: [n_targets] float sqrt(rcs); d_t_fast: [n_samples] double; d_raw: [n_pulses][n_samples] complex64;
 * accumulate != 0 adds to d_raw (a second target set, e.g. the clutter field of sar_ati_dcpa_sim_csa.py:193-196) */
int sarx_echo_synth_dev(sarx_ctx* ctx, const double* d_tau_pb, const float* d_amp, const double* d_t_fast,
                        int n_pulses, int n_targets, int n_samples, double chirp_rate_hz_s, double pulse_width_s,
                        void* d_raw, int accumulate);

/* Per pulse and target geometry of the three echo models on the device (fp64): fills the d_tau_pb table (and, for
 * model 2, d_amp_pt) that the sample kernels consume.  All pointers are device pointers.
 *   model 0: run_physics_engine (sar_satellite_sim.py:268-272)          needs d_tgt_pos, d_tx_pos; optional d_tgt_vel +
 *            d_t_pulse move the targets: run_moving_physics (sar_satellite_moving_sim.py:137-145)
 *   model 1: run_bistatic_physics_gpu (sar_ati_dcpa_sim_csa.py:151-160)  + d_tgt_vel[3], d_t_pulse, d_aux = receiver positions
 *   model 2: run_physics_spotlight (sar_batch_sim.py:127-150)           + d_aux = platform velocities, d_rcs, l_ant, wavelength */
int sarx_echo_geometry_dev(sarx_ctx* ctx, int model, int n_pulses, int n_targets, const double* d_tgt_pos,
                           const double* d_tgt_vel, const double* d_t_pulse, const double* d_tx_pos, const double* d_aux,
                           const double* d_rcs, double c_light, double fc, double l_ant, double wavelength,
                           double* d_tau_pb, float* d_amp_pt);

/* run_physics_spotlight (sar_batch_sim.py:85-169), sample loop :145-149: as above with u = t_fast[j]-tau[i][b]
 * (no Tp/2 offset) and an amplitude per pulse and target, d_amp_pt [n_pulses][n_targets] float = rcs * antenna gain */
int sarx_echo_spotlight_dev(sarx_ctx* ctx, const double* d_tau_pb, const float* d_amp_pt, const double* d_t_fast,
                            int n_pulses, int n_targets, int n_samples, double chirp_rate_hz_s, double pulse_width_s,
                            void* d_raw);

/* ---- time-domain back-projection: replaces tdbp_gpu (sar_batch_sim.py:171-238) ---------------------
 * The module constants the reference function reads (sar_batch_sim.py:13,20,23-25). */
typedef struct {
    double c;        /* C      propagation speed, m/s */
    double fc;       /* FC     carrier, Hz */
    double fs;       /* FS     sample rate, Hz */
    double t_p;      /* T_P    pulse width, s: the reference chirp has int(T_P*FS) taps (:177) */
    double k_rate;   /* K_RATE chirp rate, Hz/s */
} sarx_tdbp_params;
/* A plan owns the reference-chirp spectrum and all device scratch for [n_pulses x num_samples] complex64 pulses
 * and an [ny x nx] image.  The reference chirp may have at most 16385 taps. */
int sarx_tdbp_plan_create(sarx_ctx* ctx, int n_pulses, int num_samples, int nx, int ny, const sarx_tdbp_params* k,
                          sarx_tdbp_plan** out_plan);
int sarx_tdbp_plan_destroy(sarx_tdbp_plan* plan);
/* raw [n_pulses][num_samples] complex64; pos, vel [n_pulses][3], t_pulses [n_pulses], vel_focus [3]: HOST doubles
 * (tdbp_gpu's pos_plat, vel_plat, t_pulses, vel_focus); image [ny][nx] complex128 as the reference returns it.
 * _dev: raw and image are device pointers, asynchronous on the ctx stream after the small geometry upload.
 * _host: blocking; range_compressed_host (optional, complex64 [n_pulses][num_samples]) receives rc_data (:185). */
int sarx_tdbp_focus_dev(sarx_tdbp_plan* plan, const void* d_raw, const double* pos, const double* vel,
                        const double* t_pulses, double t_start, const double* vel_focus, double scene_size, void* d_image);
int sarx_tdbp_focus_host(sarx_tdbp_plan* plan, const void* raw_host, const double* pos, const double* vel,
                         const double* t_pulses, double t_start, const double* vel_focus, double scene_size,
                         void* image_host, void* range_compressed_host);

/* Only the samples of each pulse that the scene can touch are range-compressed (a bound from the geometry: the
 * back-projection of a 500 m scene reads ~1700 of 22004 samples); asking for range_compressed_host compresses all.
 * [*lo, *hi) of the last focus call, for inspection. */
int sarx_tdbp_last_window(const sarx_tdbp_plan* plan, int* lo, int* hi);

/* ---- multi-GPU: RCCL all-gather of the image stack over xGMI ------------- */
#define SARX_COMM_ID_BYTES 128
int sarx_comm_unique_id(void* id_out /*[SARX_COMM_ID_BYTES]*/);
/* which RCCL the collectives run on: file path, ncclGetVersion of it, NCCL_VERSION_CODE of the headers libsarx was
 * compiled against.  libsarx takes librccl from the directory of the HIP runtime the process is running on (a process
 * that imported torch first runs on torch's bundled runtime and RCCL), SARX_RCCL_PATH overrides. */
int sarx_rccl_info(char* path, size_t path_len, int* version, int* header_version);
int sarx_comm_init(sarx_ctx* ctx, const void* id, int n_ranks, int rank);
/* recv[rank r] = send of rank r; bytes_per_rank multiple of 4; async on the ctx comm stream,
 * ordered after everything already enqueued on the CURRENT lane's compute stream (sarx_select_lane) - and on that lane only:
 * a send buffer written on another lane must be gathered with that lane selected, or after sarx_lanes_join.  The same holds for
 * sarx_allreduce_max_dev, and sarx_comm_fence_compute / sarx_comm_wait_mark make later work of the current lane (only) wait. */
int sarx_allgather_dev(sarx_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank);
/* d_buf[i] = max over ranks of d_buf[i] (fp32, in place); same stream ordering as the gather.  The collective half of the
 * global display normalisation of a frame stack, g_max = max over all frames of max|frame| (sar_batch_sim.py:337-338):
 * every rank reduces its own frames with sarx_max_abs_f32_dev, one float per rank crosses xGMI. */
int sarx_allreduce_max_dev(sarx_ctx* ctx, float* d_buf, size_t count);
int sarx_comm_sync(sarx_ctx* ctx);
/* device-side only: later work on the compute stream waits for every gather enqueued so far
 * (call before overwriting a send buffer that an earlier sarx_allgather_dev may still read) */
int sarx_comm_fence_compute(sarx_ctx* ctx);
/* finer than the fence: sarx_comm_mark records "every gather enqueued so far is finished" in one of four slots on the comm stream;
 * sarx_comm_wait_mark makes later compute-stream work wait (on the device) for that slot's mark (no-op if never recorded).  A
 * double-buffered gather target marks slot (s & 1) after the gather of step s and waits for it before step s + 2 writes the
 * buffer again, so the gather of step s + 1 still overlaps the compute of step s + 2. */
int sarx_comm_mark(sarx_ctx* ctx, int slot);
int sarx_comm_wait_mark(sarx_ctx* ctx, int slot);
int sarx_comm_destroy(sarx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SARX_H */
