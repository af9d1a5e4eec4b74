// Point-target raw-echo synthesis: the sample loop of
// run_physics_engine (sar_satellite_sim.py:264-302) and run_bistatic_physics_gpu
// (sar_ati_dcpa_sim_csa.py:137-178).
//
//   raw[i][j] = sum_b amp_b * [|u| <= Tp/2] * exp(j(phase_base[i][b] + pi*k*u^2)),
//   u = (t_fast[j] - tau[i][b]) - Tp/2
//
// The host (NumPy, as in the reference) supplies per pulse and target the delay tau and
// the carrier phase in revolutions (-2 FC d/C resp. -FC tau: 3e7 revolutions, fp64), and
// the fast-time grid exactly as the reference builds it (linspace, :254 / :113).  The
// kernel does the n_pulses x n_targets x n_samples work: one thread per sample, targets
// staged through LDS, phase summed and reduced in fp64, sine/cosine and accumulation fp32.
// run_physics_spotlight (sar_batch_sim.py:145-149) is the same loop with u = t_fast - tau (no Tp/2 offset)
// and an amplitude per pulse and target (rcs times the antenna pattern): amp_pt, u_off = 0.
// Compute-bound (28 issue slots per target-sample, no reuse of HBM data), so no roofline
// in bytes: 5000 targets x 7200 x 13200 = 4.8e11 target-samples.
#include "csa_kernels.h"
#include "fft_core.hpp"

namespace sarx {

static constexpr int ECHO_THREADS = 256;

__global__ __launch_bounds__(ECHO_THREADS) void echo_synth_kernel(EchoArgs a) {
    __shared__ double s_tau[ECHO_THREADS], s_pb[ECHO_THREADS];
    __shared__ float s_amp[ECHO_THREADS];
    const int j = blockIdx.x * ECHO_THREADS + threadIdx.x;
    const int i = blockIdx.y;
    const bool live = j < a.n_samples;
    const double tf = live ? a.t_fast[j] : 0.0;
    const double half_tp = 0.5 * a.t_p, hk = 0.5 * a.kr;
    float acc_re = 0.f, acc_im = 0.f;
    for (int b0 = 0; b0 < a.n_targets; b0 += ECHO_THREADS) {
        const int b = b0 + threadIdx.x;
        __syncthreads();
        if (b < a.n_targets) {
            const double2 tp = a.tau_pb[(size_t)i * a.n_targets + b];
            s_tau[threadIdx.x] = tp.x;
            s_pb[threadIdx.x] = tp.y;
            s_amp[threadIdx.x] = a.amp_pt ? a.amp_pt[(size_t)i * a.n_targets + b] : a.amp[b];
        }
        __syncthreads();
        const int nb = min(ECHO_THREADS, a.n_targets - b0);
        for (int k = 0; k < nb; ++k) {
            const double u = (tf - s_tau[k]) - a.u_off;          // (:290,293 / :164,166)
            const float gate = (fabs(u) <= half_tp) ? s_amp[k] : 0.f;
            const cf e = cis_rev(fma(hk * u, u, s_pb[k]));      // phase_base + pi k u^2, in revolutions
            acc_re = fmaf(gate, e.x, acc_re);
            acc_im = fmaf(gate, e.y, acc_im);
        }
    }
    if (live) a.out[(size_t)i * a.n_samples + j] = make_float2(acc_re, acc_im);
}

hipError_t launch_echo_synth(const EchoArgs& a, hipStream_t st) {
    dim3 grid((a.n_samples + ECHO_THREADS - 1) / ECHO_THREADS, a.n_pulses);
    hipLaunchKernelGGL(echo_synth_kernel, grid, dim3(ECHO_THREADS), 0, st, a);
    return hipGetLastError();
}

}  // namespace sarx
