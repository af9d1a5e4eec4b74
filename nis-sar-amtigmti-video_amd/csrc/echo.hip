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
// Compute-bound (no reuse of HBM data), so no roofline in bytes: 5000 targets x 7200 x 13200 = 4.8e11 target-samples in
// 174 ms = 58 SIMD cycles per wave of 64 target-samples, i.e. 14 four-cycle issue slots of which the sine/cosine pair
// takes four.  fp64 adds and FMAs issue at the fp32 rate on this part, so the fp64 gate and phase are not the cost:
// a round-2 variant with 16 consecutive samples per thread, the pulse gate tested at the run's ends only and the phase
// stepped as the 32-bit fixed-point accumulator of phase.hpp (no fp64 per sample) passed the same parity tests and took
// 229 ms (113 VGPRs, half the occupancy, divergent edge runs).  Dropped.
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
            s_tau[threadIdx.x] = tp.x + a.u_off;       // u = t_fast - (tau + Tp/2): one subtraction per target-sample instead of two
            s_pb[threadIdx.x] = tp.y;
            s_amp[threadIdx.x] = a.amp_pt ? a.amp_pt[(size_t)i * a.n_targets + b] : a.amp[b];
        }
        __syncthreads();
        const int nb = min(ECHO_THREADS, a.n_targets - b0);
        for (int k = 0; k < nb; ++k) {
            const double u = tf - s_tau[k];                      // (:290,293 / :164,166)
            const float gate = (fabs(u) <= half_tp) ? s_amp[k] : 0.f;
            // phase_base + pi k u^2 in revolutions; its fractional part by v_fract_f64 (one instruction; p - rint(p) is two)
            const cf e = cis_frac((float)__builtin_amdgcn_fract(fma(hk * u, u, s_pb[k])));
            acc_re = fmaf(gate, e.x, acc_re);
            acc_im = fmaf(gate, e.y, acc_im);
        }
    }
    if (live) {
        cf* o = a.out + (size_t)i * a.n_samples + j;
        if (a.accumulate) { const cf x = *o; acc_re += x.x; acc_im += x.y; }
        *o = make_float2(acc_re, acc_im);
    }
}

// ------------------------------------------------------------------------------
// Per pulse and target geometry on the device (fp64, the reference's formulas): delay tau, carrier phase in
// revolutions, and for the spotlight model the amplitude rcs * sinc^2 antenna gain.  The reference does this part in
// NumPy / torch on [pulses x targets x 3] arrays (36 M pairs and 0.9 GB per clutter call of the two-channel script);
// here it is one small launch whose table the sample kernel above consumes without a host round trip.
//   model 0  run_physics_engine        (sar_satellite_sim.py:268-272): tau = 2 d / C, pb = -2 FC d / C; with a target
//            velocity run_moving_physics (sar_satellite_moving_sim.py:137-145); run_custom_physics (sar_vehicle_sim.py:108-114)
//   model 1  run_bistatic_physics_gpu  (sar_ati_dcpa_sim_csa.py:151-160): targets move, tau = (d_tx + d_rx) / C, pb = -FC tau
//   model 2  run_physics_spotlight     (sar_batch_sim.py:127-150): receiver displaced by v_sat * 2 d_tx / C, antenna gain
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void echo_geometry_kernel(EchoGeoArgs a) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (b >= a.n_targets) return;
    const double t = a.t_pulse ? a.t_pulse[i] : 0.0;
    double px = a.tgt_pos[3 * b], py = a.tgt_pos[3 * b + 1], pz = a.tgt_pos[3 * b + 2];
    if (a.tgt_vel) { px += a.tgt_vel[0] * t; py += a.tgt_vel[1] * t; pz += a.tgt_vel[2] * t; }   // model 0: optional (run_moving_physics)
    const double sx = a.tx_pos[3 * i], sy = a.tx_pos[3 * i + 1], sz = a.tx_pos[3 * i + 2];
    const double dx = px - sx, dy = py - sy, dz = pz - sz;
    const double d_tx = sqrt(dx * dx + dy * dy + dz * dz);
    double tau, pb;
    if (a.model == 0) {
        tau = 2.0 * d_tx / a.c;
        pb = -2.0 * a.fc * d_tx / a.c;
    } else if (a.model == 1) {
        const double ex = px - a.aux[3 * i], ey = py - a.aux[3 * i + 1], ez = pz - a.aux[3 * i + 2];   // aux = receiver position
        const double d_rx = sqrt(ex * ex + ey * ey + ez * ez);
        tau = (d_tx + d_rx) / a.c;
        pb = -a.fc * tau;
    } else {
        const double tau_a = 2.0 * d_tx / a.c;                                                      // aux = platform velocity
        const double rx = sx + a.aux[3 * i] * tau_a, ry = sy + a.aux[3 * i + 1] * tau_a, rz = sz + a.aux[3 * i + 2] * tau_a;
        const double ex = px - rx, ey = py - ry, ez = pz - rz;
        const double d_rx = sqrt(ex * ex + ey * ey + ez * ez);
        tau = (d_tx + d_rx) / a.c;
        pb = -a.fc * tau;
        // look direction: scene centre (origin) seen from the platform (:134-139)
        const double bn = sqrt(sx * sx + sy * sy + sz * sz);
        double cos_off = ((-sx / bn) * (dx / d_tx) + (-sy / bn) * (dy / d_tx) + (-sz / bn) * (dz / d_tx));
        cos_off = cos_off < -1.0 ? -1.0 : (cos_off > 1.0 ? 1.0 : cos_off);
        const double x = M_PI * a.l_ant * sin(acos(cos_off)) / a.lambda;                           // :140
        double gain = 1.0;
        if (fabs(x) > 1e-6) { const double q = sin(x) / x; gain = q * q; }                        // :141-144
        a.amp_pt[(size_t)i * a.n_targets + b] = (float)(a.rcs[b] * gain);                          // :150
    }
    a.tau_pb[(size_t)i * a.n_targets + b] = make_double2(tau, pb);
}

hipError_t launch_echo_geometry(const EchoGeoArgs& a, hipStream_t st) {
    dim3 grid((a.n_targets + 255) / 256, a.n_pulses);
    hipLaunchKernelGGL(echo_geometry_kernel, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_echo_synth(const EchoArgs& a, hipStream_t st) {
    dim3 grid((a.n_samples + ECHO_THREADS - 1) / ECHO_THREADS, a.n_pulses);
    hipLaunchKernelGGL(echo_synth_kernel, grid, dim3(ECHO_THREADS), 0, st, a);
    return hipGetLastError();
}

}  // namespace sarx
