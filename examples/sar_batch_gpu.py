#!/usr/bin/env python3
"""The reference's VideoSAR batch (sar_batch_sim.py main, :240-331) on the MI355X through sarx: per frame a
spotlight echo of the moving destroyer (:308-311), thermal noise + sea clutter (:313-314) and a time-domain
back-projection image focused at the target velocity ("mBP") or at rest ("StdBP") (:316-321).  The raw pulses
never leave the GPU between the three steps.

    python examples/sar_batch_gpu.py [--frames 46] [--cpi-pulses 2500] [--nx 512] [--headings 0 90 45 135]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/sar_batch_gpu.py ...   # frames f -> rank f mod N

Frames are independent (:303-331), so with N ranks each focuses every N-th frame on its own GPU and the stack is
reassembled by an all-gather per round (sarx.batch; 2 MiB per frame, through the gloo group).  Rank 0 writes the files.

Writes one <run_id>.npz per (heading, algorithm) with the frame stack the reference animates (:333-337):
frames [n x ny x nx] complex64, g_max, extent.
"""
import argparse
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # rank start, before `import sarx` / any GPU call (dmabuf IPC for RCCL)

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
import sarx  # noqa: E402
from sarx.targets import generate_destroyer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=None, help="default: all that fit in 5 s (46 of 50)")
    ap.add_argument("--cpi-pulses", type=int, default=None, help="default ceil(0.5 s * PRF) = 2500 (:249)")
    ap.add_argument("--nx", type=int, default=512)
    ap.add_argument("--headings", type=float, nargs="*", default=[0, 90, 45, 135])            # :281
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--outdir", default="batch_output")
    a = ap.parse_args()
    os.makedirs(a.outdir, exist_ok=True)

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    from sarx.batch import LocalStackComm, TorchStackComm, run_batch_host
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    k = sarx.batch_constants()                                        # :12-50
    PRF, Lambda, R0, BW = k["PRF"], k["Lambda"], k["R0"], k["BW"]
    DURATION, FPS = 5.0, 10                                           # :244-246
    NUM_FRAMES = a.frames or int(DURATION * FPS)
    TOTAL_PULSES = int(np.ceil(DURATION * PRF))
    STEP_PULSES = int(PRF / FPS)
    CPI_PULSES = a.cpi_pulses or int(np.ceil(0.5 * PRF))
    t_vec_all = np.linspace(-2.5, 2.5, TOTAL_PULSES)                  # :258
    pos_sat_all, vel_sat_all = sarx.orbit_arc(t_vec_all, k)           # :262-268

    v = {"name": "Destroyer", "speed": 15.0, "swath": 500.0}          # :277
    base_target = generate_destroyer(center_pos=(0, 0, 0))
    L_ANT = Lambda * R0 / v["swath"]                                  # :297
    snr_db_raw = sarx.calculate_raw_snr_db(R0, 5000.0, Lambda, BW, L_ANT, consts=k)           # :298
    ctx = sarx.default_context(int(os.environ.get("LOCAL_RANK", "0")) % max(1, sarx.device_count()))
    frame_ids = [f for f in range(NUM_FRAMES) if f * STEP_PULSES + CPI_PULSES <= TOTAL_PULSES]      # :305-307
    for h in a.headings:
        for algo, focus_tgt in (("mBP", True), ("StdBP", False)):     # :283-286
            run_id = f"{v['name']}_{int(v['speed'])}_{int(h)}_{algo}"
            state = {"d_raw": None, "n_sp": 0, "local_max": 0.0}
            t0 = time.time()

            def enqueue_frame(f, d_raw, d_image=None):
                """One frame of :303-331 on the current lane: echo (:308-311), noise relative to max |raw|^2 taken on the device
                (:313-314; sarx.add_noise_rel_dev = power_stats + add_noise_dev without the host round trip, same samples), TDBP
                (:318-321).  With d_image the call only enqueues; without it the image comes back to the host."""
                i0 = f * STEP_PULSES
                t_cpi, p_cpi, v_cpi = t_vec_all[i0:i0 + CPI_PULSES], pos_sat_all[i0:i0 + CPI_PULSES], vel_sat_all[i0:i0 + CPI_PULSES]
                d_raw, t_st, n_sp, v_tgt = sarx.run_physics_spotlight(base_target, t_cpi, p_cpi, v_cpi, heading_deg=h,
                                                                      speed=v["speed"], l_ant=L_ANT, consts=k, device=True,
                                                                      out=d_raw, ctx=ctx, sync=d_image is None)
                state["n_sp"] = n_sp
                sarx.add_noise_rel_dev(d_raw, len(t_cpi) * n_sp, snr_db_raw + k["SNR_BOOST_DB"], k["SCR_DB"], k["K_NU"],
                                       seed=a.seed * 100003 + f, ref="max")
                vf = v_tgt if focus_tgt else np.zeros(3)
                img = sarx.tdbp_gpu(d_raw, p_cpi, v_cpi, t_st, n_sp, vel_focus=vf, t_pulses=t_cpi,
                                    scene_size=v["swath"], nx=a.nx, ny=a.nx, consts=k, ctx=ctx, d_image=d_image)
                return d_raw, img

            if world == 1:
                # one GPU: two frames in flight on two lanes of the context, each lane with its own pulse buffer (and, inside
                # tdbp_gpu, its own plan); no host synchronisation inside the loop - the frames land in one device stack that is
                # downloaded once at the end
                class _At:
                    def __init__(self, ptr):
                        self.ptr = ptr
                slot = a.nx * a.nx * 16
                d_stack = ctx.alloc(max(len(frame_ids), 1) * slot)
                d_raws = [None, None]
                lane_ids = ctx.concurrent_lanes(2)              # two lanes that really run side by side
                for i, f in enumerate(frame_ids):
                    ctx.select_lane(lane_ids[i & 1])
                    d_raws[i & 1], _ = enqueue_frame(f, d_raws[i & 1], _At(d_stack.ptr + i * slot))
                ctx.select_lane(0)
                ctx.lanes_join()
                stack = d_stack.download(np.complex128, (len(frame_ids), a.nx, a.nx)).astype(np.complex64)
                state["local_max"] = float(np.abs(stack).max()) if stack.size else 0.0
                for b in (d_stack, *d_raws):
                    if b is not None:
                        b.release()
                comm = LocalStackComm()
            else:
                def process_frame(f):
                    state["d_raw"], img = enqueue_frame(f, state["d_raw"])                         # one pulse buffer for all frames
                    img = img.astype(np.complex64)
                    state["local_max"] = max(state["local_max"], float(np.abs(img).max()))         # this rank's share of g_max (:337)
                    return img.view(np.float32)                                                    # stack slot [ny x 2 nx]

                comm = TorchStackComm()
                stack = np.ascontiguousarray(run_batch_host(frame_ids, world, rank, process_frame, comm,
                                                            slot_shape=(a.nx, 2 * a.nx))).view(np.complex64)
                if state["d_raw"] is not None:
                    state["d_raw"].release()
            ctx.sync()
            # g_max = max over ALL frames of max|frame| (:337-338): each rank has reduced the frames it focused, one float is
            # all-reduced with max (every rank takes part) - nobody scans the gathered stack for it
            g_max = comm.all_reduce_max(state["local_max"]) or 1.0
            dt = time.time() - t0
            if rank != 0:
                continue
            frames, n_sp, t_cpi = stack, state["n_sp"], t_vec_all[:CPI_PULSES]
            out = os.path.join(a.outdir, run_id + ".npz")
            np.savez(out, frames=stack, g_max=g_max, extent=np.array([-v["swath"] / 2, v["swath"] / 2] * 2), fps=FPS)
            print(f"{run_id}: {len(frames)} frames of {len(t_cpi)} pulses x {n_sp} samples -> {a.nx}x{a.nx} in {dt:.2f} s "
                  f"({len(frames) / dt:.1f} frames/s), g_max {g_max:.4g} -> {out}")
    if world > 1:
        # leave together: a rank without a file to write reached the end of the script while rank 0 was still saving, and a gloo
        # process group torn down under a live peer can abort the process (SIGABRT out of a background thread, seen once in nine runs)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
