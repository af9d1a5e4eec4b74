#!/usr/bin/env python3
"""bench.py - focused SAR frames/sec + HBM GB/s of the range FFT+phase pass.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 16384]

One step = one CSA focus (azimuth FFT+Phi1, range FFT+Phi2+IFFT+Phi3, azimuth
IFFT) of one 16384x16384 complex64 frame whose echo is already resident in HBM
(synthetic complex Gaussian noise generated on the device).  For N > 1 the
driver starts one process per GPU (torch.distributed.run); frames shard
one-per-rank per step (weak scaling) and each step's frame is multilooked
16x16 into its VideoSAR stack slot and all-gathered with RCCL on a second
stream.  torch is used only for the gloo rendezvous/barrier; the GPU path is
libsarx through ctypes.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "nis-sar-amtigmti-video_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)
LOOKS = 16


def cpu_baseline(size, workers=1):
    """Oracle ("port") timed on this box's host cores on a bounded sample: single thread (what the reference's
    NumPy code uses) by default; workers > 1 = the same arithmetic with threaded FFTs and row blocks in a pool."""
    import numpy as np
    from oracle import csa_oracle as orc
    n = min(size, 8192)                      # ~10-20 s of single-thread CPU work
    k = orc.scaled_radar(n, n)
    rng = np.random.default_rng(0)
    raw = (rng.standard_normal((n, n), dtype=np.float32) + 1j * rng.standard_normal((n, n), dtype=np.float32))
    raw = raw.astype(np.complex64)
    best = 1e30
    for _ in range(1):
        t = time.perf_counter()
        orc.sar_focus_csa_lean(raw, *orc.focus_args(k), workers=workers, block=64 if workers > 1 else 512)
        best = min(best, time.perf_counter() - t)
    scale = (size / n) ** 2                      # samples per full frame / samples in the sample
    return {"value": 1.0 / (best * scale), "unit": "frames/s", "cores": workers, "kind": "port",
            "sample": f"{n}x{n} complex64 noise frame, oracle/csa_oracle.sar_focus_csa_lean (NumPy/scipy.fft, "
                      f"{workers} thread(s) of {os.cpu_count()}), {best:.2f} s, scaled x{scale:.0f} by sample count to "
                      f"{size}x{size}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--unfused", action="store_true", help="run range passes 2 and 3 as two launches")
    ap.add_argument("--passes", action="store_true", help="also time every pass alone (stderr)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import sarx
    from sarx import _ffi, radar

    n_dev = max(1, sarx.device_count())
    ctx = sarx.Context(local_rank % n_dev)              # one rank per GPU; wraps only when ranks outnumber GPUs (tests)
    n = a.size
    K, W = a.steps, a.warmup
    flags = 0 if a.unfused else _ffi.FUSE_RANGE
    plan = sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=flags)
    d_in, d_img = ctx.alloc(n * n * 8), ctx.alloc(n * n * 8)
    ctx.fill_noise(d_in, n * n, 1000 + rank)

    comm = host_comm = d_slot = d_recv = None
    collective = None
    slot_bytes = (n // LOOKS) * (n // LOOKS) * 4
    force_comm = os.environ.get("SARX_BENCH_FORCE_COMM") == "1"     # exercise the gather path on one GPU
    if world > 1 or force_comm:
        def bootstrap(uid):
            if dist is None:
                return uid
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        # RCCL over xGMI is the collective.  If its bootstrap fails on every rank alike (it needs dmabuf IPC,
        # HSA_ENABLE_IPC_MODE_LEGACY=0), the stack slots travel through the gloo group instead - slower, said so in
        # the JSON line - rather than losing the whole scaling run.
        uid, ok = None, 1
        if rank == 0:
            try:
                uid = ctx.comm_unique_id()
            except Exception as exc:                          # noqa: BLE001
                print(f"[bench rank 0] RCCL unique id failed: {exc}", file=sys.stderr, flush=True)
        uid = bootstrap(uid)                                  # every rank takes part, also after a failure on rank 0
        if uid is None:
            ok = 0
        else:
            try:
                ctx.comm_init(uid, world, rank)
                comm = True
            except Exception as exc:                          # noqa: BLE001
                ok = 0
                print(f"[bench rank {rank}] RCCL init failed: {exc}", file=sys.stderr, flush=True)
        if dist is not None:
            import torch
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            collective = "RCCL all-gather of the stack slot"
            d_slot = ctx.alloc(slot_bytes * 2)                # double-buffered send slots
            d_recv = ctx.alloc(slot_bytes * world * 2)        # double-buffered round blocks
        else:
            if comm is not None:
                ctx.lib.sarx_comm_destroy(ctx.h)
            comm = None
            from sarx.batch import TorchStackComm
            host_comm = TorchStackComm() if dist is not None else None
            collective = "gloo all-gather of the stack slot through host memory (RCCL init failed)"
            d_slot = ctx.alloc(slot_bytes * 2)

    import numpy as np

    def step(s, mark):
        if mark and 2 * s + 1 < 256:
            plan.mark_range(2 * s, 2 * s + 1)
        else:
            plan.mark_range(-1, -1)
        plan.focus_dev(d_in, d_img)
        if comm is not None:
            ctx.comm_fence_compute()                      # slot (s&1) was last read by the gather of step s-2
            ctx.lib.sarx_multilook_dev(ctx.h, d_img.ptr, d_slot.ptr + (s & 1) * slot_bytes, n, n, LOOKS)
            ctx.lib.sarx_allgather_dev(ctx.h, d_slot.ptr + (s & 1) * slot_bytes,
                                       d_recv.ptr + (s & 1) * slot_bytes * world, slot_bytes)
        elif host_comm is not None:
            ctx.lib.sarx_multilook_dev(ctx.h, d_img.ptr, d_slot.ptr, n, n, LOOKS)
            host_comm.all_gather(d_slot.download(np.float32, (n // LOOKS, n // LOOKS)))

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    for s in range(W):
        step(s, False)
    barrier()
    t0 = time.perf_counter()
    for s in range(K):
        step(s, True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant kernel = the range pass; events were recorded on the kernel's own stream
    marked = min(K, 127)
    rg_ms = sum(ctx.elapsed_ms(2 * s, 2 * s + 1) for s in range(marked)) / max(marked, 1)
    launches = 1 if not a.unfused else 2
    alg_bytes = 16.0 * n * n * launches            # one c64 read + one c64 write per sample per launch
    achieved = alg_bytes / (rg_ms * 1e-3) / 1e9

    import numpy as np
    probe = d_img.download(np.complex64, (4, n))
    assert np.isfinite(probe).all() and np.abs(probe).max() > 0, "focused image is not finite / all zero"

    # every pass alone, after the timed region (HIP events on the ctx stream, 5 launches each)
    per_pass = {}
    if rank == 0 and world == 1:
        tmp = ctx.alloc(n * n * 8)
        names = {_ffi.PASS_AZ_FFT_PHI1: ("az_fft_phi1", 2), _ffi.PASS_RG_FFT_PHI2: ("rg_fft_phi2", 1),
                 _ffi.PASS_RG_IFFT_PHI3: ("rg_ifft_phi3", 1), _ffi.PASS_RG_FUSED_23: ("rg_fused_fft_phi2_ifft_phi3", 1),
                 _ffi.PASS_AZ_IFFT: ("az_ifft", 2)}
        for pid, (nm, launches_) in names.items():
            plan.run_pass(pid, d_in, tmp)
            ctx.sync()
            ctx.record(250)
            for _ in range(5):
                plan.run_pass(pid, d_in, tmp)
            ctx.record(251)
            ms = ctx.elapsed_ms(250, 251) / 5
            per_pass[nm] = {"ms": round(ms, 4), "launches": launches_,
                            "GBps_per_launch": round(16.0 * n * n * launches_ / ms / 1e6, 1)}
            if a.passes:
                print(f"[pass] {nm:30s} {ms:8.3f} ms  {launches_} launch(es)  "
                      f"{16.0 * n * n * launches_ / ms / 1e6:8.1f} GB/s per launch at 16 B/sample", file=sys.stderr)
        tmp.release()

    # HBM bytes per launch from PMC counters (collected in separate rocprofv3 --pmc runs, see the file)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        if n == 16384 and not a.unfused:
            traffic = next(v["hbm_bytes_per_launch"] for k, v in pmc["kernels"].items() if "range_fused_wl" in k)
    except (OSError, StopIteration, KeyError, ValueError):
        traffic = None

    if rank == 0:
        line = {
            "metric": "focused SAR frames/sec (CSA focus, complex64)", "value": world * K / dt, "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "c64 (phase arguments f64)", "data": "synthetic",
            "config": {"workload": f"{n}x{n} complex64 single-channel CSA focus, echo resident in HBM",
                       "frames_per_step_per_gpu": 1, "range_passes": "fused 2+3" if not a.unfused else "separate",
                       "image_layout": "[n_az x n_rg]; img.T returned as a view like the reference",
                       "parallelism": f"frames sharded 1/GPU x{world}" +
                                      (f"; 16x16 multilook + {collective} per step" if collective else "")},
            "roofline": {"bound": "hbm", "kernel": ("range_fused_wl_kernel (FFT.Phi2.IFFT.Phi3, one HBM round trip)"
                                                    if (not a.unfused and n == 16384) else
                                                    "range_pass_kernel<fused>" if not a.unfused else
                                                    "range_pass_v2_kernel<FFT+Phi2>, <IFFT+Phi3>"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "launch_ms": rg_ms / launches, "algorithmic_bytes_per_launch": 16.0 * n * n},
        }
        if not a.unfused:
            # SURVEY.md 8(d) counts 16 B/sample per FFT(+phase) pass; this launch does two of them (FFT+Phi2, IFFT+Phi3)
            # on one HBM round trip.  `achieved` counts the bytes it really moves (16 B/sample); by pass accounting:
            line["roofline"]["survey_passes_in_launch"] = 2
            line["roofline"]["achieved_by_pass_accounting"] = 2 * achieved
        if per_pass:
            # the standalone range FFT + Phi_2 launch BASELINE.json's 70 % target names, and the others
            p2 = per_pass["rg_fft_phi2"]
            line["roofline_rg_fft_phi2_pass"] = {
                "bound": "hbm", "achieved": p2["GBps_per_launch"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": p2["GBps_per_launch"] / HBM_PEAK_GBS, "launch_ms": p2["ms"],
                "note": "same launch outside the timed region; the default path runs it fused with pass 3"}
            line["passes"] = per_pass
        if world == 1 and not a.no_cpu:
            line["cpu_baseline"] = cpu_baseline(n)
            mt = min(os.cpu_count() or 1, 32)
            if mt > 1:                              # the fair all-core figure next to the reference-style single thread
                line["cpu_baseline_threads"] = cpu_baseline(n, workers=mt)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
