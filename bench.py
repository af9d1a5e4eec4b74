#!/usr/bin/env python3
"""bench.py - focused SAR frames/sec + HBM GB/s of the range FFT+phase pass.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 16384] [--stack multilook|magnitude]

One step = one CSA focus (azimuth FFT+Phi1, range FFT+Phi2+IFFT+Phi3, azimuth IFFT) of one 16384x16384 complex64
frame whose echo is already resident in HBM (synthetic complex Gaussian noise generated on the device): BASELINE.json
config 4, the configuration the metric is quoted on.

--gpus N > 1: one process per GPU.  Started bare (`python bench.py --gpus N`, WORLD_SIZE unset) this file spawns the N
ranks itself as fresh child processes (`python -m torch.distributed.run ... bench.py ...`) BEFORE anything touches the
GPU, relays rank 0's JSON line and exits with the children's code; started under torch.distributed.run it is one of
the ranks (RANK / LOCAL_RANK / WORLD_SIZE from the environment).  `--gpus` disagreeing with WORLD_SIZE is an error,
never a silent 1-GPU run.  At N > 1 frames shard one per rank per step (weak scaling) and each step's frame is
multilooked 16x16 into its VideoSAR stack slot and all-gathered with RCCL on a second stream.

Every line also carries a `batch64` block: BASELINE.json config 5, the 64-frame VideoSAR batch of two-channel 8192^2
scenes, frame f -> rank f mod N, strong scaling (sarx.batch.TwoChannelBatch, the driver the tests run too).

torch is used only for the gloo rendezvous/barrier; the GPU path is libsarx through ctypes.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "nis-sar-amtigmti-video_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)
XGMI_LINK_GBS = 153.0          # one xGMI link, per direction (7 links per GPU, point to point)
LOOKS = 16


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE or 1)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--in-flight", type=int, default=None,
                    help="frames in flight per GPU: consecutive (independent) frames are enqueued on this many alternating compute lanes of the "
                         "context, each lane with its own plan and buffers, so kernels of neighbouring frames overlap (1 = one frame at a time; "
                         "default: 2 for frames of 4096^2 and more, 1 below - small frames are launch-bound and lose to the lane switches)")
    ap.add_argument("--batch-lanes", type=int, default=None, help="frames in flight of the batch64 block (default: the batch driver's own rule)")
    ap.add_argument("--range-cus", type=int, default=None,
                    help="compute units the persistent range launch sizes its grid for while frames are in flight (default 192 of 256; 0 = all)")
    ap.add_argument("--unfused", action="store_true", help="run range passes 2 and 3 as two launches")
    ap.add_argument("--passes", action="store_true", help="also print every pass alone to stderr")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-sample", action="store_true",
                    help="cpu_baseline on an 8192^2 sample scaled by sample count (~8 s) instead of the real frame on one thread (~30 s at 16384^2)")
    ap.add_argument("--no-batch", action="store_true", help="skip the batch64 block (BASELINE config 5)")
    ap.add_argument("--batch-frames", type=int, default=64)
    ap.add_argument("--batch-size", type=int, default=8192)
    ap.add_argument("--stack", choices=("multilook", "magnitude", "products", "both", "all", "default"), default="default",
                    help="batch64 stack slot: 16x16 multilook (1 MiB/frame at 8192^2; headline), full-resolution "
                         "magnitude (256 MiB/frame: loads xGMI), products (masked ATI phase, |slc1|, DPCA magnitude: 768 MiB/frame, "
                         "SURVEY.md 8(e)'s product stack), both = multilook + magnitude, all = the three one after the other; "
                         "default = multilook + products")
    ap.add_argument("--batch-scene", choices=("noise", "c3"), default="noise",
                    help="batch64 content: device noise (bandwidth does not depend on content) or SURVEY.md 8(d) C5's scene - the C3 point-target "
                         "scene with its movers advanced by f * 0.1 s per frame, synthesised on the device before the clock starts")
    ap.add_argument("--batch-reps", type=int, default=3, help="timed repetitions of the batch64 block (the median is reported; keeps the GPU busy long enough to be sampled)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `config2_4096` / `config3_two_channel` blocks (BASELINE configs 2 and 3; GPU legs of < 0.1 s each, on by default at N = 1)")
    ap.add_argument("--config3-cpu", action="store_true",
                    help="give the `config3_two_channel` block its own cpu_baseline (oracle: 2 x focus + the sar_ati_dcpa_sim_csa.py:414-419,447-449 "
                         "expressions; ~1 min)")
    ap.add_argument("--require-rccl", action="store_true",
                    help="RCCL must come up on every rank or every rank exits non-zero together; this is already the behaviour whenever the ranks sit on "
                         "distinct devices (N > 1 and at least N GPUs visible)")
    ap.add_argument("--allow-host-transport", action="store_true",
                    help="permit the gloo-through-host-memory fallback even with one GPU per rank (rehearsals only: such a line is not an xGMI measurement)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous only: every rank reports (rank, local rank, world) and exits before touching the GPU")
    return ap.parse_args(argv)


def spawn_ranks(n, argv):
    """Start n ranks as fresh child processes (this process has not touched the GPU and never will)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # RCCL across processes needs dmabuf IPC on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"[bench] spawning {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def rccl_required(world, n_dev, a):
    """One GPU per rank (the driver's scaling run: N > 1 ranks, at least N devices visible) or --require-rccl: a line whose slots
    travelled through host memory would be a host-memory number under an xGMI heading, so RCCL must come up.  Ranks outnumbering
    GPUs (a rehearsal on one card) and --allow-host-transport may fall back to gloo, and the line says so."""
    if world <= 1:
        return False
    return bool(a.require_rccl or (n_dev >= world and not a.allow_host_transport))


def exit_together_without_rccl(dist, log):
    """RCCL is required and did not come up: every rank leaves with code 3.  `use_rccl` is the result of a MIN-reduce over all ranks
    (sarx.batch.agree_on_rccl), so every rank takes this branch or none does; the barrier keeps the process group alive until all have."""
    log("RCCL did not come up on every rank and this run requires it (one GPU per rank, or --require-rccl): exiting 3 on every rank")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(3)


class _RehearsalCtx:
    """--dry-run only (no GPU): stands in for sarx.Context in agree_on_rccl so the launcher tests can rehearse the RCCL policy with
    gloo ranks; SARX_BENCH_REHEARSE_RCCL_FAIL lists the ranks whose communicator 'fails' (-1: the unique id itself fails)."""

    def __init__(self, rank, failing):
        self.rank, self.failing, self.h = rank, failing, None
        self.lib = self

    def comm_unique_id(self):
        if -1 in self.failing:
            raise RuntimeError("rehearsal: ncclGetUniqueId fails")
        return b"\0" * 128

    def comm_init(self, uid, world, rank):
        if rank in self.failing:
            raise RuntimeError(f"rehearsal: ncclCommInitRank fails on rank {rank}")

    def sarx_comm_destroy(self, h):
        return 0


def noise_frame(n, seed=0):
    """n x n complex64 Gaussian noise on the host, built in pieces (no n x n float64 temporaries)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    raw = np.empty((n, n), dtype=np.complex64)
    for i0 in range(0, n, 1024):
        blk = rng.standard_normal((min(1024, n - i0), n, 2), dtype=np.float32)
        raw[i0:i0 + blk.shape[0]] = blk[..., 0] + 1j * blk[..., 1]
    return raw


def cpu_baseline(size, workers=1, scaled=True, frame=None, keep_image=False):
    """Oracle ("port") timed on this box's host cores.  scaled: an 8192^2 sample on one thread (how the reference's
    NumPy runs), scaled x4 by sample count; not scaled: the real size x size frame, row-blocked, `workers` threads.
    frame: (host echo, radar args) of the frame the GPU legs focused - the CPU leg then runs on THAT frame and, with
    keep_image, its image ([n_az x n_rg] complex128) is returned beside the block for the parity figure."""
    from oracle import csa_oracle as orc
    if frame is not None:
        raw, args = frame
        n = raw.shape[0]
    else:
        n = min(size, 8192) if scaled else size
        raw, args = noise_frame(n), orc.focus_args(orc.scaled_radar(n, n))
    t = time.perf_counter()
    img = orc.sar_focus_csa_lean(raw, *args, workers=workers, block=64 if workers > 1 else 512)[0]
    best = time.perf_counter() - t
    scale = (size / n) ** 2                      # samples per full frame / samples in the sample
    how = (f"scaled x{scale:.0f} by sample count to {size}x{size}" if scale != 1 else "the full frame, not scaled")
    what = ("the frame the GPU legs focused (lane 0's echo, downloaded once)" if frame is not None else "complex64 noise frame")
    blk = {"value": 1.0 / (best * scale), "unit": "frames/s", "cores": workers, "kind": "port",
           "sample": f"{n}x{n} {what}, oracle/csa_oracle.sar_focus_csa_lean (NumPy/scipy.fft, "
                     f"{workers} thread(s) of {os.cpu_count()}), {best:.2f} s, {how}"}
    return (blk, img.T) if keep_image else blk


def parity_figures(gpu_img, cpu_img, rows=256):
    """Relative L2 of the GPU image against the oracle's complex128 image of the same echo, |img| and complex
    (north_star's bar: <= 1e-4), accumulated row block by row block in float64 - no full-size temporaries."""
    import numpy as np
    num_c = num_m = den = 0.0
    for i0 in range(0, gpu_img.shape[0], rows):
        g = gpu_img[i0:i0 + rows].astype(np.complex128)
        c = cpu_img[i0:i0 + rows]
        num_c += float(np.sum(np.abs(g - c) ** 2))
        num_m += float(np.sum((np.abs(g) - np.abs(c)) ** 2))
        den += float(np.sum(np.abs(c) ** 2))
    return {"rel_l2_mag": (num_m / den) ** 0.5, "rel_l2_complex": (num_c / den) ** 0.5, "bar": 1e-4}


def cpu_baseline_two_channel(n_full, workers, n_sample):
    """Oracle ("port") for BASELINE config 3: 2 x sar_focus_csa_lean + the ATI / DPCA / mask expressions of
    sar_ati_dcpa_sim_csa.py:414-419,447-449 on an n_sample^2 two-channel noise frame, scaled by sample count to n_full^2."""
    import numpy as np
    from oracle import csa_oracle as orc
    k = orc.scaled_radar(n_sample, n_sample)
    chans = [noise_frame(n_sample, seed) for seed in (0, 1)]
    t = time.perf_counter()
    s1 = orc.sar_focus_csa_lean(chans[0], *orc.focus_args(k), workers=workers, block=64 if workers > 1 else 512)[0]
    s2 = orc.sar_focus_csa_lean(chans[1], *orc.focus_args(k), workers=workers, block=64 if workers > 1 else 512)[0]
    t_focus = time.perf_counter() - t
    t = time.perf_counter()
    ati_phase = np.angle(s1 * np.conj(s2))            # :414-415
    slc1_mag = np.abs(s1)                             # :416
    dpca_mag = np.abs(s1 - s2)                        # :418-419
    ati_phase[~(slc1_mag > slc1_mag.max() * 0.05)] = 0   # :447-449
    t_prod = time.perf_counter() - t
    assert np.isfinite(dpca_mag).all()
    scale = (n_full / n_sample) ** 2
    return {"value": 1.0 / ((t_focus + t_prod) * scale), "unit": "frames/s", "cores": workers, "kind": "port",
            "sample": f"two-channel {n_sample}x{n_sample} complex64 noise frame: oracle focus x2 {t_focus:.2f} s ({workers} thread(s) of "
                      f"{os.cpu_count()}) + ATI/DPCA/mask expressions (NumPy, 1 thread) {t_prod:.2f} s" +
                      (f", scaled x{scale:.0f} by sample count to {n_full}x{n_full}" if scale != 1 else ", the full frame")}


def run_single_channel(sarx, ctx, n, steps=200, lanes=2):
    """BASELINE config 2 (4096^2 single-channel CSA focus on one GPU), echo resident in HBM, the headline region's own form: frame s on
    lane s % lanes, every lane with its own plan, echo and image."""
    from sarx import _ffi, radar
    plans = [sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE) for _ in range(lanes)]
    bufs = [(ctx.alloc(n * n * 8), ctx.alloc(n * n * 8)) for _ in range(lanes)]
    for i, (d_in, _) in enumerate(bufs):
        ctx.fill_noise(d_in, n * n, 77 + i)
    ids = ctx.concurrent_lanes(lanes)
    try:
        def run(count):
            for s in range(count):
                ctx.select_lane(ids[s % lanes])
                plans[s % lanes].focus_dev(*bufs[s % lanes])
        run(2 * lanes)
        ctx.sync()
        t0 = time.perf_counter()
        run(steps)
        ctx.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        ctx.select_lane(0)
    import numpy as np
    probe = bufs[0][1].download(np.complex64, (2, n))
    assert np.isfinite(probe).all() and np.abs(probe).max() > 0
    for x in plans:
        x.close()
    for pair in bufs:
        for b in pair:
            b.release()
    return {"metric": "focused SAR frames/sec (CSA focus, complex64)", "value": 1e3 / ms, "unit": "frames/s", "ms_per_frame": ms, "steps": steps,
            "workload": f"{n}x{n} complex64 single-channel CSA focus, echo resident in HBM (BASELINE config 2), {lanes} frames in flight",
            "frame_bandwidth_frac_of_peak_algorithmic": 64.0 * n * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def run_config3(sarx, ctx, n, frames=10, cpu=True):
    """BASELINE config 3 through the function a maintainer calls (sarx.focus_ati_dpca), device arrays in, products left on the device."""
    from sarx import radar
    from sarx.engine import DeviceArray
    px = n * n
    raw = [ctx.alloc(px * 8), ctx.alloc(px * 8)]
    for i, b in enumerate(raw):
        ctx.fill_noise(b, px, 31 + i)
    d1, d2 = (DeviceArray(b, (n, n), owner=False) for b in raw)
    ws = sarx.two_channel_workspace(ctx, n, n)
    args = radar.focus_args(n)
    call = lambda: sarx.focus_ati_dpca(d1, d2, *args, ctx=ctx, pulse_shift=False, return_slc2=False, device_output=True, workspace=ws,
                                       fetch_stats=False)
    for _ in range(2):
        call()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        call()
    ctx.sync()
    ms = (time.perf_counter() - t0) / frames * 1e3
    mx, _ = ctx.ati_stats()
    assert mx > 0
    for b in (*raw, *(v for k, v in ws.items() if k != "shape")):
        b.release()
    blk = {"metric": "two-channel frames/sec (2 x CSA focus + ATI/DPCA + 5 % mask)", "value": 1e3 / ms, "unit": "frames/s", "ms_per_frame": ms,
           "workload": f"two-channel {n}x{n} complex64, both echoes resident in HBM (BASELINE config 3), {frames} frames through "
                       "sarx.focus_ati_dpca(device arrays, device_output, reused workspace): products out of channel 2's last azimuth launch"}
    if cpu:
        blk["cpu_baseline"] = cpu_baseline_two_channel(n, 1, min(n, 4096))
        mt = min(os.cpu_count() or 1, 32)
        if mt > 1:
            blk["cpu_baseline_threads"] = cpu_baseline_two_channel(n, mt, n)
    return blk


def run_batch64(sarx, ctx, a, world, rank, dist, use_rccl, host_comm, barrier, stack):
    """BASELINE config 5 through the shared device driver; returns the block for the JSON line (rank 0) or None."""
    from sarx.batch import TwoChannelBatch
    b = TwoChannelBatch(ctx, a.batch_size, a.batch_frames, world, rank, stack=stack, looks=LOOKS, rccl=use_rccl and world > 1,
                        host_comm=None if (use_rccl or world == 1) else host_comm, scene=a.batch_scene, lanes=a.batch_lanes)
    b.prepare()                                                  # echoes of this rank's frames resident in HBM before the clock
    b.run()                                                      # warm-up batch
    times = []
    for _ in range(max(1, a.batch_reps)):
        barrier()
        t0 = time.perf_counter()
        b.run()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        times.append(dt)
    dt = sorted(times)[len(times) // 2]                            # the median repetition
    import numpy as np
    probe = b.stack(frames=[0, a.batch_frames - 1])
    assert np.isfinite(probe).all() and np.abs(probe).max() > 0, "stack slot not finite / empty"
    slot_bytes = b.slot_bytes
    per_frame = ("focus(channel 1; its last launch also leaves max|slc1|" + (" and the multilooked stack slot" if stack == "multilook" else "") +
                 ") + focus(channel 2; its last launch emits masked ATI phase, |slc1|, DPCA magnitude and never writes slc2)"
                 if b.fused_ati else "focus x 2 + one ATI/DPCA launch" + (" with the mask inside" if b.fused_mask else " + one mask launch"))
    n_rounds, lanes_used = b.n_rounds, b.lanes
    b.close()
    if rank != 0:
        return None
    what = {"multilook": f"{LOOKS}x{LOOKS} multilook of |slc1|^2", "magnitude": "full-resolution |slc1|",
            "products": "masked ATI phase + |slc1| + DPCA magnitude, full resolution (the frame's three GMTI planes, written in place "
                        "by channel 2's last azimuth launch)"}[stack]
    # which stacks can scale: one round = one frame per rank; its all-gather brings (world - 1) slots into every rank, each over its own
    # point-to-point xGMI link, and overlaps the NEXT round's focusing - so a round costs max(compute, slot / link rate)
    compute_round = dt / n_rounds if world == 1 else None
    link_s = slot_bytes / (XGMI_LINK_GBS * 1e9)
    blk = {"metric": "VideoSAR batch frames/sec (two-channel CSA focus + ATI/DPCA + mask, stack all-gather)",
           "value": a.batch_frames / dt, "unit": "frames/s", "batch_s": dt, "batch_s_all": [round(x, 5) for x in times], "scaling": "strong", "n_gpus": world,
           "workload": f"{a.batch_frames} frames x two-channel {a.batch_size}x{a.batch_size} complex64 (BASELINE config 5), "
                       f"frame f -> rank f mod {world}, every frame's two echo channels resident in HBM before the clock starts" +
                       (" (device noise)" if a.batch_scene == "noise" else " (C3 point-target scene, movers advanced by f * 0.1 s)"),
           "stack": what + f", {slot_bytes / 2**20:.0f} MiB per frame, gathered in place once per round of {world} frame(s)",
           "per_frame": per_frame, "frames_in_flight_per_gpu": lanes_used, "gather_bytes_per_rank_per_round": slot_bytes,
           "gather_s_per_round_at_one_xgmi_link": link_s}
    if compute_round is not None:
        blk["compute_s_per_frame_one_gpu"] = compute_round
        blk["link_bound_at_8_gpus"] = bool(link_s > compute_round)
        blk["expected_8_gpu_speedup_bound"] = round(8.0 * min(1.0, compute_round / link_s), 2) if link_s > 0 else 8.0
        blk["scaling_note"] = ("with N ranks a round of N frames costs max(per-frame compute, slot bytes / 153 GB/s per link) because the in-place "
                               "gather of round i overlaps the focusing of round i+1: this stack is " +
                               ("LINK-bound at 8 GPUs (the >= 6x target is not expected to hold for it)" if link_s > compute_round
                                else "compute-bound at 8 GPUs (the gather hides behind the next round)"))
    return blk


def main():
    # rank start, before `import sarx` / any GPU call, whoever launched this rank (our own spawn_ranks or a bare
    # `python -m torch.distributed.run ... bench.py`): RCCL across processes needs dmabuf IPC on this pool
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    argv = sys.argv[1:]
    a = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if a.gpus is not None and a.gpus > 1:                    # bare `python bench.py --gpus N`: become the launcher
            sys.exit(spawn_ranks(a.gpus, argv))
        world = 1
    else:
        world = int(env_world)
        if a.gpus is not None and a.gpus != world:
            print(f"[bench] --gpus {a.gpus} disagrees with WORLD_SIZE={world}; refusing to report a wrong n_gpus",
                  file=sys.stderr, flush=True)
            sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE line, rank 0's JSON: gloo and RCCL print banners to the C-level stdout, so everything a
    # rank writes to fd 1 goes to stderr from here on and the JSON line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    if a.dry_run:                                                # launcher / rendezvous check, no GPU needed
        me = {"rank": rank, "local_rank": local_rank, "world": world, "pid": os.getpid(),
              "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
        ranks = [me]
        rehearsal = None
        if dist is not None:
            ranks = [None] * world
            dist.all_gather_object(ranks, me)
            fail_env = os.environ.get("SARX_BENCH_REHEARSE_RCCL_FAIL")
            if fail_env is not None:          # rehearse the collective bring-up and its policy (SARX_BENCH_REHEARSE_DEVICES = visible GPUs)
                from sarx.batch import agree_on_rccl
                log = lambda m: print(f"[bench rank {rank}] {m}", file=sys.stderr, flush=True)
                failing = {int(x) for x in fail_env.split(",") if x.strip()}
                ok = agree_on_rccl(_RehearsalCtx(rank, failing), world, rank, dist, log)
                n_dev = int(os.environ.get("SARX_BENCH_REHEARSE_DEVICES", str(world)))
                if not ok and rccl_required(world, n_dev, a):
                    exit_together_without_rccl(dist, log)
                rehearsal = {"rccl_up": ok, "required": rccl_required(world, n_dev, a), "devices": n_dev}
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            out = {"dry_run": True, "n_gpus": world, "ranks": ranks}
            if rehearsal is not None:
                out["rccl_rehearsal"] = rehearsal
            emit(out)
        return

    import numpy as np
    import sarx
    from sarx import _ffi, radar
    from sarx.batch import TorchStackComm, agree_on_rccl

    n_dev = max(1, sarx.device_count())
    ctx = sarx.Context(local_rank % n_dev)              # one rank per GPU; wraps only when ranks outnumber GPUs (rehearsals)
    n = a.size
    K, W = a.steps, a.warmup
    flags = 0 if a.unfused else _ffi.FUSE_RANGE
    # Frames are independent (sar_batch_sim.py:303-331): frame s runs on lane s % L of the context, every lane with its own plan
    # (scratch), its own echo and its own image, so the launches of neighbouring frames share the GPU - the issue-bound range launch
    # of one frame beside the bandwidth-bound azimuth launches of the next.  Every step is still one whole focus of one frame.
    if a.in_flight is None:
        a.in_flight = 2 if n >= 4096 else 1
    L = max(1, min(a.in_flight, 4))
    range_cus = a.range_cus if a.range_cus is not None else (192 if L > 1 else 0)
    plans = [sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=flags) for _ in range(L)]
    d_ins = [ctx.alloc(n * n * 8) for _ in range(L)]
    d_imgs = [ctx.alloc(n * n * 8) for _ in range(L)]
    for i in range(L):
        ctx.fill_noise(d_ins[i], n * n, 1000 + rank + 7919 * i)
    plan, d_in, d_img = plans[0], d_ins[0], d_imgs[0]
    lane_ids = ctx.concurrent_lanes(L)                   # lanes that really run side by side (HIP may put two streams on one hardware queue)

    # ---- collective: RCCL over xGMI; gloo through host memory only if RCCL cannot come up on every rank -------------
    use_rccl, host_comm, collective, rccl = False, None, None, None
    slot_bytes = (n // LOOKS) * (n // LOOKS) * 4
    force_comm = os.environ.get("SARX_BENCH_FORCE_COMM") == "1"     # exercise the gather path on one GPU
    d_recv = None
    ranks_devices = None
    if world > 1 or force_comm:
        log = lambda m: print(f"[bench rank {rank}] {m}", file=sys.stderr, flush=True)
        use_rccl = agree_on_rccl(ctx, world, rank, dist, log)
        if not use_rccl and rccl_required(world, n_dev, a):
            exit_together_without_rccl(dist, log)
        try:
            rccl = sarx.Context.rccl_info()
        except sarx.SarxError as exc:
            rccl = {"error": str(exc)}
        if use_rccl:
            collective = "RCCL all-gather (in place) on the comm stream"
        else:
            host_comm = TorchStackComm() if dist is not None else None
            collective = "gloo all-gather through host memory (RCCL did not come up on every rank: see stderr)"
        d_recv = ctx.alloc(slot_bytes * world * 2)                # two round blocks, alternating
        me = {"rank": rank, "local_rank": local_rank, "device": ctx.device_id, "pid": os.getpid()}
        ranks_devices = [me]
        if dist is not None:
            ranks_devices = [None] * world
            dist.all_gather_object(ranks_devices, me)

    # execution span of every timed step's fused range launch from in-kernel clock stamps (sarx_csa_plan_stamp_range): with frames in
    # flight an event pair around the launch also contains its queueing behind the other lane's kernels
    N_STAMP = 128
    stamps_on = (not a.unfused) and n == 16384
    d_stamp = ctx.alloc(N_STAMP * 16)
    stamp_init = np.tile(np.array([2**64 - 1, 0], dtype=np.uint64), N_STAMP)

    def step(s, mark, lanes=L):
        lane = s % lanes
        ctx.select_lane(lane_ids[lane])
        ctx.set_range_cus(range_cus if lanes > 1 else 0)      # frames in flight: the persistent range launch leaves CUs to the other lane
        plan, d_in, d_img = plans[lane], d_ins[lane], d_imgs[lane]
        if mark and 2 * s + 1 < 256:
            plan.mark_range(2 * s, 2 * s + 1)
        else:
            plan.mark_range(-1, -1)
        plan.stamp_range(d_stamp.ptr + 16 * s if (mark and stamps_on and s < N_STAMP) else None)
        if d_recv is None:
            plan.focus_dev(d_in, d_img)
            return
        # the frame's stack slot (16x16 multilook) comes out of the focus itself: row-wise partials from the last azimuth
        # launch + a small finish launch (sarx_csa_plan_set_look_slot), +0.05 ms instead of re-reading the image
        block = d_recv.ptr + (s & 1) * slot_bytes * world
        mine = block + rank * slot_bytes
        if use_rccl:
            ctx.lib.sarx_comm_wait_mark(ctx.h, s & 1)     # block (s&1) was last read/written by the gather of step s-2 (step s-1's still overlaps)
        plan.set_look_slot(LOOKS, mine)
        plan.focus_dev(d_in, d_img)
        if use_rccl:
            ctx.lib.sarx_allgather_dev(ctx.h, mine, block, slot_bytes)
            ctx.lib.sarx_comm_mark(ctx.h, s & 1)
        else:
            if host_comm is not None:
                slot = np.empty((n // LOOKS, n // LOOKS), dtype=np.float32)
                _ffi.check(ctx.lib.sarx_memcpy_d2h(ctx.h, slot.ctypes.data, mine, slot_bytes), ctx.h)
                host_comm.all_gather(slot)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    def stamp_ms(count):
        """mean execution span (ms) of the stamped launches 0 .. count-1; None when nothing was stamped"""
        if not stamps_on or count < 1:
            return None
        st = d_stamp.download(np.uint64, (N_STAMP, 2))[:min(count, N_STAMP)]
        ok = st[:, 1] > 0
        return float(np.mean((st[ok, 1] - st[ok, 0]).astype(np.float64)) * 1e-5) if ok.any() else None      # 100 MHz ticks -> ms

    for s in range(W):
        step(s, False)
    d_stamp.upload(stamp_init)
    barrier()
    t0 = time.perf_counter()
    for s in range(K):
        step(s, True)
    barrier()
    dt = time.perf_counter() - t0
    rg_span_ms = stamp_ms(K)
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant kernel = the range pass; events were recorded on the kernel's own stream (the lane the frame ran on)
    marked = min(K, 127)
    rg_ms = sum(ctx.elapsed_ms(2 * s, 2 * s + 1) for s in range(marked)) / max(marked, 1)
    launches = 1 if not a.unfused else 2
    alg_bytes = 16.0 * n * n * launches            # one c64 read + one c64 write per sample per launch
    achieved = alg_bytes / (rg_ms * 1e-3) / 1e9

    # the same steps with ONE frame in flight (lane 0 only): a frame's own latency, and the range launch with the GPU to itself
    solo = None
    if L > 1:
        Ks = min(K, 40)
        for s in range(2):
            step(s, False, lanes=1)
        d_stamp.upload(stamp_init)
        barrier()
        t1 = time.perf_counter()
        for s in range(Ks):
            step(s, True, lanes=1)
        barrier()
        dt1 = time.perf_counter() - t1
        if dist is not None:
            import torch
            t = torch.tensor([dt1], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt1 = float(t.item())
        rg1 = sum(ctx.elapsed_ms(2 * s, 2 * s + 1) for s in range(Ks)) / Ks
        solo = {"steps": Ks, "ms_per_step": dt1 / Ks * 1e3, "value": world * Ks / dt1, "range_launch_ms": rg1 / launches,
                "range_achieved_GBps": alg_bytes / (rg1 * 1e-3) / 1e9, "range_span_ms": stamp_ms(Ks)}
    for x in plans:
        x.mark_range(-1, -1)
        x.stamp_range(None)
    ctx.select_lane(0)
    ctx.set_range_cus(0)

    # the frame lane 0 focused, for the parity figure of the line: its echo and its image, downloaded once (after every timed region)
    host_frame = None
    if rank == 0 and world == 1 and not a.no_cpu and not a.cpu_sample:
        host_frame = (d_ins[0].download(np.complex64, (n, n)), d_imgs[0].download(np.complex64, (n, n)))

    probe = d_img.download(np.complex64, (4, n))
    assert np.isfinite(probe).all() and np.abs(probe).max() > 0, "focused image is not finite / all zero"

    # every pass alone, after the timed region (HIP events on the ctx stream, 5 launches each)
    per_pass = {}
    if rank == 0 and world == 1:
        tmp = ctx.alloc(n * n * 8)
        names = {_ffi.PASS_AZ_FFT_PHI1: ("az_fft_phi1", 2), _ffi.PASS_RG_FFT_PHI2: ("rg_fft_phi2", 1),
                 _ffi.PASS_RG_IFFT_PHI3: ("rg_ifft_phi3", 1), _ffi.PASS_RG_FUSED_23: ("rg_fused_fft_phi2_ifft_phi3", 1),
                 _ffi.PASS_AZ_IFFT: ("az_ifft", 2)}
        if n == 16384:   # the pair the unfused focus runs: spectrum in permuted order between the two launches (range_wp.hip)
            names[_ffi.PASS_RG_FFT_PHI2_PERM] = ("rg_fft_phi2_permuted_spectrum", 1)
            names[_ffi.PASS_RG_IFFT_PHI3_PERM] = ("rg_ifft_phi3_permuted_spectrum", 1)
        for pid, (nm, launches_) in names.items():
            for _ in range(2):
                plan.run_pass(pid, d_in, tmp)
            ctx.sync()
            reps, rounds_ms = 10, []
            for _ in range(3):                          # three rounds of ten launches: the median round is reported, all are kept
                ctx.record(250)
                for _ in range(reps):
                    plan.run_pass(pid, d_in, tmp)
                ctx.record(251)
                rounds_ms.append(ctx.elapsed_ms(250, 251) / reps)
            ms = sorted(rounds_ms)[1]
            per_pass[nm] = {"ms": round(ms, 4), "ms_rounds": [round(x, 4) for x in rounds_ms], "launches": launches_,
                            "GBps_per_launch": round(16.0 * n * n * launches_ / ms / 1e6, 1)}
            if a.passes:
                print(f"[pass] {nm:30s} {ms:8.3f} ms  {launches_} launch(es)  "
                      f"{16.0 * n * n * launches_ / ms / 1e6:8.1f} GB/s per launch at 16 B/sample", file=sys.stderr)
        if n == 16384:
            # The permuted-spectrum pair exactly as the unfused focus runs it: IN PLACE on one buffer, FFT+Phi2 then IFFT+Phi3, every
            # launch between its own pair of events.  (Out of place the forward launch's time depends on where the two buffers lie
            # relative to each other - 0.72 ms or 0.87-0.93 ms with the output 4 KiB further on, profiles/r03_rgbench_offsets.log - and
            # that is not a property of the kernel; the two lines above keep the out-of-place figures for comparison.)
            _ffi.check(ctx.lib.sarx_memcpy_d2d(ctx.h, tmp.ptr, d_in.ptr, n * n * 8), ctx.h)
            for _ in range(2):
                plan.run_pass(_ffi.PASS_RG_FFT_PHI2_PERM, tmp, tmp)
                plan.run_pass(_ffi.PASS_RG_IFFT_PHI3_PERM, tmp, tmp)
            ctx.sync()
            rounds_f, rounds_i = [], []
            for _ in range(3):
                for i in range(10):
                    ctx.record(3 * i)
                    plan.run_pass(_ffi.PASS_RG_FFT_PHI2_PERM, tmp, tmp)
                    ctx.record(3 * i + 1)
                    plan.run_pass(_ffi.PASS_RG_IFFT_PHI3_PERM, tmp, tmp)
                    ctx.record(3 * i + 2)
                rounds_f.append(sum(ctx.elapsed_ms(3 * i, 3 * i + 1) for i in range(10)) / 10)
                rounds_i.append(sum(ctx.elapsed_ms(3 * i + 1, 3 * i + 2) for i in range(10)) / 10)
            for nm, rr in (("rg_fft_phi2_permuted_spectrum_in_place", rounds_f), ("rg_ifft_phi3_permuted_spectrum_in_place", rounds_i)):
                ms = sorted(rr)[1]
                per_pass[nm] = {"ms": round(ms, 4), "ms_rounds": [round(x, 4) for x in rr], "launches": 1,
                                "GBps_per_launch": round(16.0 * n * n / ms / 1e6, 1)}
                if a.passes:
                    print(f"[pass] {nm:40s} {ms:8.3f} ms  {16.0 * n * n / ms / 1e6:8.1f} GB/s at 16 B/sample", file=sys.stderr)
            chk = tmp.download(np.complex64, (2, n))
            assert np.isfinite(chk).all(), "the in-place pair diverged"
        tmp.release()

    # HBM bytes per launch: PMC counters cannot be read inside this process; the figure is REPLAYED from the separate
    # rocprofv3 --pmc passes kept under profiles/ (tools/pmc_traffic.sh), and labelled as such
    def replay(kernel_substr):
        for name in ("r05_pmc_range_kernels.json", "r04_pmc_range_kernels.json", "r03_pmc_range_kernels.json", "r02_pmc_range_kernels.json", "r01_pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    pmc = json.load(fh)
                val = next(v["hbm_bytes_per_launch"] for k, v in pmc["kernels"].items() if kernel_substr in k)
                return val, f"replayed from profiles/{name} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes), not measured in this run"
            except (OSError, StopIteration, KeyError, ValueError):
                continue
        return None, None
    traffic, traffic_src, traffic_p2, traffic_p2_src, traffic_alone, traffic_alone_src = None, None, None, None, None, None
    if n == 16384 and not a.unfused:
        traffic, traffic_src = replay("range_fused_wl_kernel")
        traffic_alone, traffic_alone_src = replay("range_fused_wl_touch_kernel")
        traffic_p2, traffic_p2_src = replay("range_wp_kernel<2>")

    fused_wl = (not a.unfused) and n == 16384
    kernel_name = ("range_fused_wl_kernel (FFT.Phi2.IFFT.Phi3, one HBM round trip)" if fused_wl else
                   "range_pass_kernel<fused>" if not a.unfused else "range_pass_v2_kernel<FFT+Phi2>, <IFFT+Phi3>")
    # vector instructions per launch of the fused kernel, REPLAYED from the SQ_INSTS_VALU pass kept under profiles/ (it depends on the
    # code, not on the run); peak issue = one wave64 instruction per 2 cycles per SIMD-32, 4 SIMDs x CUs, at the 2.4 GHz maximum clock
    valu_insts, valu_src = None, None
    if fused_wl:
        for name in ("r05_pmc_range_kernels.json", "r02_pmc_range_kernels.json"):       # (both kernel forms: 493.1 M / 493.9 M)
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    kk = json.load(fh)["kernels"]
                valu_insts = next(v["counters"]["SQ_INSTS_VALU"] for k2, v in kk.items() if "range_fused" in k2 and "SQ_INSTS_VALU" in v.get("counters", {}))
                valu_src = f"SQ_INSTS_VALU per launch replayed from profiles/{name} (separate rocprofv3 --pmc pass)"
                break
            except (OSError, StopIteration, KeyError, ValueError):
                continue
    valu_peak = 256 * 4 * 2.4e9 / 2.0

    def roofline_block(ms, how):
        """achieved = algorithmic bytes per launch / launch duration.  `bound` says what the counters say limits the kernel."""
        ach = 16.0 * n * n / (ms * 1e-3) / 1e9
        blk = {"bound": "hbm", "kernel": kernel_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
               "traffic": traffic, "traffic_source": traffic_src, "launch_ms": ms, "timed_with": how,
               "algorithmic_bytes_per_launch": 16.0 * n * n}
        if fused_wl:
            # profiles/r02_pmc_range_kernels.json: a wave issues vector work 43 % of its lifetime, 28 % stalled at issue, 24 % at waits /
            # barriers, two waves per SIMD (233 VGPRs), 1.001 x the algorithmic bytes: the launch is limited by dependent vector issue
            # at that occupancy, not by HBM - `frac` stays the HBM fraction the metric asks for, valu_issue is the fraction of its own bound
            blk["bound"] = "valu-issue"
            blk["bound_note"] = ("instruction issue at two waves per SIMD (233 VGPRs, one 136 KiB-LDS workgroup per CU), not HBM: 1.001 x algorithmic "
                                 "bytes moved at about half the bandwidth; achieved / peak / frac are the HBM figures BASELINE's metric asks for")
            if valu_insts:
                rate = valu_insts / (ms * 1e-3)
                blk["valu_issue"] = {"wave_instructions_per_launch": valu_insts, "achieved_per_s": rate, "peak_per_s": valu_peak,
                                     "frac": rate / valu_peak, "source": valu_src,
                                     "peak_note": "256 CUs x 4 SIMD-32 x one wave64 instruction per 2 cycles at 2.4 GHz"}
        return blk

    line = None
    if rank == 0:
        line = {
            "metric": "focused SAR frames/sec (CSA focus, complex64)", "value": world * K / dt, "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "c64 (phase arguments f64)", "data": "synthetic",
            "config": {"workload": f"{n}x{n} complex64 single-channel CSA focus, echo resident in HBM",
                       "frames_per_step_per_gpu": 1,
                       # the host-transport fallback downloads every slot with a blocking copy that waits for every lane: one frame in flight in effect
                       "frames_in_flight_per_gpu": L if (use_rccl or d_recv is None) else 1, "range_launch_cus": range_cus if L > 1 else "all",
                       "lanes": lane_ids, "lane_probe_ratio": {str(k2): round(v2, 2) for k2, v2 in ctx._lane_ratios.items()},
                       "range_passes": "fused 2+3" if not a.unfused else "separate",
                       "image_layout": "[n_az x n_rg]; img.T returned as a view like the reference",
                       "parallelism": f"frames sharded 1/GPU x{world}" +
                                      (f"; 16x16 multilook slot fused into the focus + {collective} per step" if collective else "")},
            "roofline": roofline_block(rg_span_ms if rg_span_ms else rg_ms / launches, "in-kernel clock stamps" if rg_span_ms else "HIP events"),
        }
        moved = (5 if (n > 128 and not a.unfused) else 6 if n > 128 else 3 if not a.unfused else 4) * 16.0 * n * n   # HBM round trips of the image per frame
        line["frame_bandwidth"] = {
            "algorithmic_bytes_per_frame": 64.0 * n * n, "moved_bytes_per_frame": moved,
            "achieved_GBps_algorithmic": 64.0 * n * n * world / (dt / K) / 1e9 / world, "achieved_GBps_moved": moved / (dt / K) / 1e9,
            "frac_of_peak_algorithmic": 64.0 * n * n / (dt / K) / 1e9 / HBM_PEAK_GBS, "frac_of_peak_moved": moved / (dt / K) / 1e9 / HBM_PEAK_GBS,
            "note": "whole frame per GPU over the timed region: SURVEY.md 8(d)'s four passes x 16 B/sample (algorithmic) and the image's actual HBM "
                    "round trips (two launches per two-step azimuth transform + the fused range launch)"}
        line["env_switches"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith("SARX_")}      # kernel-form switches in effect
        line["roofline"]["measured_in"] = (
            f"the HEADLINE region of this run ({L} frame(s) in flight, range grid sized for {range_cus if L > 1 else 'all'} CUs): " +
            ("execution span of every timed step's launch from in-kernel clock stamps (first workgroup's start to last workgroup's end, "
             "s_memrealtime), i.e. without the time the launch queues behind the other lane's kernels; rocprofv3 --kernel-trace --stats of the "
             "same command: profiles/r05_*_bench_kernel_stats.csv" if rg_span_ms else
             "HIP events recorded on the launch's own stream around every timed step's range launch(es)"))
        line["roofline"]["launch_ms_between_events"] = rg_ms / launches
        if solo is not None:
            # the same kernel with the GPU to itself: the one_frame_in_flight leg (lane 0 only, grid sized for every CU), HIP events around
            # every launch - the kernel's own figure, kept beside the headline configuration's
            alone_ms = solo["range_launch_ms"]
            line["roofline_kernel_alone"] = roofline_block(alone_ms, "HIP events")
            if fused_wl and traffic_alone:          # alone on the chip the launcher takes the touch-prefetch form of the kernel: its own counters
                line["roofline_kernel_alone"].update(kernel="range_fused_wl_touch_kernel (the same launch with the next-line touch prefetch: chip to itself, in place)",
                                                     traffic=traffic_alone, traffic_source=traffic_alone_src)
            line["roofline_kernel_alone"]["measured_in"] = (
                f"the one_frame_in_flight leg of this run (lane 0 only, {solo['steps']} steps between barriers, grid sized for all CUs): HIP events "
                "around every range launch, the kernel alone on the GPU; rocprofv3 summary of the same: profiles/r05_*_bench_inflight1_kernel_stats.csv "
                "(bench.py --in-flight 1)" + (f"; in-kernel stamps of the same launches: {solo['range_span_ms']:.4f} ms" if solo.get("range_span_ms") else ""))
            line["one_frame_in_flight"] = {"ms_per_step": solo["ms_per_step"], "value": solo["value"], "unit": "frames/s", "steps": solo["steps"],
                                           "note": "the same steps on lane 0 only, timed after the headline region: one frame's latency"}
        if collective:
            line["collective"] = {"transport": collective, "ranks": world, "rccl": rccl, "rccl_ranks": world if use_rccl else 0,
                                  "rank_devices": ranks_devices,
                                  "stack": {"name": f"multilook {LOOKS}x{LOOKS} of |img|^2 (one {slot_bytes / 2**20:.2f} MiB slot per frame, emitted by the focus)",
                                            "gather_bytes_per_rank_per_step": slot_bytes,
                                            "gather_s_per_step_at_one_xgmi_link": slot_bytes / (XGMI_LINK_GBS * 1e9),
                                            "link_bound_at_8_gpus": bool(slot_bytes / (XGMI_LINK_GBS * 1e9) > dt / K)}}
            # false = RCCL did not come up on every rank and the slots travelled through host memory (a rehearsal on fewer devices
            # than ranks, or a broken node): such a line is NOT an xGMI scaling measurement
            line["collective_ok"] = bool(use_rccl)
        if not a.unfused:
            # SURVEY.md 8(d) counts 16 B/sample per FFT(+phase) pass; this launch does two of them (FFT+Phi2, IFFT+Phi3)
            # on one HBM round trip.  `achieved` counts the bytes it really moves (16 B/sample), once.
            line["roofline"]["survey_passes_in_launch"] = 2
        if per_pass:
            # the standalone range FFT + Phi_2 launch BASELINE.json's 70 % target names, and the others
            perm = "rg_fft_phi2_permuted_spectrum_in_place" in per_pass
            p2 = per_pass["rg_fft_phi2_permuted_spectrum_in_place" if perm else "rg_fft_phi2"]
            line["roofline_rg_fft_phi2_pass"] = {
                "bound": "hbm", "kernel": "range_wp_kernel<FFT+Phi2> (spectrum stored in the permuted order its inverse reads)" if perm
                else "range pass FFT+Phi2", "achieved": p2["GBps_per_launch"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": p2["GBps_per_launch"] / HBM_PEAK_GBS, "launch_ms": p2["ms"], "launch_ms_rounds": p2["ms_rounds"],
                "traffic": traffic_p2 if perm else None, "traffic_source": traffic_p2_src if (perm and traffic_p2) else None,
                "algorithmic_bytes_per_launch": 16.0 * n * n,
                "note": "the fused range-FFT + chirp-scaling-phase launch north_star's 70 % target names, as the unfused focus runs it: in place, "
                        "alternating with its inverse, HIP events around every launch on the ctx stream, three rounds of 10 launches, median "
                        "round, outside the timed region; the default path runs it fused with pass 3 in one launch"}
            line["passes"] = per_pass

    # the config-4 buffers make room for config 5
    for x in plans:
        x.close()
    for x in d_ins + d_imgs:
        x.release()
    if d_recv is not None:
        d_recv.release()
    d_stamp.release()

    # ---- the other GPU legs, back to back, before any CPU leg (the driver's GPU-busy sampler then sees one busy stretch) ----------
    if not a.no_batch:
        stacks = {"both": ("multilook", "magnitude"), "all": ("multilook", "magnitude", "products"),
                  "default": ("multilook", "products")}.get(a.stack, (a.stack,))
        for st in stacks:
            blk = run_batch64(sarx, ctx, a, world, rank, dist, use_rccl, host_comm, barrier, st)
            if rank == 0:
                blk["collective"] = collective or "none (one rank)"
                line["batch64" if st == "multilook" else "batch64_" + st] = blk
    if rank == 0 and world == 1 and not a.no_configs:
        if n != 4096:
            line["config2_4096"] = run_single_channel(sarx, ctx, 4096, steps=200)         # BASELINE config 2
        line["config3_two_channel"] = run_config3(sarx, ctx, a.batch_size, frames=20, cpu=False)   # BASELINE config 3

    # ---- CPU legs (rank 0 only; never inside a timed region) ---------------------------------------------------------------------
    if rank == 0:
        if not a.no_cpu:
            if world == 1:
                mt = min(os.cpu_count() or 1, 32)
                frame = (host_frame[0], radar.focus_args(n)) if host_frame is not None else None
                if mt > 1:                          # the all-core figure on the real frame next to the reference-style single thread
                    blk, img = cpu_baseline(n, workers=mt, scaled=False, frame=frame, keep_image=True)
                    line["cpu_baseline_threads"] = blk
                    if host_frame is not None:
                        line["parity_threads"] = parity_figures(host_frame[1], img)
                    del img
                # one thread (how the reference's NumPy runs) on the REAL frame, row-blocked lean path: ~30 s at 16384^2
                blk, img = cpu_baseline(n, workers=1, scaled=a.cpu_sample, frame=None if a.cpu_sample else frame, keep_image=True)
                line["cpu_baseline"] = blk
                if host_frame is not None:
                    # north_star's bar (1): the timed frame itself, GPU image against the oracle's complex128 focus of the same echo
                    line["parity"] = parity_figures(host_frame[1], img)
                    line["parity"]["what"] = (f"lane 0's {n}x{n} frame of the timed region: GPU image (complex64, downloaded once) against "
                                              "oracle/csa_oracle.sar_focus_csa_lean of the same downloaded echo (complex128, the cpu_baseline leg's own "
                                              "output), relative L2 over all samples")
                    assert line["parity"]["rel_l2_mag"] <= 1e-4 and line["parity"]["rel_l2_complex"] <= 1e-4, line["parity"]
                else:
                    line["parity"] = None
                del img
                if a.config3_cpu and "config3_two_channel" in line:
                    line["config3_two_channel"]["cpu_baseline"] = cpu_baseline_two_channel(a.batch_size, 1, min(a.batch_size, 4096))
                    if mt > 1:
                        line["config3_two_channel"]["cpu_baseline_threads"] = cpu_baseline_two_channel(a.batch_size, mt, a.batch_size)
            else:
                # N > 1 lines carry it too (north_star: every N next to the NumPy CPU path): rank 0's host cores, one thread, the bounded
                # sample, timed AFTER every GPU leg while the other ranks wait at the final barrier - it is not part of any timed region
                line["cpu_baseline"] = cpu_baseline(n, workers=1, scaled=True)
                line["cpu_baseline"]["sample"] += f"; timed on rank 0 of {world} after the GPU legs, the other ranks idle at the barrier"
        line["leg_order"] = "GPU legs back to back first (headline, one frame in flight, passes, batch64 blocks, configs 2 and 3), CPU legs last"
        emit(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
