#!/usr/bin/env python3
"""BASELINE config 5: a 64-frame VideoSAR batch of 8192 x 8192 two-channel scenes sharded over the GPUs of one node
(frame f -> rank f mod N), RCCL all-gather of the image stack.  Strong scaling: the batch is fixed, ranks split it.

    python3 tools/bench_batch64.py [--frames 64] [--size 8192]                                  # one GPU
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
            --master-port P tools/bench_batch64.py                                               # N GPUs

Per frame: CSA focus of both channels, ATI/DPCA products with the phase-balance sum, the 5 % magnitude mask, a
16 x 16 multilook of channel 1 into the frame's stack slot; per round of N frames one all-gather of the slots on the
communication stream, overlapped with the next round's focusing.  Echoes are device-resident noise (seed 1000 + f).
Prints one JSON line on rank 0: frames/s for the whole batch, max over ranks, barrier + sync on both sides.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))
LOOKS = 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--size", type=int, default=8192)
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
    from sarx.batch import RcclStackComm, rounds, shard_frames

    n_dev = max(1, sarx.device_count())
    ctx = sarx.Context(local_rank % n_dev)              # one rank per GPU; wraps only when ranks outnumber GPUs (tests)
    n, px = a.size, a.size * a.size
    plan = sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE)
    raw1, raw2, s1, s2 = (ctx.alloc(px * 8) for _ in range(4))
    outs = {k: ctx.alloc(px * 4) for k in ("ati_phase", "slc1_mag", "dpca_mag")}
    masked = ctx.alloc(px * 4)
    slot_bytes = (n // LOOKS) * (n // LOOKS) * 4
    comm = None
    if world > 1 or os.environ.get("SARX_BENCH_FORCE_COMM") == "1":
        def bootstrap(uid):
            if dist is None:
                return uid
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        comm = RcclStackComm(ctx, world, rank, bootstrap)
    n_rounds = rounds(a.frames, world)
    d_slot = ctx.alloc(slot_bytes * 2)
    d_stack = ctx.alloc(slot_bytes * world * n_rounds)          # the whole gathered stack, round-major
    mine = shard_frames(a.frames, world, rank)

    def run_batch():
        for i in range(n_rounds):
            if i < len(mine):
                f = mine[i]
                ctx.fill_noise(raw1, px, 1000 + 2 * f)           # stands in for the frame's two echo channels
                ctx.fill_noise(raw2, px, 1001 + 2 * f)
                plan.focus_dev(raw1, s1)
                plan.focus_dev(raw2, s2)
                mx, _ = ctx.ati_dpca(s1, s2, px, 0.0, outs)
                ctx.mask_phase(outs["ati_phase"], outs["slc1_mag"], px, 0.05 * mx, masked)
            if comm is not None:
                ctx.comm_fence_compute()
            ctx.lib.sarx_multilook_dev(ctx.h, s1.ptr, d_slot.ptr + (i & 1) * slot_bytes, n, n, LOOKS)
            if comm is not None:
                ctx.lib.sarx_allgather_dev(ctx.h, d_slot.ptr + (i & 1) * slot_bytes, d_stack.ptr + i * slot_bytes * world, slot_bytes)
        if comm is not None:
            comm.finish()

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    run_batch()                                                  # warm-up batch
    barrier()
    t0 = time.perf_counter()
    run_batch()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "VideoSAR batch frames/sec (two-channel CSA focus + ATI/DPCA, stack all-gather)",
                          "value": a.frames / dt, "unit": "frames/s", "n_gpus": world, "batch_s": dt, "scaling": "strong",
                          "config": {"workload": f"{a.frames} frames x two-channel {n}x{n} complex64, frame f -> rank f mod N",
                                     "stack": f"{LOOKS}x{LOOKS} multilook of channel 1, one RCCL all-gather per round"}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
