"""VideoSAR / multi-aperture batch: independent frames sharded over the GPUs of
one node, one process per GPU, image stack reassembled with an all-gather.

Mirrors the frame loop of the reference's batch script
(sar_batch_sim.py:303-331: frame f = pulses [f*STEP, f*STEP+CPI), each frame
focused independently, outputs stacked and normalised by the global max,
:337).  Frames never exchange data while focusing, so the only collective is
the gather of the (multilooked) image stack.

Two transports implement the same two-method interface:
  RcclStackComm   - sarx_allgather_dev (RCCL over xGMI), device buffers, comm stream
  TorchStackComm  - torch.distributed (gloo) on host arrays; used by the CPU tests
"""
from __future__ import annotations

import numpy as np


def shard_frames(n_frames, world_size, rank):
    """Frame f lives on rank f mod world_size (round-robin keeps per-rank counts within 1)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, n_frames, world_size))


def rounds(n_frames, world_size):
    """Number of gather rounds: ceil(n_frames / world_size)."""
    return -(-n_frames // world_size)


def stack_from_rounds(round_blocks, n_frames):
    """round_blocks[i] is the [world, H, W] result of round i's all-gather (slot r = rank r's
    frame i*world + r).  Returns the [n_frames, H, W] stack in frame order, dropping pad slots."""
    stack = np.concatenate([np.asarray(b) for b in round_blocks], axis=0)
    return stack[:n_frames]


def normalise_stack(stack):
    """Global-max display normalisation of the frame stack (sar_batch_sim.py:337-338)."""
    g_max = float(np.max(np.abs(stack))) if stack.size else 0.0
    g_max = g_max if g_max > 0 else 1.0
    return stack / g_max, g_max


class LocalStackComm:
    """world_size 1: the gather of one rank is its own slot."""
    world, rank = 1, 0

    def all_gather(self, slot):
        return np.asarray(slot)[None]

    def finish(self):
        pass


class TorchStackComm:
    """Host all-gather over an initialised torch.distributed group (gloo on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_gather(self, slot):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(slot))
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return np.stack([o.numpy() for o in out], axis=0)

    def finish(self):
        pass


class RcclStackComm:
    """Device all-gather through libsarx (RCCL).  The unique id is created on rank 0 and
    handed to the other ranks by ``bootstrap`` (any broadcast of 128 bytes, e.g. a gloo group)."""

    def __init__(self, ctx, world, rank, bootstrap):
        uid = ctx.comm_unique_id() if rank == 0 else None
        uid = bootstrap(uid)
        ctx.comm_init(uid, world, rank)
        self.ctx, self.world, self.rank = ctx, world, rank

    def all_gather_dev(self, d_slot, d_recv_block, nbytes):
        """Asynchronous on the comm stream, ordered after the compute stream's work so far."""
        self.ctx.allgather(d_slot, d_recv_block, nbytes)

    def finish(self):
        self.ctx.comm_sync()


def run_batch_host(frame_ids_all, world, rank, process_frame, comm):
    """Reference-shaped driver on host arrays (CPU tests, small jobs).

    process_frame(f) -> 2-D float array (the frame's stack slot).  Every rank
    takes part in every round; ranks without a frame in the last round send zeros.
    """
    n = len(frame_ids_all)
    mine = shard_frames(n, world, rank)
    blocks, shape = [], None
    for i in range(rounds(n, world)):
        if i < len(mine):
            slot = np.asarray(process_frame(frame_ids_all[mine[i]]), dtype=np.float32)
            shape = slot.shape
        else:
            if shape is None:
                raise ValueError("a rank without any frame cannot size its pad slot")
            slot = np.zeros(shape, dtype=np.float32)
        blocks.append(comm.all_gather(slot))
    comm.finish()
    return stack_from_rounds(blocks, n)
