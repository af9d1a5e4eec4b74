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

    def all_reduce_max(self, value):
        return float(value)

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

    def all_reduce_max(self, value):
        """max over ranks of one float: the collective of the global normalisation (sar_batch_sim.py:337)."""
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def finish(self):
        pass


def agree_on_rccl(ctx, world, rank, dist=None, log=None):
    """Collective RCCL bootstrap that cannot strand a rank: every rank reaches every rendezvous step whatever failed
    where.  Rank 0 creates the unique id (None after a failure), the id is broadcast over the torch.distributed
    (gloo) group, every rank tries ncclCommInitRank, and a MIN-reduced flag decides for all ranks alike.
    Returns True when the communicator is up on every rank; otherwise False on every rank (any half-made communicator
    is destroyed) and the caller falls back or raises - on all ranks together."""
    uid, ok = None, 1
    if rank == 0:
        try:
            uid = ctx.comm_unique_id()
        except Exception as exc:                          # noqa: BLE001
            if log:
                log(f"RCCL unique id failed: {exc}")
    if dist is not None and world > 1:
        box = [uid]
        dist.broadcast_object_list(box, src=0)            # every rank takes part, also after a failure on rank 0
        uid = box[0]
    made = False
    if uid is None:
        ok = 0
    else:
        try:
            ctx.comm_init(uid, world, rank)
            made = True
        except Exception as exc:                          # noqa: BLE001
            ok = 0
            if log:
                log(f"RCCL init failed on rank {rank}: {exc}")
    if dist is not None and world > 1:
        import torch
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok and made:
        ctx.lib.sarx_comm_destroy(ctx.h)
    return bool(ok)


class RcclStackComm:
    """Device all-gather through libsarx (RCCL).  ``dist``: an initialised torch.distributed module (gloo group) used
    only to hand the 128-byte unique id around and to agree on success; with world == 1 it may be None.  If the
    bootstrap fails anywhere, every rank raises SarxError together (nobody is left waiting in a rendezvous)."""

    def __init__(self, ctx, world, rank, dist=None, log=None):
        from ._ffi import SarxError
        if not agree_on_rccl(ctx, world, rank, dist, log):
            raise SarxError(-5, "RCCL bootstrap failed on at least one rank (all ranks raise together)")
        self.ctx, self.world, self.rank = ctx, world, rank

    def all_gather_dev(self, d_slot, d_recv_block, nbytes):
        """Asynchronous on the comm stream, ordered after the compute stream's work so far."""
        self.ctx.allgather(d_slot, d_recv_block, nbytes)

    def finish(self):
        self.ctx.comm_sync()

    def close(self):
        self.ctx.lib.sarx_comm_destroy(self.ctx.h)


def global_max_host(local_frames, comm):
    """g_max = max([np.max(np.abs(fr)) for fr in frames]) over ALL ranks' frames (sar_batch_sim.py:337-338, 0 -> 1.0): every
    rank reduces the frames it focused itself, one float crosses the transport - nobody needs the gathered stack for it."""
    local = max((float(np.max(np.abs(fr))) for fr in local_frames if np.size(fr)), default=0.0)
    g_max = comm.all_reduce_max(local)
    return g_max if g_max > 0 else 1.0


def run_batch_host(frame_ids_all, world, rank, process_frame, comm, slot_shape=None):
    """Reference-shaped driver on host arrays (CPU tests, small jobs).

    process_frame(f) -> 2-D float array (the frame's stack slot).  Every rank
    takes part in every round; ranks without a frame in a round send zeros.  A rank that owns no frame at all
    (more ranks than frames) sizes its pad slots from ``slot_shape``; without it the shape is agreed over the
    transport first (one extra tiny gather), so no rank ever raises while its peers wait in a collective.
    """
    n = len(frame_ids_all)
    mine = shard_frames(n, world, rank)
    blocks, shape = [], tuple(slot_shape) if slot_shape is not None else None
    first = None
    if shape is None:
        if mine:
            first = np.asarray(process_frame(frame_ids_all[mine[0]]), dtype=np.float32)
            shape = first.shape
        if world > n:                                     # somebody owns nothing: agree on the shape
            mine_shape = np.array(shape if shape is not None else (0, 0), dtype=np.float32)[None, :]
            shapes = comm.all_gather(mine_shape)
            shape = tuple(int(v) for v in shapes.reshape(world, 2).max(axis=0))
    for i in range(rounds(n, world)):
        if i < len(mine):
            slot = first if (i == 0 and first is not None) else np.asarray(process_frame(frame_ids_all[mine[i]]), dtype=np.float32)
        else:
            slot = np.zeros(shape, dtype=np.float32)
        blocks.append(comm.all_gather(slot))
    comm.finish()
    return stack_from_rounds(blocks, n)


# ----------------------------------------------------------------------------------------------------------------
# BASELINE config 5 on the device: a VideoSAR batch of two-channel scenes, frame f -> rank f mod N
# ----------------------------------------------------------------------------------------------------------------
STACKS = ("multilook", "magnitude", "products")


class _Ptr:
    """A bare device address where the engine wrappers expect a buffer object."""
    def __init__(self, ptr):
        self.ptr = ptr


class TwoChannelBatch:
    """n_frames two-channel [n x n] complex64 scenes (the processing section of sar_ati_dcpa_sim_csa.py:402-449 per
    frame, frames independent as in sar_batch_sim.py:303-331), sharded frame f -> rank f mod world.

    Per frame, all on the device and without a host round trip: CSA focus of both channels, ATI/DPCA products with
    the fp64 phase-balance sum, the 5 % magnitude mask (threshold from the device-side max), and the frame's stack
    slot:  stack="multilook": looks x looks mean of |slc1|^2 (1 MiB per 8192^2 frame - the display stack the
    reference's batch script keeps, 512^2 per frame, sar_batch_sim.py:322), emitted by channel 1's last azimuth launch
    as row-wise partial sums and finished by a small launch (the image is not read again);  stack="magnitude": |slc1| at full
    resolution (256 MiB per 8192^2 frame: the configuration that loads xGMI);  stack="products": the frame's three GMTI
    planes [masked ATI phase, |slc1|, DPCA magnitude] (:414-419,447-449; 768 MiB per 8192^2 frame - SURVEY.md 8(e)'s
    [frames x 3 x n x n] product stack), written by channel 2's last azimuth launch straight at their place in the
    stack buffer, so the per-frame products leave the GPU they were computed on instead of being overwritten.

    The stack lives in one device buffer [rounds][world][slot]; a frame's slot is produced directly at its place
    and each round is gathered IN PLACE (send = recv + rank * slot), so there is no send buffer to recycle and no
    fence between compute and communication beyond stream order.  A rank without a frame in the last round zeroes its
    slot before the gather (never a stale slot).  Transports: RCCL on the ctx's comm stream (overlaps the next
    round's focusing), a torch.distributed (gloo) group through host memory (fallback / one-GPU rehearsal), or none
    (world 1).
    """

    def __init__(self, ctx, n, n_frames, world=1, rank=0, stack="multilook", looks=16, rccl=False, host_comm=None,
                 seed_base=1000, flags=None, mask_frac=0.05, resident=True, fused_mask=True, fused_ati=True, scene="noise",
                 scene_scale=1.0, lanes=None):
        from . import _ffi, radar
        from .engine import CsaPlan
        if stack not in STACKS:
            raise ValueError(f"stack must be one of {STACKS}")
        if world > 1 and not rccl and host_comm is None:
            raise ValueError("world > 1 needs a transport (rccl=True or host_comm)")
        self.ctx, self.n, self.n_frames, self.world, self.rank = ctx, int(n), int(n_frames), int(world), int(rank)
        self.stack_kind, self.looks, self.rccl, self.host_comm = stack, int(looks), bool(rccl), host_comm
        self.seed_base, self.mask_frac = int(seed_base), float(mask_frac)
        px = self.n * self.n
        self.px = px
        # scene="noise": device-resident complex noise per (frame, channel) - bandwidth does not depend on content.
        # scene="c3": SURVEY.md 8(d) C5's content - frame f is the C3 scene (5 x 5 grid + a 15 m/s radial mover + a slow mover,
        # sarx.radar.c3_scene) with the movers advanced by f * 0.1 s, both receive channels synthesised on the device by the
        # bistatic echo kernel over n + 1 pulses, the DPCA pulse shift (sar_ati_dcpa_sim_csa.py:402-403) as two views.
        if scene not in ("noise", "c3"):
            raise ValueError("scene must be 'noise' or 'c3'")
        self.scene, self.scene_scale = scene, float(scene_scale)
        self.k = radar.scaled_constants(n, n) if scene == "c3" else None
        args = radar.focus_args(n) if scene == "noise" else (self.k["Lambda"], self.k["T_p"], self.k["Kr"], self.k["FS"], self.k["PRF"],
                                                             self.k["V_eff"], self.k["R0"], self.k["t_start_fast"])
        self.focus_args = args
        # Frames in flight: this rank's i-th frame runs on compute lane i % lanes of the context (sarx_select_lane), every lane
        # with its own plan (scratch), image buffers and product planes, so the launches of consecutive frames overlap on the GPU.
        # Results do not depend on it (same kernels, separate buffers).
        if lanes is None:                   # frames of 4096^2 and more gain 3-8 %; small frames are launch-bound and lose to the lane switches
            lanes = 2 if n >= 4096 else 1
        if world > 1 and not rccl:          # host transport: every round ends in a blocking download that waits for every lane - one frame in flight
            lanes = 1
        self.lanes = max(1, min(int(lanes), 4))
        self.range_cus = 192                # of 256: sarx_set_range_cus while frames are in flight (only the persistent 16384-sample launch looks at it)
        self._lane_state = []
        for _ in range(self.lanes):
            st = {"plan": CsaPlan(ctx, n, n, *args, flags=_ffi.FUSE_RANGE if flags is None else flags),
                  "s1": ctx.alloc(px * 8), "s2": ctx.alloc(px * 8),
                  "outs": {k: ctx.alloc(px * 4) for k in ("ati_phase", "slc1_mag", "dpca_mag")},
                  "masked": ctx.alloc(px * 4), "d_max": ctx.alloc(_ffi.MAX_SLOT_BYTES)}
            self._lane_state.append(st)
        self.lane_ids = ctx.concurrent_lanes(self.lanes)       # lanes that really run side by side (probed once per context)
        self._cur = 0
        # the 5 % mask inside the ATI launch: channel 1's focus leaves max|slc1| in d_max while it writes the image
        # (sarx_csa_plan_set_max_slot), so no further pass over the phase and magnitude planes is needed
        self.fused_mask = bool(fused_mask)
        # ... and the ATI / DPCA products out of channel 2's last azimuth launch (sarx_csa_plan_set_ati): slc2 is never written,
        # neither image is read again; sizes without that epilogue keep the separate launch
        self.fused_ati = bool(fused_ati) and self.fused_mask and n % 64 == 0
        self.slot_shape = (n // looks, n // looks) if stack == "multilook" else (n, n) if stack == "magnitude" else (3, n, n)
        self.slot_bytes = int(np.prod(self.slot_shape)) * 4
        self.n_rounds = rounds(self.n_frames, self.world)
        self.mine = shard_frames(self.n_frames, self.world, self.rank)
        self.d_stack = ctx.alloc(self.slot_bytes * self.world * self.n_rounds)
        # echoes: resident = every frame of this rank has its own two channel buffers, synthesised once by prepare()
        # (BASELINE config 5: inputs in HBM before the clock starts; 1 GiB per 8192^2 two-channel frame, 64 GiB for the
        # whole batch on one GPU of 288 GB); otherwise one pair of buffers refilled inside run() frame by frame
        self.resident = bool(resident)
        n_buf = len(self.mine) if self.resident else self.lanes          # not resident: one refilled pair per lane
        rows = self.n + (1 if scene == "c3" else 0)                    # c3: one more pulse, consumed by the DPCA pulse shift
        self._alloc = [(ctx.alloc(rows * self.n * 8), ctx.alloc(rows * self.n * 8)) for _ in range(max(n_buf, 1))]
        # what the focuser reads: the buffers themselves, or (c3) rx1[1:], rx2[:-1] as views
        self.raw = [(_Ptr(a.ptr + (self.n * 8 if scene == "c3" else 0)), _Ptr(b.ptr)) for a, b in self._alloc]
        self._prepared = False

    # the current lane's plan and buffers
    plan = property(lambda self: self._lane_state[self._cur]["plan"])
    s1 = property(lambda self: self._lane_state[self._cur]["s1"])
    s2 = property(lambda self: self._lane_state[self._cur]["s2"])
    outs = property(lambda self: self._lane_state[self._cur]["outs"])
    masked = property(lambda self: self._lane_state[self._cur]["masked"])
    d_max = property(lambda self: self._lane_state[self._cur]["d_max"])

    def use_lane(self, lane):
        """Later focus_frame calls run on compute lane `lane` with that lane's plan and buffers."""
        self._cur = int(lane) % self.lanes
        self.ctx.select_lane(self.lane_ids[self._cur])

    # -- one frame -------------------------------------------------------------------------------------------
    def synth_frame(self, f, slot):
        """Frame f's two echo channels into buffer pair `slot`: device-resident complex noise with seeds per (frame, channel),
        or the C3 scene with its movers advanced to frame f (scene="c3")."""
        if self.scene == "noise":
            self.ctx.fill_noise(self.raw[slot][0], self.px, self.seed_base + 2 * f)
            self.ctx.fill_noise(self.raw[slot][1], self.px, self.seed_base + 2 * f + 1)
            return
        from . import radar
        from .echo import synth_device
        k, n = self.k, self.n
        n_pulses = n + 1
        t_int = n_pulses / k["PRF"]
        t_vec = np.linspace(-t_int / 2, t_int / 2, n_pulses)
        p_tx, v_tx = radar.orbit_track(t_vec, k)
        v_dir = v_tx / np.linalg.norm(v_tx, axis=1, keepdims=True)
        t_start = 2 * k["R0"] / k["C"] - n / k["FS"] / 2                       # the focuser's window (radar.scaled_constants)
        t_fast = t_start + np.linspace(0, n / k["FS"], n)                      # the generators' linspace grid (:113), n samples
        for ch, off in enumerate((-k["d_rx"] / 2, k["d_rx"] / 2)):
            first = True
            for targets, vel in radar.c3_scene(f):
                p0 = np.array([[t["position"][0] * self.scene_scale, t["position"][1] * self.scene_scale, t["position"][2]] for t in targets])
                amp = np.sqrt(np.array([t["rcs"] for t in targets], dtype=np.float64))
                synth_device(self.ctx, 1, p0, np.asarray(vel, dtype=np.float64), t_vec, p_tx, p_tx + v_dir * off, amp, t_fast,
                             k["Kr"], k["T_p"], k["C"], k["FC"], out=self._alloc[slot][ch], accumulate=not first)
                first = False

    def prepare(self):
        """Synthesise this rank's echoes (resident mode): call before the clock starts."""
        if self.resident and not self._prepared:
            for i, f in enumerate(self.mine):
                self.synth_frame(f, i)
            self.ctx.sync()
        self._prepared = True

    def focus_frame(self, bufs, slot_ptr=None):
        """raw1, raw2 -> s1, s2, ATI/DPCA planes, masked phase (and, for the multilook stack, channel 1's slot at slot_ptr,
        emitted by the focus itself: sarx_csa_plan_set_look_slot).  Only enqueues (no host synchronisation)."""
        ctx = self.ctx
        fused = slot_ptr is not None and self.stack_kind == "multilook"
        if fused:
            self.plan.set_look_slot(self.looks, slot_ptr)
        if self.fused_mask:
            self.plan.set_max_slot(self.d_max)
        self.plan.focus_dev(bufs[0], self.s1)
        if fused:
            self.plan.set_look_slot(self.looks, None)
        if self.fused_mask:
            self.plan.set_max_slot(None)
        if self.fused_ati:
            # the full-resolution magnitude slot IS the |slc1| plane of the products: written at its place in the stack buffer
            self._slot_written = slot_ptr is not None and self.stack_kind in ("magnitude", "products")
            masked, mag, dm = self.masked, self.outs["slc1_mag"], self.outs["dpca_mag"]
            if self._slot_written and self.stack_kind == "magnitude":
                mag = _Ptr(slot_ptr)
            elif self._slot_written:                       # the three planes of the frame, in place in the stack
                masked, mag, dm = (_Ptr(slot_ptr + i * self.px * 4) for i in range(3))
            self.plan.set_ati(self.s1, self.d_max, self.mask_frac, 0.0, masked, mag, dm)
            self.plan.focus_dev(bufs[1], self.s2)               # s2 serves as scratch only
            self.plan.set_ati(None)
            return
        self.plan.focus_dev(bufs[1], self.s2)
        self._slot_written = False
        outs, masked = self.outs, self.masked
        if slot_ptr is not None and self.stack_kind == "products":     # separate launches, same destination
            masked = _Ptr(slot_ptr)
            outs = dict(self.outs, slc1_mag=_Ptr(slot_ptr + self.px * 4), dpca_mag=_Ptr(slot_ptr + 2 * self.px * 4))
            self._slot_written = True
        if self.fused_mask:
            outs = dict(outs, ati_phase=masked)                    # the phase plane comes out masked; no unmasked copy is kept
            ctx.ati_dpca_masked(self.s1, self.s2, self.px, 0.0, self.d_max, self.mask_frac, outs)
        else:
            ctx.ati_dpca(self.s1, self.s2, self.px, 0.0, outs, want_stats=False)
            ctx.mask_phase_frac(outs["ati_phase"], outs["slc1_mag"], self.px, self.mask_frac, masked)

    def _slot_ptr(self, i, r):
        return self.d_stack.ptr + (i * self.world + r) * self.slot_bytes

    def write_slot(self, dst_ptr):
        """The full-resolution magnitude slot (the multilook slot comes out of the focus itself)."""
        ctx = self.ctx
        if self.stack_kind == "magnitude" and not getattr(self, "_slot_written", False):
            from ._ffi import check
            check(ctx.lib.sarx_magnitude_dev(ctx.h, self.s1.ptr, dst_ptr, self.px), ctx.h)

    # -- the batch ---------------------------------------------------------------------------------------------
    def run(self):
        """Focus this rank's frames and assemble the whole stack on every rank.  Returns after everything is
        enqueued and (host transport only) gathered; call ctx.sync() to wait for the device."""
        from ._ffi import check
        ctx = self.ctx
        self.prepare()
        if self.lanes > 1:
            ctx.set_range_cus(self.range_cus)                       # persistent range launches leave CUs to the other lane's azimuth tiles
        try:
            self._run_rounds()
        finally:                                                    # the CU share and the lane are context state: never left behind by an exception
            self.use_lane(0)
            if self.lanes > 1:
                ctx.set_range_cus(0)
        if self.lanes > 1:
            ctx.lanes_join()                                        # whatever follows on any lane sees every frame finished
        if self.rccl and self.world > 1:
            ctx.comm_sync()

    def _run_rounds(self):
        from ._ffi import check
        ctx = self.ctx
        for i in range(self.n_rounds):
            self.use_lane(i)
            mine_ptr = self._slot_ptr(i, self.rank)
            if i < len(self.mine):
                bufs = self.raw[i] if self.resident else self.raw[i % self.lanes]
                if not self.resident:
                    self.synth_frame(self.mine[i], i % self.lanes)
                self.focus_frame(bufs, mine_ptr)
                self.write_slot(mine_ptr)
            else:                                                   # pad round: zeros, never a stale slot
                check(ctx.lib.sarx_memset(ctx.h, mine_ptr, 0, self.slot_bytes), ctx.h)
            if self.world == 1:
                continue
            if self.rccl:                                           # in place: send = recv + rank * slot
                check(ctx.lib.sarx_allgather_dev(ctx.h, mine_ptr, self._slot_ptr(i, 0), self.slot_bytes), ctx.h)
            else:
                slot = np.empty(self.slot_shape, dtype=np.float32)
                check(ctx.lib.sarx_memcpy_d2h(ctx.h, slot.ctypes.data, mine_ptr, self.slot_bytes), ctx.h)
                block = np.ascontiguousarray(self.host_comm.all_gather(slot))
                check(ctx.lib.sarx_memcpy_h2d(ctx.h, self._slot_ptr(i, 0), block.ctypes.data, block.nbytes), ctx.h)

    def global_max(self):
        """g_max of the whole stack (sar_batch_sim.py:337-338: max over all frames of max|frame|; 0 -> 1.0) without reading the
        gathered stack: every rank reduces the slots of its OWN frames on the device (sarx_max_abs_f32_dev), then one float is
        all-reduced with max - ncclAllReduce on the comm stream (sarx_allreduce_max_dev), the gloo group on the host-transport
        fallback, nothing at world 1.  Call after run(); synchronises (the value is returned to the host)."""
        from ._ffi import check
        ctx = self.ctx
        if not hasattr(self, "d_gmax"):
            self.d_gmax = ctx.alloc(4)
        check(ctx.lib.sarx_memset(ctx.h, self.d_gmax.ptr, 0, 4), ctx.h)
        for i in range(len(self.mine)):
            ctx.max_abs(_Ptr(self._slot_ptr(i, self.rank)), self.slot_bytes // 4, self.d_gmax)
        if self.rccl and self.world > 1:
            ctx.allreduce_max(self.d_gmax, 1)
            ctx.comm_sync()
        g = float(self.d_gmax.download(np.float32, (1,))[0])
        if self.world > 1 and not self.rccl:
            g = self.host_comm.all_reduce_max(g)
        return g if g > 0 else 1.0

    def stack(self, frames=None):
        """The assembled [n_frames, H, W] float32 stack (or the listed frames of it) on the host."""
        if frames is None:
            return self.d_stack.download(np.float32, (self.n_rounds * self.world, *self.slot_shape))[:self.n_frames]
        out = np.empty((len(frames), *self.slot_shape), dtype=np.float32)
        from ._ffi import check
        for k, f in enumerate(frames):
            check(self.ctx.lib.sarx_memcpy_d2h(self.ctx.h, out[k].ctypes.data, self.d_stack.ptr + f * self.slot_bytes,
                                               self.slot_bytes), self.ctx.h)
        return out

    def close(self):
        self.ctx.select_lane(0)
        for b in (self.d_stack, *(x for pair in self._alloc for x in pair)):
            b.release()
        for st in self._lane_state:
            for b in (st["s1"], st["s2"], st["masked"], st["d_max"], *st["outs"].values()):
                b.release()
            st["plan"].close()
        if hasattr(self, "d_gmax"):
            self.d_gmax.release()
