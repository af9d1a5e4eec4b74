"""BASELINE config 5 - the 64-frame VideoSAR batch of two-channel 8192 x 8192 scenes - through the shared device
driver (sarx.batch.TwoChannelBatch, what bench.py's `batch64` block times), on one GPU:
slot f of the stack is bit-identical to the multilook of an independently focused frame f, slots are in frame order,
pad slots are zeros, and a two-rank run (gloo transport, both ranks on the one GPU) assembles the same stack as one rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _independent_slot(sarx, ctx, n, f, stack, looks=16, seed_base=1000):
    """Frame f focused on its own with fresh buffers and a fresh plan: channel 1's stack slot."""
    from sarx import _ffi, radar
    px = n * n
    plan = sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE)
    raw, s1 = ctx.alloc(px * 8), ctx.alloc(px * 8)
    ctx.fill_noise(raw, px, seed_base + 2 * f)
    if stack == "multilook":                                  # the slot the focus itself emits (sarx_csa_plan_set_look_slot) ...
        d = ctx.alloc((n // looks) ** 2 * 4)
        plan.set_look_slot(looks, d.ptr)
        plan.focus_dev(raw, s1)
        out = d.download(np.float32, (n // looks, n // looks))
        d2 = ctx.alloc((n // looks) ** 2 * 4)                 # ... is the multilook of the finished image (other summation order)
        ctx.multilook(s1, d2, n, n, looks)
        ref = d2.download(np.float32, out.shape)
        assert np.abs(out - ref).max() <= 2e-6 * ref.max()
        d2.release()
    else:
        plan.focus_dev(raw, s1)
        d = ctx.alloc(px * 4)
        ctx.magnitude(s1, d, px)
        out = d.download(np.float32, (n, n))
    for b in (raw, s1, d):
        b.release()
    plan.close()
    return out


def test_config5_64_frames_two_channel_8192():
    import sarx
    from sarx.batch import TwoChannelBatch
    ctx = sarx.default_context()
    n, frames = 8192, 64
    b = TwoChannelBatch(ctx, n, frames, stack="multilook")
    b.run()
    ctx.sync()
    st = b.stack()
    assert st.shape == (frames, n // 16, n // 16) and np.isfinite(st).all() and st.min() > 0
    # the last frame's products are still in the buffers of the lane it ran on (frame i of this rank -> lane i % lanes):
    # two different channels went through ATI/DPCA
    assert b.lanes == 2
    b.use_lane(frames - 1)
    mx, sm = ctx.ati_stats()
    assert mx > 0 and abs(sm) > 0
    masked = b.masked.download(np.float32, (8, n))
    mag = b.outs["slc1_mag"].download(np.float32, (8, n))
    # the driver masks inside the ATI launch (threshold from the focus's fused maximum) and keeps no unmasked phase: take it
    # from a plain ATI launch over the last frame's two images, which are still in the driver's buffers
    assert b.fused_mask and b.fused_ati
    assert np.float32(mx) == b.d_max.download(np.float32, (256, 32))[:, 0].max()
    # ... and emits the products from channel 2's last azimuth launch, so slc2 itself was never written: focus it here
    b.plan.focus_dev(b.raw[-1][1], b.s2)
    plain = {k: ctx.alloc(n * n * 4) for k in ("ati_phase", "slc1_mag", "dpca_mag")}
    ctx.ati_dpca(b.s1, b.s2, n * n, 0.0, plain, want_stats=False)
    ph = plain["ati_phase"].download(np.float32, (8, n))
    np.testing.assert_array_equal(mag, plain["slc1_mag"].download(np.float32, (8, n)))
    np.testing.assert_array_equal(b.outs["dpca_mag"].download(np.float32, (8, n)), plain["dpca_mag"].download(np.float32, (8, n)))
    for v in plain.values():
        v.release()
    thr = np.float32(mx) * np.float32(0.05)
    np.testing.assert_array_equal(masked, np.where(mag > thr, ph, np.float32(0)))       # device-side threshold == host rule
    assert float(b.outs["dpca_mag"].download(np.float32, (8, n)).max()) > 0             # channels differ
    b.use_lane(0)
    for f in (0, 1, 17, 40, 63):
        np.testing.assert_array_equal(st[f], _independent_slot(sarx, ctx, n, f, "multilook"))
    # frame order: every slot is its own frame (noise frames have distinct multilooked images)
    sig = st.reshape(frames, -1)[:, :64]
    assert len({s.tobytes() for s in sig}) == frames
    b.run()                                                      # a second batch reproduces the stack bit for bit
    ctx.sync()
    np.testing.assert_array_equal(b.stack(frames=[5, 63]), st[[5, 63]])
    b.close()


def test_magnitude_stack_slot_is_full_resolution_channel_1():
    import sarx
    from sarx.batch import TwoChannelBatch
    ctx = sarx.default_context()
    n, frames = 2048, 5
    b = TwoChannelBatch(ctx, n, frames, stack="magnitude")
    b.run()
    ctx.sync()
    st = b.stack()
    assert st.shape == (frames, n, n)
    for f in (0, 4):
        np.testing.assert_array_equal(st[f], _independent_slot(sarx, ctx, n, f, "magnitude"))
    b.close()
    # echoes refilled frame by frame inside run() instead of resident per frame: the same stack
    b2 = TwoChannelBatch(ctx, n, frames, stack="magnitude", resident=False)
    b2.run()
    ctx.sync()
    np.testing.assert_array_equal(b2.stack(), st)
    b2.close()


def test_products_stack_holds_every_frames_three_planes():
    """stack="products" (the [frames x 3 x n x n] product stack): slot f = [masked ATI phase, |slc1|,
    DPCA magnitude] of frame f, written in place by channel 2's last azimuth launch; equal bit for bit to the planes of
    an independently focused frame f through separate ATI / mask launches, and to the form without the fused epilogue."""
    import sarx
    from sarx import _ffi, radar
    from sarx.batch import TwoChannelBatch
    ctx = sarx.default_context()
    n, frames = 2048, 4
    b = TwoChannelBatch(ctx, n, frames, stack="products")
    assert b.fused_ati and b.slot_shape == (3, n, n) and b.slot_bytes == 3 * n * n * 4
    b.run()
    ctx.sync()
    st = b.stack()
    assert st.shape == (frames, 3, n, n) and np.isfinite(st).all()
    b.close()
    px = n * n
    plan = sarx.CsaPlan(ctx, n, n, *radar.focus_args(n), flags=_ffi.FUSE_RANGE)
    raw, s1, s2 = ctx.alloc(px * 8), ctx.alloc(px * 8), ctx.alloc(px * 8)
    outs = {k: ctx.alloc(px * 4) for k in ("ati_phase", "slc1_mag", "dpca_mag")}
    masked = ctx.alloc(px * 4)
    for f in (0, 3):
        ctx.fill_noise(raw, px, 1000 + 2 * f)
        plan.focus_dev(raw, s1)
        ctx.fill_noise(raw, px, 1000 + 2 * f + 1)
        plan.focus_dev(raw, s2)
        mx, _ = ctx.ati_dpca(s1, s2, px, 0.0, outs)
        ctx.mask_phase(outs["ati_phase"], outs["slc1_mag"], px, np.float32(mx) * np.float32(0.05), masked)
        np.testing.assert_array_equal(st[f, 0], masked.download(np.float32, (n, n)))
        np.testing.assert_array_equal(st[f, 1], outs["slc1_mag"].download(np.float32, (n, n)))
        np.testing.assert_array_equal(st[f, 2], outs["dpca_mag"].download(np.float32, (n, n)))
        assert 0.005 < (st[f, 0] == 0).mean() < 0.3           # noise frames: a few per cent of the pixels lie under 5 % of the maximum
    for x in (raw, s1, s2, masked, *outs.values()):
        x.release()
    plan.close()
    b2 = TwoChannelBatch(ctx, n, frames, stack="products", fused_ati=False)      # separate ATI launch, same destination
    b2.run()
    ctx.sync()
    np.testing.assert_array_equal(b2.stack(), st)
    b2.close()


@pytest.mark.parametrize("n,scene_scale", [(2048, 0.25), (4096, 0.5)])
def test_c3_scene_frames_against_the_oracle(n, scene_scale):
    """scene="c3" (frame f = the C3 scene - 5 x 5 grid, a 15 m/s radial mover, a slow mover - with the
    movers advanced by f * 0.1 s): the echoes the driver synthesises on the device are downloaded and focused by the
    oracle in complex128; the frame's three planes in the product stack are held to the oracle's, and the radial mover's
    ATI phase and its motion between frames are there."""
    import sarx
    from oracle import csa_oracle as orc
    from sarx.batch import TwoChannelBatch
    ctx = sarx.default_context()
    frames = 3
    b = TwoChannelBatch(ctx, n, frames, stack="products", scene="c3", scene_scale=scene_scale)
    b.run()
    ctx.sync()
    st = b.stack()
    peaks = []
    for f in (0, 2):
        rx1 = b._alloc[f][0].download(np.complex64, (n + 1, n))
        rx2 = b._alloc[f][1].download(np.complex64, (n + 1, n))
        r1, r2 = rx1[1:], rx2[:-1]                                       # :402-403
        assert np.abs(r1).max() > 0 and np.abs(r1 - r2).max() > 1e-3 * np.abs(r1).max()
        o1 = orc.sar_focus_csa_lean(r1, *b.focus_args, workers=8)[0].T   # [n_az x n_rg] like the stack planes
        o2 = orc.sar_focus_csa_lean(r2, *b.focus_args, workers=8)[0].T
        ref = orc.ati_dpca(o1, o2)
        inside = ref["slc1_mag"] > 0.05 * ref["max_mag"] * (1 + 1e-4)
        outside = ref["slc1_mag"] < 0.05 * ref["max_mag"] * (1 - 1e-4)
        assert inside.sum() > 20
        d = np.angle(np.exp(1j * (st[f, 0][inside].astype(np.float64) - ref["ati_phase"][inside])))
        assert np.linalg.norm(d) / np.linalg.norm(ref["ati_phase"][inside]) < 1e-4
        assert (st[f, 0][outside] == 0).all()
        assert orc.rel_l2(st[f, 1], ref["slc1_mag"]) < 1e-4
        assert orc.rel_l2(st[f, 2][inside], ref["dpca_mag"][inside]) < 1e-3      # difference of nearly equal images: relative to ITSELF ...
        assert np.linalg.norm(st[f, 2].astype(np.float64) - ref["dpca_mag"]) < 1e-4 * np.linalg.norm(ref["slc1_mag"])   # ... and to ||slc1||
        assert np.abs(ref["ati_phase"][inside]).max() > 0.2              # the radial mover shows an ATI phase
        mover = np.where(inside & (np.abs(ref["ati_phase"]) > 0.2), ref["slc1_mag"], 0)
        peaks.append(np.unravel_index(np.argmax(mover), mover.shape))
    assert peaks[0] != peaks[1]                                          # the mover has moved between frame 0 and frame 2
    b.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_frames_in_flight_do_not_change_the_stack():
    """Frame i of a rank runs on compute lane i % lanes with that lane's own plan and buffers (sarx_select_lane): one, two or
    three frames in flight give the same product stack bit for bit, and so does the global maximum."""
    import sarx
    from sarx.batch import TwoChannelBatch
    ctx = sarx.default_context()
    n, frames = 1024, 7
    ref = gref = None
    for lanes in (1, 2, 3):
        b = TwoChannelBatch(ctx, n, frames, stack="products", lanes=lanes)
        assert b.lanes == lanes
        b.run()
        ctx.sync()
        st, g = b.stack(), b.global_max()
        b.close()
        if ref is None:
            ref, gref = st, g
            assert np.isfinite(st).all() and g == float(np.abs(st).max())
        else:
            np.testing.assert_array_equal(st, ref)
            assert g == gref


@pytest.mark.parametrize("stack,n,frames", [("multilook", 2048, 5), ("magnitude", 1024, 4)])
def test_two_ranks_one_gpu_equal_single_rank(tmp_path, stack, n, frames):
    import sarx
    from sarx.batch import TwoChannelBatch
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_batch64_worker.py"), str(tmp_path), str(n),
           str(frames), stack]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    s0 = np.load(tmp_path / "stack64_rank0.npy")
    s1 = np.load(tmp_path / "stack64_rank1.npy")
    np.testing.assert_array_equal(s0, s1)                        # every rank holds the whole stack
    ctx = sarx.default_context()
    b = TwoChannelBatch(ctx, n, frames, stack=stack)
    b.run()
    ctx.sync()
    single = b.stack()
    g_single = b.global_max()
    b.close()
    rounds = -(-frames // 2)
    assert s0.shape[0] == 2 * rounds
    np.testing.assert_array_equal(s0[:frames], single)           # N-rank stack == 1-rank stack, bit for bit, frame order
    assert (s0[frames:] == 0).all()                              # the pad slot of the last round is zeros, not a stale slot
    # global normalisation (sar_batch_sim.py:337-338): own-frames reduction on the device + all-reduce(max) over the ranks
    # = the maximum of the gathered stack, on every rank and on one rank alone
    g = float(np.abs(single).max())
    assert g_single == g > 0
    for r in (0, 1):
        assert float(np.load(tmp_path / f"gmax64_rank{r}.npy")[0]) == g


@pytest.mark.parametrize("stack", ["multilook", "products"])
def test_rccl_two_gpus_equal_single_rank(tmp_path, stack):
    """The RCCL path itself (in-place ncclAllGather per round on the comm stream) with one rank per GPU: needs two
    devices, skipped on the one-GPU test box (the driver's 8-GPU node is the first place it can run)."""
    import sarx
    from sarx.batch import TwoChannelBatch
    if sarx.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device")
    n, frames = 1024, 5
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SARX_TEST_RCCL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_batch64_worker.py"), str(tmp_path), str(n),
           str(frames), stack]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    s0 = np.load(tmp_path / "stack64_rank0.npy")
    np.testing.assert_array_equal(s0, np.load(tmp_path / "stack64_rank1.npy"))
    ctx = sarx.default_context()
    b = TwoChannelBatch(ctx, n, frames, stack=stack)
    b.run()
    ctx.sync()
    np.testing.assert_array_equal(s0[:frames], b.stack())
    assert (s0[frames:] == 0).all()
    for r in (0, 1):                                             # ncclAllReduce(max) of the per-rank maxima
        assert float(np.load(tmp_path / f"gmax64_rank{r}.npy")[0]) == b.global_max() == float(np.abs(s0).max())
    b.close()
