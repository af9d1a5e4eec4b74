"""VideoSAR batch sharding / stack assembly: unit tests + a world_size-2 gloo run on CPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from sarx import batch


def test_shard_frames_round_robin():
    assert batch.shard_frames(8, 2, 0) == [0, 2, 4, 6]
    assert batch.shard_frames(8, 2, 1) == [1, 3, 5, 7]
    assert batch.shard_frames(5, 4, 3) == [3]
    assert batch.shard_frames(3, 4, 3) == []
    assert batch.shard_frames(0, 2, 1) == []
    assert sorted(sum((batch.shard_frames(64, 8, r) for r in range(8)), [])) == list(range(64))
    with pytest.raises(ValueError):
        batch.shard_frames(4, 2, 2)


def test_stack_from_rounds_orders_and_drops_padding():
    world, n = 4, 6
    blocks = []
    for i in range(batch.rounds(n, world)):
        blk = np.zeros((world, 2, 2), np.float32)
        for r in range(world):
            f = i * world + r
            blk[r] = f if f < n else -1
        blocks.append(blk)
    st = batch.stack_from_rounds(blocks, n)
    assert st.shape == (6, 2, 2)
    assert [int(st[f, 0, 0]) for f in range(n)] == list(range(n))
    norm, g = batch.normalise_stack(st)
    assert g == 5.0 and norm.max() == 1.0
    assert batch.normalise_stack(np.zeros((2, 2, 2)))[1] == 1.0


def test_run_batch_host_frameless_rank_and_pad_rounds():
    """More ranks than frames: the rank that owns nothing sizes its zero slots from slot_shape (or from a shape
    agreed over the transport) instead of raising while its peers wait in the collective."""
    class FakeComm:                       # plays rank `me` of `world`; records what this rank sends each round
        def __init__(self, world, me):
            self.world, self.me, self.sent = world, me, []

        def all_gather(self, slot):
            self.sent.append(np.array(slot))
            return np.stack([slot] * self.world)

        def finish(self):
            pass

    frame = lambda f: np.full((3, 4), float(f + 1), np.float32)
    c = FakeComm(4, 3)
    st = batch.run_batch_host([0, 1], 4, 3, frame, c, slot_shape=(3, 4))       # rank 3 of 4 owns no frame of 2
    assert st.shape == (2, 3, 4) and len(c.sent) == 1 and (c.sent[0] == 0).all() and c.sent[0].shape == (3, 4)
    c = FakeComm(4, 3)                                                         # no slot_shape: one shape round first
    c.all_gather_plain, first = c.all_gather, [True]

    def gather_with_peer_shapes(slot):
        if first[0]:                                                           # ranks 0, 1 own a frame and report (3, 4)
            first[0] = False
            c.sent.append(np.array(slot))
            return np.array([[[3, 4]], [[3, 4]], [[0, 0]], [[0, 0]]], np.float32)
        return c.all_gather_plain(slot)
    c.all_gather = gather_with_peer_shapes
    batch.run_batch_host([0, 1], 4, 3, frame, c)
    assert c.sent[0].tolist() == [[0.0, 0.0]] and c.sent[1].shape == (3, 4) and (c.sent[1] == 0).all()
    c = FakeComm(2, 1)
    batch.run_batch_host([0, 1, 2], 2, 1, frame, c)                           # rank 1 of 2: frame 1, then a pad round
    assert [float(x.max()) for x in c.sent] == [2.0, 0.0] and c.sent[1].shape == (3, 4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_frames", [4, 5])
def test_two_rank_gloo_stack_equals_single_process(tmp_path, n_frames):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_batch_worker.py"), str(tmp_path), str(n_frames)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    s0 = np.load(tmp_path / "stack_rank0.npy")
    s1 = np.load(tmp_path / "stack_rank1.npy")
    np.testing.assert_array_equal(s0, s1)                    # every rank holds the whole stack
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _batch_worker as w
    single = np.stack([w.frame_slot(f) for f in range(n_frames)])
    assert s0.shape == single.shape == (n_frames, 16, 16)
    np.testing.assert_array_equal(s0, single)                # N-rank stack == 1-rank stack, bit for bit


def test_eight_rank_gloo_thirteen_frames(tmp_path):
    """The 8-GPU launch shape on the CPU: 13 frames on 8 ranks (two rounds, the second with five frames and three pad
    slots, ranks 5-7 own one frame only).  Every rank ends with the whole stack = the one-process stack bit for bit, and the
    global maximum from own-frames + all-reduce(max) equals the maximum of the gathered stack on every rank."""
    n_frames, world = 13, 8
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_batch_worker.py"), str(tmp_path), str(n_frames)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _batch_worker as w
    single = np.stack([w.frame_slot(f) for f in range(n_frames)])
    for rank in range(world):
        s = np.load(tmp_path / f"stack_rank{rank}.npy")
        assert s.shape == (n_frames, 16, 16)
        np.testing.assert_array_equal(s, single)
        assert float(np.load(tmp_path / f"gmax_rank{rank}.npy")[0]) == float(np.abs(single).max())


def test_global_max_host_local_and_empty():
    c = batch.LocalStackComm()
    assert batch.global_max_host([np.array([[1.0, -3.0]]), np.array([[2.0]])], c) == 3.0
    assert batch.global_max_host([], c) == 1.0                       # a rank without frames / an all-zero stack: 0 -> 1.0 (:338)
    assert batch.global_max_host([np.zeros((2, 2))], c) == 1.0
