"""Worker for tests/test_batch.py: one rank of a gloo job running the VideoSAR batch driver on
host arrays.  The per-frame engine here is the CPU oracle (test stand-in for the GPU engine)."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # rank start, before anything can touch a GPU (dmabuf IPC for RCCL)

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import conftest  # noqa: F401,E402
from oracle import csa_oracle as orc  # noqa: E402
from sarx import batch  # noqa: E402


def frame_slot(f):
    raw, k = orc.point_scene(64, 64, seed=1000 + f, n_targets=3)
    img = orc.sar_focus_csa(raw, *orc.focus_args(k))[0]
    p = np.abs(img) ** 2
    return p.reshape(16, 4, 16, 4).mean(axis=(1, 3)).astype(np.float32)       # 4x4 multilook


def main():
    out, n_frames = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    comm = batch.TorchStackComm()
    stack = batch.run_batch_host(list(range(n_frames)), comm.world, comm.rank, frame_slot, comm)
    np.save(os.path.join(out, f"stack_rank{comm.rank}.npy"), stack)
    # the global display maximum (sar_batch_sim.py:337-338) from each rank's OWN frames + one all-reduce(max)
    mine = [stack[f] for f in batch.shard_frames(n_frames, comm.world, comm.rank)]
    np.save(os.path.join(out, f"gmax_rank{comm.rank}.npy"), np.array([batch.global_max_host(mine, comm)]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
