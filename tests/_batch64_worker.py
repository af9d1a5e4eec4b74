"""One rank of a two-rank job for tests/test_gpu_batch64.py: sarx.batch.TwoChannelBatch on the GPU with the stack
gathered through a gloo group (both ranks share the one GPU of the test box; RCCL refuses two ranks on one device,
which is what the host transport is for) - or, with SARX_TEST_RCCL=1 on a box with two GPUs, one rank per GPU and the
in-place RCCL all-gather."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # rank start, before `import sarx` / any GPU call (dmabuf IPC for RCCL)

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import conftest  # noqa: F401,E402
import sarx  # noqa: E402
from sarx.batch import TorchStackComm, TwoChannelBatch  # noqa: E402


def main():
    out, n, n_frames, stack = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo")
    comm = TorchStackComm()
    if os.environ.get("SARX_TEST_RCCL") == "1":
        from sarx.batch import RcclStackComm
        ctx = sarx.Context(int(os.environ.get("LOCAL_RANK", "0")))
        RcclStackComm(ctx, comm.world, comm.rank, dist)                            # raises on every rank together if the bootstrap fails
        b = TwoChannelBatch(ctx, n, n_frames, comm.world, comm.rank, stack=stack, looks=16, rccl=True)
    else:
        ctx = sarx.Context(0)
        b = TwoChannelBatch(ctx, n, n_frames, comm.world, comm.rank, stack=stack, looks=16, host_comm=comm)
    b.run()
    ctx.sync()
    whole = b.d_stack.download(np.float32, (b.n_rounds * comm.world, *b.slot_shape))      # pad slots included
    np.save(os.path.join(out, f"stack64_rank{comm.rank}.npy"), whole)
    np.save(os.path.join(out, f"gmax64_rank{comm.rank}.npy"), np.array([b.global_max()]))
    b.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
