#!/usr/bin/env python3
"""BASELINE config 5 alone (for rocprofv3): the 64-frame VideoSAR batch of two-channel 8192 x 8192 scenes on ONE GPU
through sarx.batch.TwoChannelBatch - the driver bench.py's `batch64` block and tests/test_gpu_batch64.py run.
Multi-GPU runs go through bench.py (`python bench.py --gpus N`), which owns the rendezvous and the RCCL bootstrap.

    python3 tools/bench_batch64.py [--frames 64] [--size 8192] [--stack multilook|magnitude]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nis-sar-amtigmti-video_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--stack", choices=("multilook", "magnitude", "products"), default="multilook")
    ap.add_argument("--scene", choices=("noise", "c3"), default="noise", help="c3: SURVEY.md 8(d) C5's content, movers advanced by f * 0.1 s per frame")
    ap.add_argument("--lanes", type=int, default=None, help="frames in flight (compute lanes of the context)")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import sarx
    from sarx.batch import TwoChannelBatch
    ctx = sarx.Context(0)
    b = TwoChannelBatch(ctx, a.size, a.frames, stack=a.stack, scene=a.scene, lanes=a.lanes)
    b.prepare()
    b.run()
    ctx.sync()
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        b.run()
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    print(json.dumps({"metric": "VideoSAR batch frames/sec (two-channel CSA focus + ATI/DPCA + mask, stack slot)",
                      "value": a.frames / dt, "unit": "frames/s", "n_gpus": 1, "batch_s": dt, "ms_per_frame": dt / a.frames * 1e3,
                      "config": {"workload": f"{a.frames} frames x two-channel {a.size}x{a.size} complex64", "stack": a.stack, "scene": a.scene, "frames_in_flight": b.lanes}}))


if __name__ == "__main__":
    main()
