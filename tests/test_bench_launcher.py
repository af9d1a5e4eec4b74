"""bench.py's own launcher: `python bench.py --gpus N` with WORLD_SIZE unset must start N ranks as child processes
(before anything touches a GPU), and a --gpus that disagrees with WORLD_SIZE must fail instead of reporting 1 GPU.
Runs on the CPU box: --dry-run stops every rank after the gloo rendezvous."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    env["OMP_NUM_THREADS"] = "1"
    return subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True, timeout=300)


def test_bare_gpus_n_spawns_n_ranks():
    r = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "spawning 2 ranks" in r.stderr
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout       # gloo / RCCL banners must not reach stdout: ONE JSON line
    line = json.loads(r.stdout)
    assert line["dry_run"] and line["n_gpus"] == 2
    assert sorted(x["rank"] for x in line["ranks"]) == [0, 1]
    assert sorted(x["local_rank"] for x in line["ranks"]) == [0, 1]
    assert len({x["pid"] for x in line["ranks"]}) == 2 and all(x["world"] == 2 for x in line["ranks"])


def test_single_rank_needs_no_launcher():
    r = _run(["--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "spawning" not in r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    r = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_gpus_disagreeing_with_world_size_is_an_error():
    r = _run(["--gpus", "8", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2
    assert "disagrees with WORLD_SIZE" in r.stderr
    assert "{" not in r.stdout                                   # no JSON line with a wrong n_gpus


def test_child_failure_propagates():
    """A rank that fails (here: an unknown flag) makes the launcher exit non-zero."""
    r = _run(["--gpus", "2", "--dry-run", "--no-such-flag"])
    assert r.returncode != 0


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_driver_style_launch_of_eight_ranks_sets_the_ipc_mode_itself():
    """The other documented launch - `python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8`, the way the driver
    starts the 8-GPU scaling run - with HSA_ENABLE_IPC_MODE_LEGACY absent from the environment: every rank sets it to 0 at its own
    start (RCCL across processes needs dmabuf IPC on this pool), before anything could touch a GPU; one JSON line, eight ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "8", "--dry-run"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["dry_run"] and line["n_gpus"] == 8 and sorted(x["rank"] for x in line["ranks"]) == list(range(8))
    assert all(x["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for x in line["ranks"])


def _eight(env_extra, extra_args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "8", "--dry-run", *extra_args]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)


def test_first_scaling_run_cannot_fall_back_to_host_memory_silently():
    """Eight ranks, eight devices (the driver's scaling run), RCCL bring-up failing on ONE rank: every rank exits non-zero together
    and no JSON line is printed - the first 8-GPU number cannot be a gloo-through-host-memory figure that only a flag marks.  The
    same failure with ranks outnumbering the GPUs (a rehearsal on one card), or with --allow-host-transport, falls back and says so;
    --require-rccl makes the rehearsal strict too.  (agree_on_rccl and the policy are bench.py's own; the communicator is a stand-in.)"""
    r = _eight({"SARX_BENCH_REHEARSE_RCCL_FAIL": "5", "SARX_BENCH_REHEARSE_DEVICES": "8"})
    assert r.returncode != 0, r.stdout
    assert "exiting 3 on every rank" in r.stderr and r.stderr.count("exiting 3 on every rank") == 8
    assert not [x for x in r.stdout.splitlines() if x.startswith("{")]
    r = _eight({"SARX_BENCH_REHEARSE_RCCL_FAIL": "-1", "SARX_BENCH_REHEARSE_DEVICES": "8"})       # the unique id itself fails on rank 0
    assert r.returncode != 0 and r.stderr.count("exiting 3 on every rank") == 8
    r = _eight({"SARX_BENCH_REHEARSE_RCCL_FAIL": "5", "SARX_BENCH_REHEARSE_DEVICES": "1"})        # eight ranks on one card: may fall back
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][0])
    assert line["rccl_rehearsal"] == {"rccl_up": False, "required": False, "devices": 1}
    r = _eight({"SARX_BENCH_REHEARSE_RCCL_FAIL": "5", "SARX_BENCH_REHEARSE_DEVICES": "1"}, ["--require-rccl"])
    assert r.returncode != 0
    r = _eight({"SARX_BENCH_REHEARSE_RCCL_FAIL": "5", "SARX_BENCH_REHEARSE_DEVICES": "8"}, ["--allow-host-transport"])
    assert r.returncode == 0
    r = _eight({"SARX_BENCH_REHEARSE_RCCL_FAIL": "", "SARX_BENCH_REHEARSE_DEVICES": "8"})          # nothing fails: RCCL up, required, fine
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][0])
    assert line["rccl_rehearsal"] == {"rccl_up": True, "required": True, "devices": 8}
