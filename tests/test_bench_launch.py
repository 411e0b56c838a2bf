"""bench.py --gpus N: the launcher logic that runs before anything touches a GPU (CPU tests).

BASELINE.json's north star asks for frames/s at 1, 2, 4 and 8 GPUs; `python bench.py --gpus N` has to measure N GPUs whether
or not it was started under torch.distributed.run (round 3: --gpus was parsed and never read)."""
import argparse
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(gpus):
    return argparse.Namespace(gpus=gpus)


def test_single_gpu_run_is_this_process():
    assert bench.launch_plan(_args(1), [], {}) == ("run",)
    assert bench.launch_plan(_args(1), [], {"WORLD_SIZE": "1"}) == ("run",)


@pytest.mark.parametrize("n", [2, 4, 8])
def test_gpus_n_outside_a_launcher_spawns_n_ranks(n):
    argv = ["--gpus", str(n), "--steps", "10", "--shard-db"]
    kind, cmd = bench.launch_plan(_args(n), argv, {})
    assert kind == "spawn"
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert f"--nproc-per-node={n}" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                     # the ranks get exactly this command line


def test_rank_of_a_matching_launch_runs():
    assert bench.launch_plan(_args(8), [], {"WORLD_SIZE": "8", "RANK": "3"}) == ("run",)


@pytest.mark.parametrize("gpus, world", [(8, "4"), (1, "2"), (2, "1"), (2, "x")])
def test_mismatch_is_refused(gpus, world):
    kind, text = bench.launch_plan(_args(gpus), [], {"WORLD_SIZE": world})
    assert kind == "error" and "WORLD_SIZE" in text


def test_mismatch_exits_nonzero_before_any_gpu_call():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and not r.stdout.strip()


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="on a GPU box this would run the whole benchmark")
def test_bare_gpus_2_starts_two_ranks_here():
    """No GPU in the build container: both ranks stop at the 'needs an MI355X' check -- which shows that `python bench.py
    --gpus 2` really started torch.distributed.run with two ranks of bench.py and relayed their exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-2000:]


def test_force_dist_initialises_a_group_of_one_rank(monkeypatch):
    """--force-dist: a process group at world size 1 (gloo here; nccl = RCCL on the GPU box, tests/_rccl_world1.py), usable for the
    barrier and the all_reduce(MAX) of the elapsed time; without the flag a single rank gets no group at all."""
    import torch
    for k in ("MASTER_PORT", "MASTER_ADDR", "WORLD_SIZE", "RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.init_dist(argparse.Namespace(force_dist=False, backend="gloo"), torch, 0, 1, 0) is None
    dist = bench.init_dist(argparse.Namespace(force_dist=True, backend="gloo"), torch, 0, 1, 0)
    try:
        assert dist is not None and dist.is_initialized() and dist.get_world_size() == 1
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and 1024 < int(os.environ["MASTER_PORT"]) < 65536
        dist.barrier()
        t = torch.tensor([1.25], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.25
    finally:
        if dist is not None:
            dist.destroy_process_group()
