"""bench.py's multi-rank launch paths, on CPU: `--dry-run` keeps the argument handling, the rank spawn, the
rendezvous (gloo instead of RCCL), the film reduce, the barrier-bracketed timing and the one JSON line of rank 0,
and leaves out only the GPU render."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def test_bare_shell_gpus_2_spawns_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["dry_run"] and d["reduce_ok"] and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["samples_per_step"] == 512 * 512 * 512 * 2          # weak scaling: 512 spp per rank


def test_under_torch_distributed_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["reduce_ok"]


def test_single_rank_dry_run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True, timeout=300,
                       env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 1


import pytest


def test_a_dead_rank_ends_the_launch_quickly():
    """A rank that exits before the rendezvous must not leave the others (and the shell) waiting for the process-group
    timeout: the launcher stops the siblings and returns the failing rank's code."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=dict(os.environ, MSK_BENCH_TEST_FAIL_RANK="1"))
    assert r.returncode == 3 and "rank 1 exited with code 3" in r.stderr and time.time() - t0 < 60


@pytest.mark.gpu
def test_two_ranks_rehearsed_on_one_gpu():
    """The N = 2 path with real renders: two rank processes spawned from a bare shell, both on cuda:0, sample shards
    (each rank every tile, its half of the sample indices), the warm-up's speed-proportional re-split, the film summed onto
    rank 0 (gloo through host copies here, RCCL on a real node) — checked against one rank rendering all the samples."""
    common = ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"]
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--spp", "8"] + common,
                        capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r2.returncode == 0, r2.stderr[-2000:]
    d2 = _last_json(r2.stdout)
    assert d2["n_gpus"] == 2 and d2["rehearsal"] and d2["film_finite"] and d2["value"] > 0
    assert d2["config"]["samples_per_step"] == 512 * 512 * 16
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rehearse-on-one-gpu", "--spp", "16"] + common,
                        capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = _last_json(r1.stdout)
    # the same 16 samples per pixel either way: the filter-weight sums agree up to the re-association of two partial sums
    assert abs(d2["film_weight_sum"] - d1["film_weight_sum"]) <= 1e-5 * abs(d1["film_weight_sum"])
