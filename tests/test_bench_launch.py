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
