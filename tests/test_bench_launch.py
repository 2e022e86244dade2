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


def _scalar_config(d):
    """The driver's parser keeps only the scalar values of `config` (round 4's record lost `config.other`, a nested dict):
    everything under `config` is a scalar, `workload` fits 120 characters and names the timer's scope."""
    for k, v in d["config"].items():
        assert v is None or isinstance(v, (str, int, float, bool)), (k, v)
    assert len(d["config"]["workload"]) <= 120
    return True


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
    # the equal sample-stride split is the default (--balance is opt-in); the line says how many ranks met and what the node showed
    assert d["balance"] is None and d["config"]["balanced"] is False and d["config"]["rccl_ranks"] == 2 and isinstance(d["config"]["devices_seen"], int)
    assert d["config"]["shard"] == "samples" and _scalar_config(d)
    # N > 1: BASELINE configs 4 and 5 follow the headline across the same ranks, each with its own one-reduce step — config 4 as a
    # pixel-tile shard (BASELINE's wording), config 5 as a sample shard; scalars under `config` (the driver's parser keeps those)
    c = d["config"]
    assert (c["c4_shard"], c["c4_spp"], c["c5_shard"], c["c5_spp"]) == ("tiles", 4096, "samples", 1024)
    assert c["c4_reduce_ok"] and c["c5_reduce_ok"] and c["c4_ms"] > 0 and c["c5_ms"] > 0
    assert "1920x1080" in c["c4_workload"] and "1024x1024" in c["c5_workload"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--no-other-configs"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0 and not any(k.startswith(("c4_", "c5_")) for k in _last_json(r.stdout)["config"]), r.stderr[-2000:]
    # the default N > 1 step keeps two films in flight (multigpu.ReducePipeline); --sync-reduce is rounds 1-5's step
    assert c["reduce"].startswith("pipelined")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--sync-reduce", "--no-other-configs"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    d_sync = _last_json(r.stdout)
    assert r.returncode == 0 and d_sync["reduce_ok"] and d_sync["config"]["reduce"] == "per step", r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--shard", "tiles"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0 and _last_json(r.stdout)["config"]["shard"] == "tiles", r.stderr[-2000:]


def test_a_rank_that_cannot_set_a_sharded_config_up_does_not_hang_the_others():
    """sharded_configs: every rank learns before the first collective of a config whether every rank is ready; a rank that failed
    (out of memory on its GPU, say) makes all of them skip that config — the line carries c4_error, config 5 still runs, exit code 0."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=dict(_env(), MSK_BENCH_TEST_FAIL_CONFIG="1:c4"), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    c = _last_json(r.stdout)["config"]
    assert "another rank" in c["c4_error"] and "c4_ms" not in c and c["c5_reduce_ok"] and _scalar_config({"config": c})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=dict(_env(), MSK_BENCH_TEST_FAIL_CONFIG="0:c5"), cwd=ROOT)
    c = _last_json(r.stdout)["config"]
    assert r.returncode == 0 and "test: rank 0" in c["c5_error"] and c["c4_reduce_ok"]


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


def test_too_few_gpus_ends_before_any_rendezvous():
    """`--gpus N` on a node that shows fewer than N devices: one line on stderr, a non-zero exit code, nothing spawned (bare
    shell) / no rendezvous entered (under a launcher) — asked of a box without a GPU here, so 0 < 2 either way."""
    import time
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=_env(), cwd=ROOT)
    assert r.returncode == 2 and r.stdout == "" and len(r.stderr.strip().split("\n")) == 1 and "this node shows" in r.stderr, r.stderr
    env = dict(_env(), RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_PORT="1")         # a rank as a launcher would start it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode == 2 and "rank 1 stops" in r.stderr and time.time() - t0 < 100
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--in-process", "2", "--gpus", "2"], capture_output=True, text=True,
                       timeout=120, env=_env(), cwd=ROOT)
    assert r.returncode == 2 and "two different multi-GPU paths" in r.stderr


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
    (each rank every tile, its half of the sample indices; once with the equal split that is the default, once with --balance's
    speed-proportional re-split after the warm-up), the film summed onto rank 0 (gloo through host copies here, RCCL on a real
    node) and copied to the host — checked against one rank rendering all the samples."""
    common = ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"]
    for extra in ([], ["--balance"], ["--shard", "tiles"]):
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--spp", "8"] + common + extra,
                            capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
        assert r2.returncode == 0, r2.stderr[-2000:]
        d2 = _last_json(r2.stdout)
        assert d2["n_gpus"] == 2 and d2["rehearsal"] and d2["film_finite"] and d2["value"] > 0
        assert d2["config"]["samples_per_step"] == 512 * 512 * 16 and d2["config"]["rccl_ranks"] == 2
        assert _scalar_config(d2) and d2["timing_scope"] == d2["config"]["timing_scope"] == "incl_copyback" and "copy-back" in d2["config"]["workload"]
        assert d2["config"]["shard"] == ("tiles" if "tiles" in extra else "samples")
        if not extra:
            assert d2["balance"] is None and d2["config"]["balanced"] is False
        if "--balance" in extra:
            assert d2["config"]["balanced"] == (d2["balance"] is not None)
        if "tiles" in extra:
            d_tiles = d2
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rehearse-on-one-gpu", "--spp", "16"] + common,
                        capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = _last_json(r1.stdout)
    # the same 16 samples per pixel either way: the filter-weight sums agree up to the re-association of two partial sums
    assert abs(d2["film_weight_sum"] - d1["film_weight_sum"]) <= 1e-5 * abs(d1["film_weight_sum"])
    assert abs(d_tiles["film_weight_sum"] - d1["film_weight_sum"]) <= 1e-5 * abs(d1["film_weight_sum"])
    assert _scalar_config(d1) and d1["config"]["shard"] == "none"
    # the sharded config 4 / 5 runs behind the headline (reduced sample counts: two ranks share this one GPU and its memory)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--spp", "8", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--c4-spp", "16", "--c5-spp", "8"], capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    c = _last_json(r.stdout)["config"]
    assert c["c4_film_finite"] and c["c5_film_finite"] and c["c4_msamples_per_s"] > 0 and c["c5_msamples_per_s"] > 0 and "c4_error" not in c
    # every sample splats filter weights that sum to about one: all of them arrived in the reduced film
    assert abs(c["c4_weight_per_sample"] - 1) < 0.01 and abs(c["c5_weight_per_sample"] - 1) < 0.01 and _scalar_config({"config": c})


@pytest.mark.gpu
def test_in_process_group_rehearsed_on_one_gpu():
    """`--in-process 2`: ONE process, two member contexts behind one msk_ctx (both on cuda:0 here), the library shards the
    samples over them and sums the films with k_film_sum — the line of the other multi-GPU path (msk_multi.h)."""
    common = ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--spp", "8"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--in-process", "2", "--rehearse-on-one-gpu"] + common,
                       capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["rehearsal"] and d["film_finite"] and d["config"]["in_process_members"] == 2
    assert d["config"]["samples_per_step"] == 512 * 512 * 16 and "k_film_sum" in d["config"]["parallelism"] and _scalar_config(d)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rehearse-on-one-gpu", "--spp", "16"] + common[:-2],
                        capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    d1 = _last_json(r1.stdout)
    assert abs(d["film_weight_sum"] - d1["film_weight_sum"]) <= 1e-5 * abs(d1["film_weight_sum"])


def test_build_id_ignores_comments_and_white_space():
    """bench.py's `roofline.profile_stale` compares tools/build_id.py's hash of the GPU library's sources: it must change with the
    code and not with a reworded comment; the committed profile summaries carry the id of the tree they were taken on."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_id as b
    assert b._strip('int a = 1; // one\n/* two */ const char *s = "// not a comment";') == 'inta=1;constchar*s="//notacomment";'
    assert b._strip("x = 1;") != b._strip("x = 2;")
    assert len(b.build_id()) == 12
    summary = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
    assert summary.get("_build_id") == b.build_id(), "profiles/pmc_summary.json was taken on another build: re-run tools/profile_all.sh + summarize_profiles.py"


def test_build_flags_are_the_contracts_and_part_of_the_build_id(monkeypatch):
    """The two numerics flags of DESIGN.md §3 are in the one place every build reads (__graft_entry__.HIPCC_FLAGS, through
    tools/build_id.py --flags also tools/build_variant.sh and kernel_regs.sh), and a change of flags is a change of build."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_id as b
    flags = b._flags()
    assert "-ffp-contract=off" in flags and "-fhip-fp32-correctly-rounded-divide-sqrt" in flags and "--offload-arch=gfx950" in flags
    assert not any(f.startswith("-ffast-math") or f == "-Ofast" for f in flags)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert flags == g.HIPCC_FLAGS
    before = b.build_id()
    monkeypatch.setattr(b, "_flags", lambda: flags + ["-DSOMETHING"])
    assert b.build_id() != before
