"""The N > 1 path on CPU: two processes, gloo backend, each renders its pixel-tile shard with the
CPU oracle (standing in for the GPU back end behind the same msk_render_params shard selectors),
one film reduce onto rank 0 — the exact sequence bench.py runs over RCCL."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_path, mode="tiles"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    abi = importlib.import_module("misaki-render_amd.abi")
    hm = importlib.import_module("misaki-render_amd.hostmirror")
    mg = importlib.import_module("misaki-render_amd.multigpu")
    import oracle_binding
    orc = oracle_binding.load()
    flat = hm.cbox_scene(96, 64, coeff_lookup=lambda rgb: (0.0, 0.0, 1.0))
    sc = orc.scene(flat)
    spp_total = mg.weak_scaling_spp(2, world)
    if mode == "range":
        # the speed-proportional split bench.py switches to when the GPUs of a node differ: every rank derives the same
        # shares from the gathered step times, then renders its contiguous range of sample indices
        none, times = mg.speed_proportional_shares(dist, 50.0 if rank == 0 else 51.0, spp_total)        # within 4 %: keep the equal split
        assert none is None and times == [50.0, 51.0]
        shares, times = mg.speed_proportional_shares(dist, 30.0 if rank == 0 else 90.0, spp_total)
        assert shares == [3, 1] and times == [30.0, 90.0]
        prm = mg.shard_params(abi, spp_total, rank, world, mode="range", shares=shares, seed=5)
        assert (prm.sample_first, prm.sample_stride, prm.spp) == ((0, 1, 3) if rank == 0 else (3, 1, 4))
    else:
        prm = mg.shard_params(abi, spp_total, rank, world, mode=mode, seed=5)
    if mode == "range":
        pass
    elif mode == "tiles":
        assert prm.block_first == rank and prm.block_stride == world and prm.spp == 2 * world
    else:
        assert prm.sample_first == rank and prm.sample_stride == world and prm.spp == 2 * world and prm.block_stride == 1
    film_np, st = sc.render(prm, threads=2)
    film = torch.from_numpy(film_np.copy())
    mg.reduce_film(film, dist)
    n = torch.tensor([float(st.samples)], dtype=torch.float64)
    dist.all_reduce(n)
    if rank == 0:
        full, fst = sc.render(abi.render_params(spp=spp_total, seed=5), threads=2)
        np.savez(out_path, reduced=film.numpy(), full=full, samples=n.numpy(), full_samples=fst.samples)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_shard_and_film_reduce(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "film.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    d = np.load(out)
    reduced, full = d["reduced"], d["full"]
    assert d["samples"][0] == d["full_samples"] == 96 * 64 * 4       # every sample rendered exactly once
    interior = np.ones((64, 96), bool)
    for k in range(0, 96, 32):
        interior[:, max(0, k - 2):k + 2] = False
    for k in range(0, 64, 32):
        interior[max(0, k - 2):k + 2, :] = False
    assert np.array_equal(reduced[interior], full[interior])         # exactly one rank contributes
    assert np.allclose(reduced, full, rtol=3e-7, atol=1e-6)           # tile borders: <= 4 terms re-associated
    assert reduced[..., 4].min() > 0


def test_two_rank_sample_shard_and_film_reduce(tmp_path):
    """bench.py's default for N > 1: every rank renders all tiles for its sample indices; one reduce."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "film_s.npz")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out, "samples"), nprocs=2, join=True)
    d = np.load(out)
    reduced, full = d["reduced"], d["full"]
    assert d["samples"][0] == d["full_samples"] == 96 * 64 * 4       # every sample rendered exactly once
    assert np.allclose(reduced, full, rtol=3e-6, atol=1e-6)           # two partial sums per pixel re-associated
    hm = importlib.import_module("misaki-render_amd.hostmirror")
    err = np.linalg.norm(hm.develop(reduced)[..., :3] - hm.develop(full)[..., :3], axis=-1)
    assert err.max() < 1e-4                                           # the north star's per-pixel L2 tolerance


def test_tile_shard_and_sample_shard_reduce_to_the_same_film(tmp_path):
    """BASELINE config 4 words its decomposition "pixel-tile shard", bench.py defaults to the sample shard (`--shard`): the two
    world-size-2 runs must reduce to films within the north star's 1e-4 per-pixel L2 of EACH OTHER (they already are of the
    single-rank film, separately) — whichever axis a node's ranks split, the film on rank 0 is the same film."""
    import torch.multiprocessing as mp
    films = {}
    for k, mode in enumerate(("tiles", "samples")):
        out = str(tmp_path / ("film_%s.npz" % mode))
        mp.spawn(_worker, args=(2, 35500 + (os.getpid() % 2000) + k, out, mode), nprocs=2, join=True)
        d = np.load(out)
        assert d["samples"][0] == d["full_samples"] == 96 * 64 * 4
        films[mode] = d["reduced"]
    hm = importlib.import_module("misaki-render_amd.hostmirror")
    a, b = hm.develop(films["tiles"])[..., :3], hm.develop(films["samples"])[..., :3]
    err = np.linalg.norm(a.astype(np.float64) - b.astype(np.float64), axis=-1)
    assert err.max() < 1e-4 and np.allclose(films["tiles"], films["samples"], rtol=3e-6, atol=1e-6)
    assert np.array_equal(films["tiles"][..., 4] > 0, films["samples"][..., 4] > 0)


def _config_worker(rank, world, port, out_path, which):
    """bench.py's sharded_configs on CPU: config 4's shape (a 16:9 film whose last row of blocks is ragged, pixel-tile shard) and
    config 5's (a rough-dielectric mesh in the room, sample shard) through the same shard selectors and the same one reduce."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    abi = importlib.import_module("misaki-render_amd.abi")
    hm = importlib.import_module("misaki-render_amd.hostmirror")
    mg = importlib.import_module("misaki-render_amd.multigpu")
    import oracle_binding
    orc = oracle_binding.load()
    if which == "c4":
        flat, spp, mode = hm.cbox_scene(120, 68, coeff_lookup=lambda rgb: (0.0, 0.0, 1.0)), 6, "tiles"       # 1920 x 1080 / 16: blocks 4 x 3, the last row 4 px high
    else:
        blob = hm.blob_mesh("blob", (278, 200, 280), 120, 14, 14, hm.WHITE, seed=7, bump=0.25)
        blob.bsdf = {"type": "roughdielectric", "alpha": 0.1, "int_ior": 1.5, "ext_ior": 1.0}
        flat, spp, mode = hm.flatten(hm.cbox_meshes()[:6] + [blob], 64, 64, coeff_lookup=lambda rgb: (0.0, 0.0, 1.0)), 4, "samples"
    sc = orc.scene(flat)
    prm = mg.shard_params(abi, spp, rank, world, mode=mode, seed=0)
    film_np, st = sc.render(prm, threads=2)
    film = torch.from_numpy(film_np.copy())
    mg.reduce_film(film, dist)
    n = torch.tensor([float(st.samples)], dtype=torch.float64)
    dist.all_reduce(n)
    if rank == 0:
        full, fst = sc.render(abi.render_params(spp=spp, seed=0), threads=2)
        np.savez(out_path, reduced=film.numpy(), full=full, samples=n.numpy(), full_samples=fst.samples)
    dist.barrier()
    dist.destroy_process_group()


def test_the_sharded_baseline_configs_reduce_to_the_single_rank_film(tmp_path):
    """BASELINE configs 4 and 5 as bench.py runs them on N > 1 ranks (sharded_configs): config 4 = pixel-tile shard of a 16:9 film
    with a ragged last block row, config 5 = sample shard of a scene with a rough-dielectric mesh; each reduced film equals the
    single-rank film within the north star's 1e-4 per-pixel L2, every sample rendered once."""
    import torch.multiprocessing as mp
    hm = importlib.import_module("misaki-render_amd.hostmirror")
    for k, (which, n_samples) in enumerate((("c4", 120 * 68 * 6), ("c5", 64 * 64 * 4))):
        out = str(tmp_path / (which + ".npz"))
        mp.spawn(_config_worker, args=(2, 37500 + (os.getpid() % 2000) + k, out, which), nprocs=2, join=True)
        d = np.load(out)
        assert d["samples"][0] == d["full_samples"] == n_samples, which
        assert np.allclose(d["reduced"], d["full"], rtol=3e-6, atol=1e-6), which
        err = np.linalg.norm(hm.develop(d["reduced"])[..., :3].astype(np.float64) - hm.develop(d["full"])[..., :3].astype(np.float64), axis=-1)
        assert err.max() < 1e-4 and d["reduced"][..., 4].min() > 0, which


def _pipeline_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mg = importlib.import_module("misaki-render_amd.multigpu")
    films = [torch.zeros((6, 4, 5)) for _ in range(2)]
    hosts = [torch.zeros((6, 4, 5)) for _ in range(2)] if rank == 0 else [None, None]
    pipe = mg.ReducePipeline(films, hosts, dist, rank)
    seen = []
    for k in range(5):                       # "render" step k: every rank writes (k + 1) * (rank + 1) into the slot it is handed
        f = pipe.begin()
        f.fill_(float((k + 1) * (rank + 1)))
        slot = pipe.submit()
        assert slot == k % 2
        if rank == 0:
            seen.append(float(pipe.last_host_film()[0, 0, 0]))
    pipe.drain()
    if rank == 0:
        np.save(out_path, np.array(seen))
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_pipeline_hands_out_two_slots_and_reduces_every_step(tmp_path):
    """multigpu.ReducePipeline (bench.py's N > 1 step): slots alternate, every step's film is reduced onto rank 0 and copied to its
    host film of that slot — step k of two ranks sums to (k + 1) * (1 + 2)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "seen.npy")
    mp.spawn(_pipeline_worker, args=(2, 38500 + (os.getpid() % 2000), out), nprocs=2, join=True)
    assert np.array_equal(np.load(out), np.array([3.0, 6.0, 9.0, 12.0, 15.0]))


def test_two_rank_speed_proportional_ranges(tmp_path):
    """Unequal shares (3 : 1) as contiguous sample ranges: every sample still rendered exactly once."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "film_r.npz")
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out, "range"), nprocs=2, join=True)
    d = np.load(out)
    reduced, full = d["reduced"], d["full"]
    assert d["samples"][0] == d["full_samples"] == 96 * 64 * 4
    assert np.allclose(reduced, full, rtol=3e-6, atol=1e-6)


def test_balanced_shares():
    mg = importlib.import_module("misaki-render_amd.multigpu")
    for times in ([54, 54, 60, 60, 55, 57, 54, 60], [50, 50], [10, 1000], [1, 1, 1]):
        for total in (len(times), 512 * len(times), 4097):
            sh = mg.balanced_shares(total, times)
            assert sum(sh) == total and min(sh) >= 1 and len(sh) == len(times)
            if total >= 64 * len(times):
                # proportional to speed within one sample
                speed = np.array([1.0 / t for t in times]); want = total * speed / speed.sum()
                assert np.abs(np.array(sh) - want).max() <= 1.0 + 1e-9
    assert mg.balanced_shares(1024, [50, 50]) == [512, 512]
    abi = importlib.import_module("misaki-render_amd.abi")
    import pytest
    with pytest.raises(ValueError):
        mg.shard_params(abi, 8, 0, 2, mode="range", shares=[5, 4])
