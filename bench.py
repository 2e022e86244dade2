#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s of the path-tracing hot path on the synthetic
Cornell box at 512x512, 512 spp per GPU (BASELINE.json configs[1]), diffuse BSDFs.

    python bench.py --gpus N --steps K --warmup W

A step = one complete render through the C ABI (msk_gpu_render_device): wavefront path tracing
of every sample + the ordered film resolve, film left in HBM.  Inputs (scene, BVH) are resident
in HBM before the timed region.  For N > 1 the driver launches one rank per GPU
(torch.distributed.run); rank r renders the spiral blocks id = r (mod N) at 512*N spp — per-GPU
work is constant (weak scaling) — and the films are summed onto rank 0 with one RCCL reduce
inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH = HEIGHT = 512
SPP_PER_GPU = 512
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable


def cpu_baseline(abi, hm, flat, seconds_target=15.0):
    """The CPU oracle (a port of the reference's Embree3+TBB loop, see oracle/oracle.cpp) timed on
    this box's host cores on a bounded sample of the same workload.  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding
    orc = oracle_binding.load()
    threads = min(8, os.cpu_count() or 1)         # the reference CLI caps TBB at 8 threads (main.cpp:43-44)
    sc = orc.scene(flat)
    spp = 1
    prm = abi.render_params(spp=spp, rng_mode=abi.MSK_RNG_PCG_BLOCK)
    t0 = time.perf_counter()
    sc.render(prm, threads)
    dt = time.perf_counter() - t0
    spp = int(max(1, min(64, seconds_target / max(dt, 1e-3))))
    prm = abi.render_params(spp=spp, rng_mode=abi.MSK_RNG_PCG_BLOCK)
    t0 = time.perf_counter()
    _, st = sc.render(prm, threads)
    dt = time.perf_counter() - t0
    sc.close()
    return {"value": round(st.samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"cbox {WIDTH}x{HEIGHT} @ {spp} spp ({st.samples} samples, {dt:.1f} s), pcg_block sampler, "
                      f"own BVH instead of Embree", "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP_PER_GPU, help="spp per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", choices=("samples", "tiles"), default="samples",
                    help="N > 1: which axis the ranks split (misaki-render_amd/multigpu.py); both end in one film reduce")
    ap.add_argument("--no-balance", action="store_true",
                    help="N > 1: keep the equal interleaved sample split even when the GPUs differ in speed")
    args = ap.parse_args()

    # Per-kernel HIP-event timing on every 4th sync group (8 iterations each), rotating over the steps: a timed dispatch costs
    # ~6 us, 2 % of a step if every launch carries events.  The roofline's average launch duration is the
    # average over the timed launches.
    os.environ.setdefault("MSK_TIMING_EVERY", "4")
    import torch
    abi = importlib.import_module("misaki-render_amd.abi")
    hm = importlib.import_module("misaki-render_amd.hostmirror")
    mg = importlib.import_module("misaki-render_amd.multigpu")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n = args.gpus
    if world != n and world > 1:
        n = world
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    flat = hm.cbox_scene(WIDTH, HEIGHT)
    ctx = abi.Context(local_rank)
    scene = abi.Scene(ctx, flat)
    spp_total = mg.weak_scaling_spp(args.spp, world)
    prm = mg.shard_params(abi, spp_total, rank, world, mode=args.shard, seed=0)
    film = torch.zeros((HEIGHT, WIDTH, 5), dtype=torch.float32, device="cuda")

    def step():
        st = scene.render_device(prm, film.data_ptr())
        mg.reduce_film(film, dist)
        return st

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    warm = [step() for _ in range(args.warmup)]
    balance = None
    if dist is not None and args.shard == "samples" and not args.no_balance and warm:
        # The GPUs of a node are not equally fast on this workload (the shading kernel differs by up to 20 %
        # between boxes of the pool) and an equal split waits for the slowest.  Every rank's device time of the last
        # warm-up step decides speed-proportional contiguous sample ranges; the per-GPU average stays args.spp.
        # (kernel time of the wavefront launches that carried events — the same sync groups on every rank — not the step's
        # wall or device total, which on a first step also holds the one-off uploads of the render plan)
        shares, times = mg.speed_proportional_shares(dist, warm[-1].ms_trace + warm[-1].ms_shade, spp_total, device="cuda")
        if shares is not None:
            prm = mg.shard_params(abi, spp_total, rank, world, mode="range", shares=shares, seed=0)
            step()                       # untimed: the new shares' plan and record buffers are set up here
            balance = {"equal_split_kernel_ms": [round(t, 2) for t in times], "spp_shares": shares}
    fence()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        samples_step = WIDTH * HEIGHT * spp_total              # all ranks together
        value = samples_step * args.steps / dt / 1e6
        # ---- roofline of the dominant kernel, from the HIP events the library records around every
        #      launch on its own stream (rank 0's launches)
        seg = sum(s.segments for s in stats)
        smp = sum(s.samples for s in stats)
        shd = sum(s.shadow_rays for s in stats)
        ms_trace = sum(s.ms_trace for s in stats)
        ms_shade = sum(s.ms_shade for s in stats)
        n_trace = sum(s.n_trace_launches for s in stats)
        n_shade = sum(s.n_shade_launches for s in stats)
        # algorithmic bytes of the SoA path state each kernel must move:
        #   k_trace      48 B/segment (ray_o, ray_d in; hit out) + 16 B/shadow ray (sh in)
        #   k_shade_gen  224 B/segment (96 in, 128 out) + 16 B/shadow ray (contrib in) + 20 B/sample (record out)
        bytes_trace = seg * 48 + shd * 16
        bytes_shade = seg * 224 + shd * 16 + smp * 20
        # the two kernels take nearly the same time on this workload and which one is ahead depends on the box; the shading
        # kernel (five times the bytes) is reported unless the traversal kernel is clearly the longer one
        if ms_trace > 1.03 * ms_shade:
            name, b, ms, nl = "k_trace", bytes_trace, ms_trace, n_trace
        else:
            name, b, ms, nl = "k_shade_gen", bytes_shade, ms_shade, n_shade
        iters = sum(s.iterations for s in stats)               # one launch of each kernel per wavefront iteration
        bytes_per_launch = b / max(iters, 1)
        avg_launch_ms = ms / max(nl, 1)                        # over the timed launches (nl of iters)
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        ms_wavefront = sum(s.ms_total - s.ms_resolve for s in stats)
        pair_achieved = (bytes_trace + bytes_shade) / (ms_wavefront * 1e-3) / 1e9 if ms_wavefront > 0 else 0.0
        def pmc_traffic(prefix):
            prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
            try:
                pm = json.load(open(prof))
                return next((v["hbm_bytes_per_launch"] for k, v in pm.items() if k.startswith(prefix)), None)
            except Exception:
                return None
        n_streams = int(os.environ.get("MSK_STREAMS", "4"))
        per_kernel = {"kernel": name, "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
                      "bytes_per_launch": round(bytes_per_launch), "avg_launch_ms": round(avg_launch_ms, 4),
                      "avg_launch_ms_shade": round(ms_shade / max(n_shade, 1), 4), "avg_launch_ms_trace": round(ms_trace / max(n_trace, 1), 4)}
        if n_streams > 1:
            # The wavefront loop runs on several streams at once: at any moment a few launches of BOTH kernels
            # share the GPU, a launch's wall duration says how long it shared, not how fast it could go, and the unit that
            # has a bandwidth is the phase: both kernels' algorithmic bytes of one iteration over the wall time an iteration
            # takes.  The per-launch figures (what a rocprofv3 kernel trace of this command shows) stay under `per_kernel`.
            name = "k_shade_gen || k_trace (%d streams)" % n_streams
            achieved = pair_achieved
            per_iter = max(iters // n_streams, 1)            # iterations of the whole pool (every stream launches its own)
            bytes_per_launch = (bytes_trace + bytes_shade) / per_iter
            avg_launch_ms = ms_wavefront / per_iter
            ts, tt = pmc_traffic("k_shade_gen"), pmc_traffic("k_trace")
            traffic = round((ts + tt) * n_streams) if ts is not None and tt is not None else None   # PMC run = the same 4-stream launches
        else:
            t = pmc_traffic(name)
            traffic = round(t) if t is not None else None
        out = {
            "metric": "Msamples/s (paths x spp) on cbox@512spp", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cbox {WIDTH}x{HEIGHT} @ {args.spp} spp per GPU ({spp_total} spp total), diffuse BSDFs, "
                                   f"path integrator (NEE+MIS, RR from depth 4), counter RNG, Gaussian filter, "
                                   f"ordered film resolve included",
                       "parallelism": f"{args.shard[:-1]}-shard x{world} + RCCL film reduce" if world > 1 else "single GPU",
                       "samples_per_step": samples_step, "balance": balance,
                       "segments_per_sample": round(seg / max(smp, 1), 3)},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_launch": round(bytes_per_launch), "avg_launch_ms": round(avg_launch_ms, 4),
                         "per_kernel": per_kernel, "launches": iters, "timed_launches": nl, "timed": "launches of every %s-th sync group" % os.environ.get("MSK_TIMING_EVERY", "1"),
                         "ms_trace_timed": round(ms_trace, 2), "ms_shade_timed": round(ms_shade, 2),
                         "ms_resolve": round(sum(s.ms_resolve for s in stats), 2),
                         "ms_total_device": round(sum(s.ms_total for s in stats), 2)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(abi, hm, flat)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    scene.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
