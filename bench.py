#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s of the path-tracing hot path on the synthetic
Cornell box at 512x512, 512 spp per GPU (BASELINE.json configs[1]), diffuse BSDFs.

    python bench.py --gpus N --steps K --warmup W

A step = one complete render through the C ABI (msk_gpu_render_device): wavefront path tracing
of every sample + the ordered film replay, film left in HBM.  Inputs (scene, BVH) are resident
in HBM before the timed region.

N > 1: one process per GPU.  Under torch.distributed.run (WORLD_SIZE set) this process is one rank;
from a bare shell (`python bench.py --gpus 8`) this process only spawns the N ranks — before anything
touches a GPU — waits for them and passes rank 0's JSON line on.  Rank r renders every tile for the
sample indices s = r (mod N) of 512*N spp (`--shard samples`, the default; speed-proportional contiguous
ranges when the GPUs differ) or the spiral blocks id = r (mod N) (`--shard tiles`): per-GPU work is
constant (weak scaling), and the films are summed onto rank 0 with one RCCL reduce inside the timed region.
Rank 0 prints ONE JSON line.

`--dry-run` exercises the launch / rendezvous / reduce / timing plumbing without a GPU (gloo backend, a
host tensor as the film, no render): what tests/test_bench_launch.py runs on CPU.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH = HEIGHT = 512
SPP_PER_GPU = 512
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
# VALU issue peak: 256 CUs x 4 SIMD-32, one wave64 instruction per 2 cycles per SIMD at 2.4 GHz (same guide)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP_PER_GPU, help="spp per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the config 3 / 4 / 5 class renders that N = 1 adds under `other_configs`")
    ap.add_argument("--shard", choices=("samples", "tiles"), default="samples",
                    help="N > 1: which axis the ranks split (misaki-render_amd/multigpu.py); both end in one film reduce")
    ap.add_argument("--no-balance", action="store_true",
                    help="N > 1: keep the equal interleaved sample split even when the GPUs differ in speed")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: launch, rendezvous (gloo), reduce and timing only")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks that ALL render on cuda:0 and reduce over gloo through host copies of the film: the whole "
                         "multi-process path but the RCCL call, on a one-GPU box (tests/test_bench_launch.py); its line says "
                         "\"rehearsal\": true and is not a measurement")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` from a bare shell: start N ranks as CHILD processes (this process has not touched a
    GPU and never will), one per GPU, rendezvous on 127.0.0.1.  Every child is watched: the first rank that exits with an
    error ends the others (a rank that dies before or inside the rendezvous would otherwise leave the rest waiting for the
    process-group timeout), and that rank's code is the exit code.  Returns the exit code for the shell."""
    import threading
    # the port stays bound (SO_REUSEADDR on both sides) until just before the children start, which narrows the window in
    # which another process can take it; a rendezvous failure ends every rank through the watch loop below
    sock = socket.socket()
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)     # drain rank 0's pipe meanwhile
    reader.start()
    rc = 0
    pending = set(range(args.gpus))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = abs(code)
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
                deadline = time.time() + 10
                for q in pending:
                    try:
                        procs[q].wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        procs[q].kill()
        if pending:
            time.sleep(0.05)
    reader.join(timeout=5)
    sys.stdout.write("".join(x for x in out if x))
    sys.stdout.flush()
    return rc


def cpu_baseline(abi, hm, flat, threads, seconds_target=12.0):
    """The CPU oracle (a port of the reference's Embree3+TBB loop, see oracle/oracle.cpp) timed on
    this box's host cores on a bounded sample of the same workload.  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding
    orc = oracle_binding.load()
    sc = orc.scene(flat)
    spp = 1
    prm = abi.render_params(spp=spp, rng_mode=abi.MSK_RNG_PCG_BLOCK)
    t0 = time.perf_counter()
    sc.render(prm, threads)
    dt = time.perf_counter() - t0
    spp = int(max(1, min(512, seconds_target / max(dt, 1e-3))))
    prm = abi.render_params(spp=spp, rng_mode=abi.MSK_RNG_PCG_BLOCK)
    t0 = time.perf_counter()
    _, st = sc.render(prm, threads)
    dt = time.perf_counter() - t0
    sc.close()
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if quota[0] == "max" else round(int(quota[0]) / int(quota[1]), 2)
    except Exception:
        quota = None
    return {"value": round(st.samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "affinity_cpus": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": quota,
            "sample": f"cbox {WIDTH}x{HEIGHT} @ {spp} spp ({st.samples} samples, {dt:.1f} s), pcg_block sampler, "
                      f"own BVH instead of Embree, one 32x32 tile per task", "host_cpus": os.cpu_count()}


def l2_vs_cpu(abi, hm, flat, film_gpu, prm):
    """BASELINE's second half of the metric: per-pixel L2 of the developed image against the CPU path — the oracle renders
    the SAME workload (full size, same counter RNG, same seed) on every host thread, both films are developed as
    HDRFilm::image() does (films/hdrfilm.cpp:48-90: rgb = xyz_to_srgb(XYZ) / W) and compared per pixel."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding
    sc = oracle_binding.load().scene(flat)
    threads = len(os.sched_getaffinity(0))
    t0 = time.perf_counter()
    film_cpu, st = sc.render(prm, threads)
    dt = time.perf_counter() - t0
    sc.close()
    a, b = hm.develop(film_gpu)[..., :3].astype(np.float64), hm.develop(film_cpu)[..., :3].astype(np.float64)
    l2 = np.sqrt(((a - b) ** 2).sum(-1))
    return {"max": float(l2.max()), "rmse": float(np.sqrt((l2 ** 2).mean())), "pixels_gt_1e-4": int((l2 > 1e-4).sum()),
            "pixels": int(l2.size), "film_bit_identical": bool(np.array_equal(film_gpu, film_cpu)),
            "cpu": f"oracle, counter RNG, seed {prm.seed}, {threads} threads, cbox {WIDTH}x{HEIGHT} @ {prm.spp} spp = {st.samples} samples in {dt:.1f} s "
                   f"({st.samples / dt / 1e6:.2f} Msamples/s)", "cpu_msamples_per_s": round(st.samples / dt / 1e6, 4), "cpu_threads": threads,
            "on": "developed linear-sRGB pixels (HDRFilm::image)"}


def mesh_profile(tag):
    """profiles/r03_<tag>.json (tools/profile_mesh.sh + summarize_mesh_profiles.py on one MI355X): per-kernel PMC figures of the
    mesh configs — HBM bytes per segment / per ray and wave64 VALU instructions per ray of the committed profile."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", f"r03_{tag}.json")))
    except Exception:
        return None


def other_configs(abi, hm, ctx):
    """BASELINE configs 3, 4 and 5 as one GPU sees them (N = 1 only, after the timed region): config 4 whole on this GPU,
    configs 3 / 5 on the bunny- / teapot-class stand-ins (the reference ships no meshes), config 5 at the 128-spp share
    one GPU of eight renders.  Three renders each: the first allocates the workspace, the faster of the other two is reported.
    Every entry carries its own `roofline`: algorithmic bytes of the path state the two wavefront kernels move (DESIGN.md §5;
    the general shading variant carries 8 more bytes per segment in and out) over the render's device time against the HBM
    peak, and — for the mesh configs — the PMC-measured bytes and VALU instructions of the committed profile scaled to this
    run's segments and rays (labelled as coming from the profile)."""
    out = []
    jobs = [
        ("config 3 class: 70 k-triangle rough-conductor mesh in the Cornell room, 1024x1024 @ 256 spp",
         lambda: hm.bunny_class_scene(1024), 256, "c3"),
        ("config 5 class: 146 k-triangle rough-dielectric mesh in the Cornell room, 1024x1024 @ 128 spp (one GPU's share of 1024 spp on 8)",
         lambda: hm.teapot_class_scene(1024), 128, "c5"),
        ("config 4 on ONE GPU: cbox 1920x1080 @ 4096 spp", lambda: hm.cbox_scene(1920, 1080), 4096, None),
    ]
    for name, make, spp, tag in jobs:
        try:
            flat = make()
            t0 = time.perf_counter()
            sc = abi.Scene(ctx, flat)
            t_scene = time.perf_counter() - t0
            prm = abi.render_params(spp=spp)
            import numpy as np
            film = np.zeros((flat.desc.film.height, flat.desc.film.width, 5), np.float32)      # reused: its pages stay mapped
            sc.render(prm, out=film)                              # allocates the workspace, uploads the plan
            dt, st = None, None
            for _ in range(2):                                    # the faster of two is reported (a host hiccup is not the GPU's)
                t0 = time.perf_counter()
                _, st_k = sc.render(prm, out=film)
                dt_k = time.perf_counter() - t0
                if dt is None or dt_k < dt:
                    dt, st = dt_k, st_k
            general = tag is not None                         # non-diffuse BSDFs: the general shading variant (aux in the state)
            seg, smp, shd = int(st.segments), int(st.samples), int(st.shadow_rays)
            b_shade = seg * (176 + (16 if general else 0)) - smp * 64 + shd * 48 + smp * 20
            b_trace = seg * 48 + shd * (16 if general else 32)        # k_trace_r reads ray_o once per slot; k_trace_q once per queue
            ms_wave = max(st.ms_total - st.ms_resolve, 1e-6)
            gbs = (b_shade + b_trace) / (ms_wave * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_shade_gen || k_trace", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_sample": round((b_shade + b_trace) / max(smp, 1), 1),
                    "note": "state bytes of both wavefront kernels over the wavefront phase's device time; the traversal of a tree in "
                            "HBM/L2 is bound by load issue and latency, not by these bytes (DESIGN.md §9)"}
            prof = mesh_profile(tag) if tag else None
            if prof:
                k = prof.get("kernels", {})
                tr = next((v for n, v in k.items() if n.startswith("k_trace")), {})
                sh = next((v for n, v in k.items() if n.startswith("k_shade")), {})
                rays = seg + shd
                if tr.get("hbm_bytes_per_ray") and sh.get("hbm_bytes_per_segment"):
                    t_gbs = (tr["hbm_bytes_per_ray"] * rays + sh["hbm_bytes_per_segment"] * seg) / (ms_wave * 1e-3) / 1e9
                    roof["traffic_gbs_from_profile"] = round(t_gbs, 1)
                    roof["traffic_frac_from_profile"] = round(t_gbs / HBM_PEAK_GBS, 4)
                if tr.get("valu_insts_per_ray") and sh.get("valu_insts_per_segment"):
                    ginst = (tr["valu_insts_per_ray"] * rays + sh["valu_insts_per_segment"] * seg) / (ms_wave * 1e-3) / 1e9
                    roof["valu"] = {"achieved": round(ginst, 1), "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instr/s",
                                    "frac": round(ginst / VALU_PEAK_GINST, 4),
                                    "note": "peak = 2 cycles per wave64 instruction (v_fma / v_add class); min / max / cmp / cndmask / "
                                            "shifts / conversions take 4 (tools/micro/valu_ops.hip), so full issue is reached well below 1"}
                if tr.get("l2_hit_rate") is not None:
                    roof["l2_hit_rate_trace_from_profile"] = tr["l2_hit_rate"]
                roof["profile_source"] = prof.get("_source")
            out.append({"workload": name, "triangles": int(flat.desc.n_faces), "samples": smp,
                        "value": round(st.samples / dt / 1e6, 1), "unit": "Msamples/s", "ms": round(dt * 1e3, 1),
                        "ms_device": round(st.ms_total, 1), "segments_per_sample": round(st.segments / max(st.samples, 1), 3),
                        "iterations": int(st.iterations), "passes": int(st.passes), "ms_resolve": round(st.ms_resolve, 1),
                        "scene_create_s": round(t_scene, 2), "finite": bool(__import__("numpy").isfinite(film).all()),
                        "timing": "wall time of msk_gpu_render incl. the film copy-back (PCIe)", "roofline": roof})
            sc.close()
        except Exception as e:                                       # reported, never fatal for the headline line
            out.append({"workload": name, "error": str(e)[:300]})
    return out


def profile_counters():
    """Per-launch counters of the committed rocprofv3 PMC passes (profiles/pmc_summary.json, profiles/*_sq.json):
    NOT measured in this run — they describe the build and configuration the profile was taken on."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
    except Exception:
        pm = {}
    return pm


def profile_single_stream():
    """Achieved HBM-side GB/s of the two wavefront kernels when each launch has the GPU to itself, from the COMMITTED profiles:
    PMC bytes per launch (profiles/pmc_summary.json) over the average launch duration of the MSK_STREAMS=1 kernel trace
    (profiles/r03_kernel_stats_1stream.csv).  Not measured in this run; labelled as such in the line."""
    import csv
    pm = profile_counters()
    out = {}
    try:
        rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r03_kernel_stats_1stream.csv"))))
    except Exception:
        return None
    for key in ("k_shade_gen", "k_trace"):
        b = next((v.get("hbm_bytes_per_launch") for k, v in pm.items() if k.startswith(key) and isinstance(v, dict)), None)
        r = next((r for r in rows if ("msk::" + key) in r["Name"]), None)
        if b is None or r is None:
            continue
        us = float(r["AverageNs"]) / 1e3
        gbs = b / (us * 1e-6) / 1e9
        out[key] = {"hbm_bytes_per_launch": round(b), "avg_launch_us": round(us, 1), "achieved": round(gbs, 1), "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4)}
    out["source"] = "committed profiles (rocprofv3 PMC + MSK_STREAMS=1 kernel trace), not this run"
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    # Per-kernel HIP-event timing on every 4th sync group (8 iterations each), rotating over the steps: a timed dispatch costs
    # ~6 us, 2 % of a step if every launch carries events.  The roofline's average launch duration is the
    # average over the timed launches.
    os.environ.setdefault("MSK_TIMING_EVERY", "4")
    import torch
    mg = importlib.import_module("misaki-render_amd.multigpu")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MSK_BENCH_TEST_FAIL_RANK") == str(rank):      # tests/test_bench_launch.py: a rank that dies before the rendezvous
        sys.exit(3)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's world size wins", file=sys.stderr)
    dev = "cpu" if args.dry_run else "cuda"
    rehearsal = bool(args.rehearse_on_one_gpu) and not args.dry_run
    if rehearsal:
        local_rank = 0                                   # every rank on the one GPU there is
    if not args.dry_run:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run or rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    spp_total = mg.weak_scaling_spp(args.spp, world)
    if args.dry_run:
        film = torch.ones((8, 8, 5), dtype=torch.float32)
        abi = hm = scene = ctx = flat = None

        def step():
            film.fill_(1.0)
            mg.reduce_film(film, dist)
            return None
    else:
        abi = importlib.import_module("misaki-render_amd.abi")
        hm = importlib.import_module("misaki-render_amd.hostmirror")
        flat = hm.cbox_scene(WIDTH, HEIGHT)
        ctx = abi.Context(local_rank)
        scene = abi.Scene(ctx, flat)
        prm = mg.shard_params(abi, spp_total, rank, world, mode=args.shard, seed=0)
        film = torch.zeros((HEIGHT, WIDTH, 5), dtype=torch.float32, device="cuda")

        def step():
            # render_device returns when the film is complete on the library's stream; reduce_film returns when the
            # reduce has finished reading it (multigpu.reduce_film synchronises): the next render may overwrite it
            st = scene.render_device(prm, film.data_ptr())
            if rehearsal and dist is not None:           # gloo has no device reduce: through the host (rehearsal only)
                host = film.cpu()
                mg.reduce_film(host, dist)
                film.copy_(host)
            else:
                mg.reduce_film(film, dist)
            return st

    def fence():
        if not args.dry_run:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if not args.dry_run:
                torch.cuda.synchronize()

    warm = [step() for _ in range(args.warmup)]
    balance = None
    if dist is not None and args.shard == "samples" and not args.no_balance and warm and not args.dry_run:
        # The GPUs of a node are not equally fast on this workload (the shading kernel differs by up to 20 %
        # between boxes of the pool) and an equal split waits for the slowest.  Every rank's device time of the last
        # warm-up step decides speed-proportional contiguous sample ranges; the per-GPU average stays args.spp.
        # (kernel time of the wavefront launches that carried events — the same sync groups on every rank — not the step's
        # wall or device total, which on a first step also holds the one-off uploads of the render plan)
        shares, times = mg.speed_proportional_shares(dist, warm[-1].ms_trace + warm[-1].ms_shade, spp_total,
                                                     device="cpu" if rehearsal else "cuda")
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
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0 and args.dry_run:
        ok = bool((film == float(world)).all())
        print(json.dumps({"metric": "Msamples/s (paths x spp) on cbox@512spp", "value": None, "unit": "Msamples/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "dry_run": True, "reduce_ok": ok,
                          "config": {"workload": "none (launch / rendezvous / reduce only)", "samples_per_step": WIDTH * HEIGHT * spp_total}}),
              flush=True)
    elif rank == 0:
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
        # algorithmic bytes of the SoA path state each kernel must move; a "segment" is a slot that is live
        # after a shading sweep (written by it, traced, read by the next sweep), a "shadow ray" one that carries a shadow ray:
        #   k_trace_q    48 B/segment (ray_o, ray_d in; hit out) + 32 B/shadow ray (ray_o, sh in: the shadow queue reads the origin too)
        #   k_shade_gen  176 B/segment (id 8 B, wl, thr, res, ray_d, hit in; id 8 B, wl, thr, res, ray_o, ray_d out)
        #                - 64 B/sample (a new camera sample's thr = 1 and res = 0 are neither written nor read)
        #                + 48 B/shadow ray (contrib in; sh, contrib out) + 20 B/sample (record out)
        bytes_trace = seg * 48 + shd * 32
        bytes_shade = seg * 176 - smp * 64 + shd * 48 + smp * 20
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
        pm = profile_counters()

        def prof(prefix, key):
            return next((v.get(key) for k, v in pm.items() if k.startswith(prefix) and isinstance(v, dict)), None)
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
            # (the thin end of a pass runs as k_wavefront launches that are booked as 16 iterations each: where the committed
            # single-stream profile of this workload says how many whole-pool launches a step is, that count is used, so that
            # `bytes_per_launch` and `traffic` describe the same unit)
            prof_launches = prof("k_shade_gen", "launches")
            if prof_launches and args.spp == SPP_PER_GPU:
                per_iter = int(prof_launches) * args.steps
            bytes_per_launch = (bytes_trace + bytes_shade) / per_iter
            avg_launch_ms = ms_wavefront / per_iter
            ts, tt = prof("k_shade_gen", "hbm_bytes_per_launch"), prof("k_trace", "hbm_bytes_per_launch")
            # the PMC passes run single-stream (tools/profile_all.sh): one launch of each kernel there IS one whole-pool iteration
            traffic = round(ts + tt) if ts is not None and tt is not None else None
        else:
            t = prof(name, "hbm_bytes_per_launch")
            traffic = round(t) if t is not None else None
        # VALU side (the kernels are issue-bound, not HBM-bound — DESIGN.md §8): wave-level VALU instructions per path
        # segment from the committed SQ counter pass, times this run's segments, over the wavefront phase's wall time
        valu = None
        vs, vt = prof("k_shade_gen", "valu_insts_per_segment"), prof("k_trace", "valu_insts_per_segment")
        if vs is not None and vt is not None and ms_wavefront > 0:
            ginst = (vs + vt) * seg / (ms_wavefront * 1e-3) / 1e9
            valu = {"bound": "valu", "achieved": round(ginst, 1), "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instr/s",
                    "frac": round(ginst / VALU_PEAK_GINST, 4), "insts_per_segment": {"k_shade_gen": vs, "k_trace": vt},
                    "source": pm.get("_source", "profiles/pmc_summary.json"),
                    "note": "wave64 VALU instructions (SQ_INSTS_VALU of the committed profile / its segments) x this run's segments "
                            "/ wavefront-phase wall time; peak = 1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction — the rate of "
                            "the v_fma / v_add / v_mul / v_and class only: v_min / v_max / v_cmp / v_cndmask / shifts / conversions / "
                            "packed and fp64 operations issue at 4 cycles, transcendentals at 8 (measured: tools/micro/valu_ops.hip, "
                            "profiles/r03_valu_ops.txt), so an instruction stream of this mix saturates issue at a frac of about 0.6"}
        out = {
            "metric": "Msamples/s (paths x spp) on cbox@512spp", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            **({"rehearsal": True, "film_finite": bool(torch.isfinite(film).all()), "film_weight_sum": float(film[..., 4].double().sum())} if rehearsal else {}),
            "config": {"workload": f"cbox {WIDTH}x{HEIGHT} @ {args.spp} spp per GPU ({spp_total} spp total), diffuse BSDFs, "
                                   f"path integrator (NEE+MIS, RR from depth 4), counter RNG, Gaussian filter, "
                                   f"ordered film resolve included",
                       "parallelism": f"{args.shard[:-1]}-shard x{world} + RCCL film reduce" if world > 1 else "single GPU",
                       "samples_per_step": samples_step, "balance": balance,
                       "segments_per_sample": round(seg / max(smp, 1), 3)},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         "traffic_source": "NOT measured in this run: per-launch FETCH_SIZE/WRITE_SIZE of the committed rocprofv3 PMC "
                                           "passes (%s)" % pm.get("_source", "profiles/pmc_summary.json"),
                         "bytes_per_launch": round(bytes_per_launch), "avg_launch_ms": round(avg_launch_ms, 4),
                         "per_kernel": per_kernel, "launches": iters,
                         "launches_note": "wavefront iterations as the library books them: an upper bound on kernel-pair launches (a k_wavefront "
                                          "launch of the thin end of a pass is booked as 16 iterations even when its regions empty earlier)",
                         "timed_launches": nl, "timed": "launches of every %s-th sync group" % os.environ.get("MSK_TIMING_EVERY", "1"),
                         "ms_trace_timed": round(ms_trace, 2), "ms_shade_timed": round(ms_shade, 2),
                         "ms_resolve": round(sum(s.ms_resolve for s in stats), 2),
                         "ms_total_device": round(sum(s.ms_total for s in stats), 2),
                         "valu": valu, "per_kernel_single_stream_profile": profile_single_stream()},
        }
        if world == 1:
            # SURVEY 8(d) defines t_render "incl. final film copy-back": the same K steps through msk_gpu_render (host film,
            # 5 MB over PCIe per step); `value` above keeps the film in HBM, as the multi-GPU reduce needs it
            import numpy as np
            film_host = np.zeros((HEIGHT, WIDTH, 5), np.float32)
            scene.render(prm, out=film_host)                  # (maps the array's pages, allocates the library's pinned staging buffer)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                scene.render(prm, out=film_host)
            dt_host = time.perf_counter() - t1
            out["ms_per_step_incl_copyback"] = round(dt_host / args.steps * 1e3, 3)
            out["value_incl_copyback"] = round(samples_step * args.steps / dt_host / 1e6, 2)
        # the measured headline is safe on stderr before anything slower or riskier runs (CPU baselines, other configs)
        print("bench.py headline (extras follow on stdout): " + json.dumps(out), file=sys.stderr, flush=True)
        if not args.no_cpu_baseline and world == 1:
            try:
                out["l2_vs_cpu"] = l2_vs_cpu(abi, hm, flat, film_host, prm)
            except Exception as e:
                out["l2_vs_cpu"] = {"error": str(e)[:300]}
            # the reference CLI caps TBB at 8 threads (main.cpp:43-44): that run is `cpu_baseline`; the same port on every
            # hardware thread of this host rides along (SURVEY §8d asks for both)
            out["cpu_baseline"] = cpu_baseline(abi, hm, flat, min(8, os.cpu_count() or 1))
            # every hardware thread of this host: the full-size counter-RNG render l2_vs_cpu just timed (same port, same work
            # per sample but for the sampler; rendering it a second time with the PCG sampler would add 20 s for the same figure)
            l2 = out.get("l2_vs_cpu") or {}
            if l2.get("cpu_msamples_per_s"):
                out["cpu_baseline_all_threads"] = {"value": l2["cpu_msamples_per_s"], "unit": "Msamples/s", "cores": l2["cpu_threads"], "kind": "port",
                                                   "sample": l2["cpu"], "host_cpus": os.cpu_count()}
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_other_configs:
            out["other_configs"] = other_configs(abi, hm, ctx)
        print(json.dumps(out), flush=True)
    if scene is not None:
        scene.close()
        ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
