#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s of the path-tracing hot path on the synthetic
Cornell box at 512x512, 512 spp per GPU (BASELINE.json configs[1]), diffuse BSDFs.

    python bench.py --gpus N --steps K --warmup W

A step = one complete render through the C ABI: wavefront path tracing of every sample, the ordered film
replay, and the film's copy-back to the host (t_render "incl. final film copy-back";
integrator.cpp:43-78 is the reference's timer scope).  Inputs (scene, BVH, tables) are resident in HBM
before the timed region.  N = 1: the step is msk_gpu_render; the same K steps with the film left in HBM
(msk_gpu_render_device) ride along as `value_film_in_hbm`.

N > 1: one process per GPU.  Under torch.distributed.run (WORLD_SIZE set) this process is one rank; from
a bare shell (`python bench.py --gpus 8`) this process only spawns the N ranks — before anything touches
a GPU — waits for them and passes rank 0's JSON line on.  Rank r renders every tile for the sample
indices s = r (mod N) of 512*N spp (`--shard samples`, the default: per-GPU work is constant, weak
scaling) or the spiral blocks id = r (mod N) (`--shard tiles`); the films are summed onto rank 0 with ONE
RCCL reduce and rank 0 copies the sum to the host, all inside the timed region.  `--balance` re-splits
the samples by measured speed after the warm-up (an extra, reported under config.balance; the equal split
is the default).  A node with fewer than N visible GPUs ends the run with one line on stderr before any
rendezvous.  Rank 0 prints ONE JSON line.

`--in-process N`: the other multi-GPU path — ONE process, N devices behind one context of the C ABI
(msk_gpu_init(ids, N): sample shards on member contexts, films summed over peer access by k_film_sum).

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
# Before the HIP runtime starts (nothing has imported torch yet): streams map onto GPU_MAX_HW_QUEUES hardware queues per priority
# level (default 4), and a rank holds torch's and RCCL's streams beside the library's four — streams that share a queue serialise
# (DESIGN.md §6 "Hardware queues": 39.4 vs 32.5 ms per step).  The library's own streams sit at the highest priority, in a pool of
# their own; this is the belt to those braces, and what a caller of MSK_STREAM_PRIORITY=normal needs.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

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
                    help="skip the config 3 / 4 / 5 class renders that N = 1 adds under `other_configs`, and the sharded config 4 / 5 runs of N > 1")
    ap.add_argument("--c4-spp", type=int, default=4096, help="N > 1: total spp of the config-4 run that follows the headline (rehearsals use less)")
    ap.add_argument("--c5-spp", type=int, default=1024, help="N > 1: total spp of the config-5-class run that follows the headline")
    ap.add_argument("--shard", choices=("samples", "tiles"), default="samples",
                    help="N > 1: which axis the ranks split (misaki-render_amd/multigpu.py); both end in one film reduce")
    ap.add_argument("--balance", action="store_true",
                    help="N > 1: after the warm-up, re-split the samples in proportion to the ranks' measured speed (default: equal split)")
    ap.add_argument("--in-process", type=int, default=0, metavar="N",
                    help="one process, N devices behind one msk_ctx (msk_gpu_init(ids, N) + k_film_sum) instead of one process per GPU")
    ap.add_argument("--cpus", type=int, default=0, metavar="C",
                    help="restrict this process (and the ranks it spawns: they share the set) to the first C CPUs of its affinity mask "
                         "before anything touches a GPU — the host-side waits under a CPU quota (tools/cpu_contention.py, DESIGN.md §7)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: launch, rendezvous (gloo), reduce and timing only")
    ap.add_argument("--rccl-world1", action="store_true",
                    help="N = 1 through the N > 1 step: a process group of ONE rank over RCCL (nccl), msk_gpu_render_device + the film "
                         "reduce + rank 0's pinned copy-back, and the sharded config 4 / 5 runs — everything of the multi-GPU step but a "
                         "peer, on a one-GPU box; the line says \"rccl_world1\": true")
    ap.add_argument("--sync-reduce", action="store_true",
                    help="N > 1: every step waits for its own film reduce and copy-back before the next render starts (rounds 1-5); default: "
                         "two films in flight, the reduce + copy-back of step k overlap the render of step k + 1 (multigpu.ReducePipeline)")
    ap.add_argument("--pageable-film", action="store_true",
                    help="the film's copy-back target is ordinary pageable memory (default: pinned, which msk_gpu_render copies into directly)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks (or --in-process members) that ALL render on cuda:0, the ranks reducing over gloo through host "
                         "copies of the film: the whole multi-GPU path but the RCCL call / the peer reads, on a one-GPU box "
                         "(tests/test_bench_launch.py); its line says \"rehearsal\": true and is not a measurement")
    return ap.parse_args()


def build_id():
    """What the committed profiles are matched against (tools/build_id.py): a hash of the GPU library's sources without their
    comments and white space."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_id as _b
    return _b.build_id()


def devices_seen():
    """torch.cuda.device_count() does not initialise a GPU on this image (so the launcher may call it before it spawns)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def launch_ranks(args):
    """`python bench.py --gpus N` from a bare shell: start N ranks as CHILD processes (this process has not touched a
    GPU and never will), one per GPU, rendezvous on 127.0.0.1.  Every child is watched: the first rank that exits with an
    error ends the others (a rank that dies before or inside the rendezvous would otherwise leave the rest waiting for the
    process-group timeout), and that rank's code is the exit code.  Returns the exit code for the shell."""
    import threading
    if not args.dry_run and not args.rehearse_on_one_gpu and devices_seen() < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this node shows {devices_seen()} GPU(s); nothing was launched", file=sys.stderr)
        return 2
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


def _cgroup_cpu_quota():
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota[0] == "max" else round(int(quota[0]) / int(quota[1]), 2)
    except Exception:
        return None


def _thread_cpu():
    """{tid: (comm, cpu seconds)} of this process's threads (/proc/self/task): who burns the host CPU during a step."""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                f = open(f"/proc/self/task/{tid}/stat").read()
                comm = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / tick)
            except Exception:
                pass
    except Exception:
        pass
    return out


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding
    return oracle_binding.load()


def cpu_baseline(abi, hm, flat, threads, seconds_target=12.0):
    """The CPU oracle (a port of the reference's Embree3+TBB loop, see oracle/oracle.cpp) timed on this box's host cores on a
    bounded sample of the headline workload, plus BASELINE config 1 (cbox 256x256 @ 16 spp, the reference's own CPU-runnable
    case) in full.  Reported, never the target."""
    sc = _oracle().scene(flat)
    prm = abi.render_params(spp=1, rng_mode=abi.MSK_RNG_PCG_BLOCK)
    t0 = time.perf_counter()
    sc.render(prm, threads)
    dt = time.perf_counter() - t0
    spp = int(max(1, min(512, seconds_target / max(dt, 1e-3))))
    prm = abi.render_params(spp=spp, rng_mode=abi.MSK_RNG_PCG_BLOCK)
    t0 = time.perf_counter()
    _, st = sc.render(prm, threads)
    dt = time.perf_counter() - t0
    sc.close()
    out = {"value": round(st.samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
           "affinity_cpus": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": _cgroup_cpu_quota(),
           "sample": f"cbox {WIDTH}x{HEIGHT} @ {spp} spp ({st.samples} samples, {dt:.1f} s), pcg_block sampler, "
                     f"own BVH instead of Embree, one 32x32 tile per task", "host_cpus": os.cpu_count()}
    # config 1: the whole job (1 M samples), the faster of two runs
    c1 = _oracle().scene(hm.cbox_scene(256, 256))
    p1 = abi.render_params(spp=16, rng_mode=abi.MSK_RNG_PCG_BLOCK)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        _, s1 = c1.render(p1, threads)
        d1 = time.perf_counter() - t0
        best = d1 if best is None else min(best, d1)
    c1.close()
    out["config1"] = {"value": round(s1.samples / best / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
                      "sample": f"BASELINE config 1 in full: cbox 256x256 @ 16 spp ({s1.samples} samples, {best:.2f} s), pcg_block sampler"}
    return out


def l2_vs_cpu(abi, hm, flat, film_gpu, prm):
    """BASELINE's second half of the metric: per-pixel L2 of the developed image against the CPU path — the oracle renders
    the SAME workload (full size, same counter RNG, same seed) on every host thread, both films are developed as
    HDRFilm::image() does (films/hdrfilm.cpp:48-90: rgb = xyz_to_srgb(XYZ) / W) and compared per pixel."""
    import numpy as np
    sc = _oracle().scene(flat)
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
    """The newest committed profiles/rNN_<tag>.json (tools/profile_mesh.sh + summarize_mesh_profiles.py on one MI355X): per-kernel
    PMC figures of the mesh configs — HBM bytes per segment / per ray and wave64 VALU instructions per ray."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{tag}.json")))
    try:
        return json.load(open(files[-1])) if files else None
    except Exception:
        return None


def other_configs(abi, hm, ctx, skip_cpu=False):
    """BASELINE configs 3, 4 and 5 as one GPU sees them (N = 1 only, after the timed region): config 4 whole on this GPU,
    configs 3 / 5 on the bunny- / teapot-class stand-ins (the reference ships no meshes), config 5 at the 128-spp share
    one GPU of eight renders.  Three renders each: the first allocates the workspace, the faster of the other two is reported.
    Every entry carries its own `roofline`: algorithmic bytes of the path state the two wavefront kernels move (DESIGN.md §5;
    the general shading variant carries 8 more bytes per segment in and out) over the render's device time against the HBM
    peak, and — for the mesh configs — the PMC-measured bytes and VALU instructions of the committed profile scaled to this
    run's segments and rays (labelled as coming from the profile)."""
    import numpy as np
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
            import torch
            film = torch.zeros((flat.desc.film.height, flat.desc.film.width, 5), dtype=torch.float32).pin_memory().numpy()   # pinned, reused
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
            # algorithmic bytes of the two kernels: counted by the library for THIS render (msk_stats::bytes_*, ABI v7)
            b_shade, b_trace = int(st.bytes_shade), int(st.bytes_trace)
            ms_wave = max(st.ms_total - st.ms_resolve, 1e-6)
            gbs = (b_shade + b_trace) / (ms_wave * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_shade_gen || k_trace", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_sample": round((b_shade + b_trace) / max(smp, 1), 1),
                    "note": "state bytes of both wavefront kernels over the wavefront phase's device time; the traversal of a tree in "
                            "HBM/L2 is bound by VALU issue and load latency, not by these bytes (DESIGN.md §9)"}
            l2 = None
            if tag and not skip_cpu:
                # film-level parity at the config's full size (north star: per-pixel L2 vs the CPU path < 1e-4): 16 spp = 16.8 M
                # samples = 8192 regions (the four-part loop), the oracle on every host thread, same counter RNG and seed
                try:
                    p16 = abi.render_params(spp=16, seed=0)
                    f16 = np.zeros(film.shape, np.float32)
                    sc.render(p16, out=f16)
                    osc = _oracle().scene(flat)
                    t0 = time.perf_counter()
                    ref, rst = osc.render(p16, len(os.sched_getaffinity(0)))
                    t_cpu = time.perf_counter() - t0
                    osc.close()
                    a, b = hm.develop(f16)[..., :3].astype(np.float64), hm.develop(ref)[..., :3].astype(np.float64)
                    per_px = np.sqrt(((a - b) ** 2).sum(-1))
                    l2 = {"max": float(per_px.max()), "pixels_gt_1e-4": int((per_px > 1e-4).sum()), "film_bit_identical": bool(np.array_equal(f16, ref)),
                          "spp": 16, "samples": int(rst.samples), "cpu_s": round(t_cpu, 1), "cpu_msamples_per_s": round(rst.samples / t_cpu / 1e6, 3)}
                except Exception as e:
                    l2 = {"error": str(e)[:200]}
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
                roof["profile_stale"] = prof.get("_build_id") != build_id()
            out.append({"workload": name, "tag": tag or "c4_1gpu", "triangles": int(flat.desc.n_faces), "samples": smp,
                        "value": round(st.samples / dt / 1e6, 1), "unit": "Msamples/s", "ms": round(dt * 1e3, 1),
                        "ms_device": round(st.ms_total, 1), "segments_per_sample": round(st.segments / max(st.samples, 1), 3),
                        "iterations": int(st.iterations), "passes": int(st.passes), "ms_resolve": round(st.ms_resolve, 1),
                        "scene_create_s": round(t_scene, 2), "finite": bool(np.isfinite(film).all()),
                        "timing": "wall time of msk_gpu_render incl. the film copy-back (PCIe) into pinned host memory", "roofline": roof, "l2_vs_cpu": l2})
            sc.close()
        except Exception as e:                                       # reported, never fatal for the headline line
            out.append({"workload": name, "tag": tag or "c4_1gpu", "error": str(e)[:300]})
    return out


SHARDED_CONFIGS = (
    # tag, workload, scene maker (hostmirror attribute, arguments), film (w, h), shard, option holding the total spp
    ("c4", "BASELINE config 4: cbox 1920x1080 @ {spp} spp, pixel-tile shard across {n} ranks + one film reduce", ("cbox_scene", (1920, 1080)), (1920, 1080), "tiles", "c4_spp"),
    ("c5", "BASELINE config 5 class: 146 k-triangle rough-dielectric mesh 1024x1024 @ {spp} spp, sample shard across {n} ranks + one film reduce",
     ("teapot_class_scene", (1024,)), (1024, 1024), "samples", "c5_spp"),
)


def sharded_configs(args, abi, hm, mg, ctx, dist, rank, world, rehearsal, fence):
    """N > 1 only, after the headline, EVERY rank: BASELINE configs 4 and 5 as BASELINE words them — config 4 (cbox 1920x1080,
    4096 spp) as a pixel-tile shard (rank r renders the spiral blocks id = r mod N at the full sample count), config 5 class
    (the 146 k-triangle dielectric mesh, 1024 x 1024, 1024 spp) as a sample shard — each ending in the same ONE film reduce
    onto rank 0 and rank 0's copy-back as the headline step.  The total work is the config's (fixed): these are strong-scaling
    points.  One untimed render (workspace, plan), then two timed steps between fences, the maximum over the ranks.  Returns
    scalars for `config` on rank 0 (c4_msamples_per_s, c4_ms, c4_shard, ...), {} elsewhere.  --dry-run: the same sequence on
    host tensors of the films' shapes, no render."""
    import torch
    out = {}
    steps = 2
    for tag, what, (maker, margs), (w, h), shard, opt in SHARDED_CONFIGS:
        spp = getattr(args, opt)
        scene, err = None, None
        try:
            if os.environ.get("MSK_BENCH_TEST_FAIL_CONFIG") == f"{rank}:{tag}":        # tests/test_bench_launch.py: a rank that cannot set a config up
                raise RuntimeError("test: rank %d cannot set %s up" % (rank, tag))
            if args.dry_run:
                film = torch.ones((h, w, 5), dtype=torch.float32)
            else:
                import numpy as np
                flat = getattr(hm, maker)(*margs)
                scene = abi.Scene(ctx, flat)
                prm = mg.shard_params(abi, spp, rank, world, mode=shard, seed=0)
                film = torch.zeros((h, w, 5), dtype=torch.float32, device="cuda")
                host_t = torch.zeros((h, w, 5), dtype=torch.float32)
                if not rehearsal:
                    host_t = host_t.pin_memory()
                host_film = host_t.numpy()
                scene.render_device(prm, film.data_ptr())        # untimed: workspace, plan (no collective: a failure here is local)
        except Exception as e:
            err = str(e)[:200]
        # every rank learns whether every rank is ready BEFORE the first collective of this config: a rank that could not set
        # the config up (out of memory, say) must not leave the others waiting inside a reduce
        ready = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device="cpu" if (rehearsal or args.dry_run) else "cuda")
        dist.all_reduce(ready, op=dist.ReduceOp.MIN)
        if float(ready.item()) < 1.0:
            if rank == 0:
                out[tag + "_error"] = err or "another rank could not set this config up"
            if scene is not None:
                scene.close()
            continue

        def step():
            if args.dry_run:
                film.fill_(1.0)
                mg.reduce_film(film, dist)
                return None
            st = scene.render_device(prm, film.data_ptr())
            if rehearsal:
                host = film.cpu()
                mg.reduce_film(host, dist)
                if rank == 0:
                    host_t.copy_(host)
            else:
                mg.reduce_film(film, dist, force=args.rccl_world1)
                if rank == 0:
                    host_t.copy_(film)
            return st
        fence()
        t0 = time.perf_counter()
        sts = [step() for _ in range(steps)]
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if (rehearsal or args.dry_run) else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if rank == 0:
            samples = w * h * spp
            out[tag + "_msamples_per_s"] = None if args.dry_run else round(samples * steps / dt / 1e6, 1)
            out[tag + "_ms"] = round(dt / steps * 1e3, 2)
            out[tag + "_shard"] = shard
            out[tag + "_spp"] = spp
            out[tag + "_workload"] = what.format(spp=spp, n=world)[:118]
            if args.dry_run:
                out[tag + "_reduce_ok"] = bool((film == float(world)).all())
            else:
                out[tag + "_film_finite"] = bool(np.isfinite(host_film).all())
                out[tag + "_rank0_passes"] = int(sts[-1].passes)
                out[tag + "_weight_per_sample"] = round(float(host_film[..., 4].astype(np.float64).sum()) / samples, 6)
        if scene is not None:
            scene.close()
    return out


def profile_counters():
    """Per-launch counters of the committed rocprofv3 PMC passes (profiles/pmc_summary.json): NOT measured in this run — they
    describe the build the profile was taken on (`_build_id`; bench.py flags `profile_stale` when that is not this build)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
    except Exception:
        return {}


def profile_single_stream(pm):
    """Achieved HBM-side GB/s of the two wavefront kernels when each launch has the GPU to itself, from the COMMITTED profiles:
    PMC bytes per launch (profiles/pmc_summary.json) over the average launch duration of the MSK_STREAMS=1 kernel trace
    (the newest profiles/rNN_kernel_stats_1stream.csv).  Not measured in this run; labelled as such in the line."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_stats_1stream.csv")))
    try:
        rows = list(csv.DictReader(open(files[-1])))
    except Exception:
        return None
    out = {}
    for key in ("k_shade_gen", "k_trace"):
        b = next((v.get("hbm_bytes_per_launch") for k, v in pm.items() if k.startswith(key) and isinstance(v, dict)), None)
        r = next((r for r in rows if ("msk::" + key) in r["Name"]), None)
        if b is None or r is None:
            continue
        us = float(r["AverageNs"]) / 1e3
        gbs = b / (us * 1e-6) / 1e9
        out[key] = {"hbm_bytes_per_launch": round(b), "avg_launch_us": round(us, 1), "achieved": round(gbs, 1), "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4)}
    out["source"] = "committed profiles (%s + rocprofv3 PMC), not this run" % os.path.basename(files[-1])
    return out


def stream_yardstick(roof):
    """What THIS box's HBM gives plain streaming kernels, measured now (misaki-render_amd/lib/msk_hbm_stream, built by
    __graft_entry__.build from tools/micro/hbm_stream.hip; a child process, 0.3 s): read-only, write-only, copy, and `soa7` — a
    kernel of the shading sweep's access shape (one wave per region, 1 KB per step from each of seven arrays and to each of
    seven others, non-temporal) without any arithmetic.  `roofline.peak` stays the guide's 8 TB/s; this is the practical ceiling
    next to it: `shade_vs_soa7` = the shading kernel's single-stream HBM rate of the committed profile / soa7's."""
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "misaki-render_amd", "lib", "msk_hbm_stream")
    if not os.path.exists(exe):
        return None
    try:
        p = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        rates = {}
        for line in p.stdout.splitlines():
            f = line.split()
            if len(f) >= 3 and f[2] == "GB/s" and f[0] not in rates:
                rates[f[0]] = float(f[1])
        if p.returncode != 0 or "soa7" not in rates:
            return {"error": (p.stdout + p.stderr)[-200:]}
        y = {"unit": "GB/s", "read": rates.get("read"), "write": rates.get("write"), "copy": rates.get("copy"), "soa7": rates["soa7"],
             "what": "streaming kernels without arithmetic on this box, measured in this run; soa7 = the shading sweep's access shape"}
        # the ratio mixes a committed profile (another run, maybe another build) with this run's soa7: only when the profile is this build's
        shade = ((roof.get("per_kernel_single_stream_profile") or {}).get("k_shade_gen") or {}).get("achieved")
        if shade and not roof.get("profile_stale"):
            y["shade_vs_soa7"] = round(shade / rates["soa7"], 3)
            y["shade_vs_soa7_source"] = "k_shade_gen's single-stream HBM rate of the committed profile (build %s = this build) / this run's soa7" % roof.get("profile_build_id")
        elif shade:
            y["shade_vs_soa7"] = None
            y["shade_vs_soa7_source"] = "omitted: the committed profile was taken on build %s, this is %s" % (roof.get("profile_build_id"), roof.get("build_id"))
        return y
    except Exception as e:                      # never let the yardstick cost the bench line
        return {"error": str(e)[:200]}


def roofline(stats, args):
    """`roofline` of the JSON line from the HIP events the library records around its launches on its own streams.
    Algorithmic bytes of the SoA path state each kernel must move; a "segment" is a slot that is live after a
    shading sweep (written by it, traced, read by the next sweep), a "shadow ray" one that carries a shadow ray:
      k_trace_q    48 B/segment (ray_o, ray_d in; hit out) + 32 B/shadow ray (ray_o, sh in: the shadow queue reads the origin too)
      k_shade_gen  176 B/segment (id 8 B, wl, thr, res, ray_d, hit in; id 8 B, wl, thr, res, ray_o, ray_d out)
                   - 64 B/sample (a new camera sample's thr = 1 and res = 0 are neither written nor read)
                   + 48 B/shadow ray (contrib in; sh, contrib out) + 20 B/sample (record out)"""
    seg = sum(s.segments for s in stats)
    smp = sum(s.samples for s in stats)
    shd = sum(s.shadow_rays for s in stats)
    ms_trace = sum(s.ms_trace for s in stats)
    ms_shade = sum(s.ms_shade for s in stats)
    n_trace = sum(s.n_trace_launches for s in stats)                 # launches that carried timing events
    n_shade = sum(s.n_shade_launches for s in stats)
    l_trace = sum(s.launches_trace for s in stats)                   # launches actually made
    l_shade = sum(s.launches_shade for s in stats)
    l_wave = sum(s.launches_wavefront for s in stats)
    # counted by the library for these very renders (msk_stats::bytes_shade / bytes_trace, ABI v7: what its kernels' launches were
    # asked to move, from its own layout) — the formulas of the docstring, evaluated where the layout lives
    bytes_trace = sum(s.bytes_trace for s in stats)
    bytes_shade = sum(s.bytes_shade for s in stats)
    bytes_agree = bytes_trace == seg * 48 + shd * 32 and bytes_shade == seg * 176 - smp * 64 + shd * 48 + smp * 20       # DESIGN.md §5, restated here
    # the two kernels take nearly the same time on this workload and which one is ahead depends on the box; the shading
    # kernel (five times the bytes) is reported unless the traversal kernel is clearly the longer one
    if ms_trace > 1.03 * ms_shade:
        name, b, ms, nt, nl = "k_trace", bytes_trace, ms_trace, n_trace, l_trace
    else:
        name, b, ms, nt, nl = "k_shade_gen", bytes_shade, ms_shade, n_shade, l_shade
    # per launch: the kernel's bytes over the launches that moved them (its own + the k_wavefront launches of the thin end of
    # a pass, which run both kernels' code); the average duration is over the timed launches
    bytes_per_launch = b / max(nl + l_wave, 1)
    avg_launch_ms = ms / max(nt, 1)
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    ms_wavefront = sum(s.ms_total - s.ms_resolve for s in stats)
    pm = profile_counters()
    stale = pm.get("_build_id") != build_id()

    def prof(prefix, key):
        return next((v.get(key) for k, v in pm.items() if k.startswith(prefix) and isinstance(v, dict)), None)
    n_streams = int(os.environ.get("MSK_STREAMS", "4"))
    per_kernel = {"kernel": name, "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
                  "bytes_per_launch": round(bytes_per_launch), "avg_launch_ms": round(avg_launch_ms, 4),
                  "launches": {"k_shade_gen": l_shade, "k_trace": l_trace, "k_wavefront": l_wave},
                  "avg_launch_ms_shade": round(ms_shade / max(n_shade, 1), 4), "avg_launch_ms_trace": round(ms_trace / max(n_trace, 1), 4)}
    if n_streams > 1:
        # The wavefront loop runs on several streams at once: at any moment a few launches of BOTH kernels
        # share the GPU, a launch's wall duration says how long it shared, not how fast it could go, and the unit that
        # has a bandwidth is the phase: both kernels' algorithmic bytes of one whole-pool iteration over the wall time
        # such an iteration takes (every stream launches its own part of the pool: n_streams kernel pairs = one iteration).
        name = "k_shade_gen || k_trace (%d streams)" % n_streams
        per_iter = max((l_shade + l_wave) / n_streams, 1.0)
        bytes_per_launch = (bytes_trace + bytes_shade) / per_iter
        avg_launch_ms = ms_wavefront / per_iter
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
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
    return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "profile_stale": bool(stale), "profile_build_id": pm.get("_build_id"), "build_id": build_id(),
            "traffic_source": "NOT measured in this run: per-launch FETCH_SIZE/WRITE_SIZE of the committed rocprofv3 PMC "
                              "passes (%s)" % pm.get("_source", "profiles/pmc_summary.json"),
            "bytes_per_launch": round(bytes_per_launch), "avg_launch_ms": round(avg_launch_ms, 4),
            "per_kernel": per_kernel,
            "timed_launches": nt, "timed": "launches of every %s-th sync group" % os.environ.get("MSK_TIMING_EVERY", "1"),
            "ms_trace_timed": round(ms_trace, 2), "ms_shade_timed": round(ms_shade, 2),
            "ms_resolve": round(sum(s.ms_resolve for s in stats), 2),
            "ms_total_device": round(sum(s.ms_total for s in stats), 2),
            "segments_per_sample": round(seg / max(smp, 1), 3),
            # the two per-sample figures side by side: this layout's (msk_stats::bytes_*, + 40 B/sample of film replay input) and
            # SURVEY 8(d)'s canonical 150 + 428 L for L segments per sample.  The canonical layout carries a 76-byte path state and
            # a 32-byte ray in AND out of every shade, a 20-byte hit record and a 48-byte shadow item per segment; this one
            # keeps id in 8 bytes, no rng counter / eta / pdf fields (recomputed or folded into the ray), writes no thr / res for a
            # fresh sample and moves shadow data only for slots that carry a shadow ray — about half the bytes for the same path.
            "bytes_per_sample": {"this_layout": round((bytes_shade + bytes_trace) / max(smp, 1) + 40.0, 1),
                                 "survey_canonical_150_plus_428L": round(150.0 + 428.0 * seg / max(smp, 1), 1),
                                 "source": "msk_stats::bytes_shade + bytes_trace of this run (+ 40 B/sample the film replay reads)",
                                 "library_count_equals_design_formula": bool(bytes_agree)},
            "frac_if_canonical_bytes": round((150.0 + 428.0 * seg / max(smp, 1)) * smp / max(ms_wavefront * 1e-3, 1e-9) / 1e9 / HBM_PEAK_GBS, 4),
            "valu": valu, "per_kernel_single_stream_profile": profile_single_stream(pm)}


def main():
    args = parse_args()
    if args.cpus > 0 and "WORLD_SIZE" not in os.environ:         # (a spawned rank inherits the launcher's set)
        os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:args.cpus])
    if args.in_process and args.gpus > 1:
        print("bench.py: --in-process N and --gpus N are two different multi-GPU paths; give one", file=sys.stderr)
        sys.exit(2)
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
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's world size wins", file=sys.stderr)
    dev = "cpu" if args.dry_run else "cuda"
    rehearsal = bool(args.rehearse_on_one_gpu) and not args.dry_run
    members = max(args.in_process, 1)                    # devices behind this process's ONE context
    n_dev = devices_seen()
    need = max(world, members)
    if not args.dry_run and not rehearsal and n_dev < need:
        # before any rendezvous and before anything touches a GPU: the other ranks find the same and leave the same way
        print(f"bench.py: {need} GPU(s) asked for but this node shows {n_dev}; rank {rank} stops", file=sys.stderr)
        sys.exit(2)
    if rehearsal:
        local_rank = 0                                   # every rank on the one GPU there is
    if not args.dry_run:
        torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or (args.rccl_world1 and not args.dry_run)       # the N > 1 step (render_device + reduce + rank 0's copy-back)
    if multi:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29733")
        if args.dry_run or rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    spp_total = mg.weak_scaling_spp(args.spp, max(world, members))
    host_film = None
    if args.dry_run:
        film = torch.ones((8, 8, 5), dtype=torch.float32)
        abi = hm = scene = ctx = flat = None

        pipe = None
        if dist is not None and not args.sync_reduce:
            films2 = [torch.ones((8, 8, 5), dtype=torch.float32) for _ in range(2)]
            pipe = mg.ReducePipeline(films2, [torch.zeros((8, 8, 5), dtype=torch.float32) for _ in range(2)], dist, rank)

        def step():
            if pipe is not None:
                f = pipe.begin()
                f.fill_(1.0)
                pipe.submit()
                film.copy_(pipe.last_host_film() if rank == 0 else f)
                return None
            film.fill_(1.0)
            mg.reduce_film(film, dist)
            return None
        step_hbm = None
    else:
        import numpy as np
        abi = importlib.import_module("misaki-render_amd.abi")
        hm = importlib.import_module("misaki-render_amd.hostmirror")
        flat = hm.cbox_scene(WIDTH, HEIGHT)
        ctx = abi.Context([0] * members if rehearsal and members > 1 else list(range(members)) if members > 1 else local_rank)
        scene = abi.Scene(ctx, flat)
        # --in-process: the library itself shards the call's samples over the members
        prm = mg.shard_params(abi, spp_total, rank, world, mode=args.shard, seed=0) if multi else abi.render_params(spp=spp_total, seed=0)
        film = torch.zeros((HEIGHT, WIDTH, 5), dtype=torch.float32, device="cuda")
        # rank 0's copy-back target: caller-owned PINNED host memory (what a host application that wants its film fast allocates).
        # N = 1: msk_gpu_render sends the DMA straight into a pinned target (0.2 ms for 5 MB; a pageable one goes through the
        # library's staging buffer + one memcpy, ~0.8 ms: --pageable-film measures that).  N > 1: torch copies the reduced film into
        # it, and a pageable target makes that copy 2-3 ms of a 31 ms step (a tenth of the weak-scaling budget)
        host_t = torch.zeros((HEIGHT, WIDTH, 5), dtype=torch.float32)
        if not rehearsal and not args.pageable_film:
            host_t = host_t.pin_memory()
        host_film = host_t.numpy()

        def step_hbm():
            return scene.render_device(prm, film.data_ptr())

        pipe = None
        if multi and not rehearsal and not args.sync_reduce:
            # two films in flight: the reduce + rank 0's copy-back of step k overlap the render of step k + 1
            films2 = [film, torch.zeros_like(film)]
            hosts2 = [host_t, torch.zeros((HEIGHT, WIDTH, 5), dtype=torch.float32).pin_memory()] if rank == 0 else [None, None]
            pipe = mg.ReducePipeline(films2, hosts2, dist, rank, force=args.rccl_world1)
        if not multi:
            def step():
                return scene.render(prm, out=host_film)[1]           # msk_gpu_render: the render + the film's copy-back
        elif pipe is not None:
            def step():
                f = pipe.begin()                              # (waits for this slot's reduce / copy-back of two steps ago)
                t_r = time.perf_counter()
                st = scene.render_device(prm, f.data_ptr())
                own["render_s"] += time.perf_counter() - t_r
                pipe.submit()
                return st
        else:
            def step():
                # render_device returns when the film is complete on the library's stream; reduce_film returns when the
                # reduce has finished reading it (multigpu.reduce_film synchronises): the next render may overwrite it
                t_r = time.perf_counter()
                st = scene.render_device(prm, film.data_ptr())
                own["render_s"] += time.perf_counter() - t_r          # this rank's own render, without waiting for anybody
                if rehearsal:                                # gloo has no device reduce: through the host (rehearsal only)
                    host = film.cpu()
                    mg.reduce_film(host, dist)
                    if rank == 0:
                        host_t.copy_(host)
                else:
                    mg.reduce_film(film, dist, force=args.rccl_world1)
                    if rank == 0:
                        host_t.copy_(film)                   # the reduced film's copy-back (synchronous)
                return st

    own = {"render_s": 0.0}

    def fence():
        if not args.dry_run:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if not args.dry_run:
                torch.cuda.synchronize()

    def timed(fn, k):
        fence()
        t0 = time.perf_counter()
        sts = [fn() for _ in range(k)]
        if pipe is not None and fn is step:
            pipe.drain()                  # every step's reduced film has arrived on rank 0's host: inside the timed region
        fence()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if (rehearsal or args.dry_run) else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, sts

    warm = [step() for _ in range(args.warmup)]
    if pipe is not None:
        pipe.drain()
    balance = None
    if dist is not None and args.shard == "samples" and args.balance and warm and not args.dry_run:
        # Opt-in extra: the GPUs of a node are not equally fast on this workload and an equal split waits for
        # the slowest.  Every rank's kernel time of the last warm-up step decides speed-proportional contiguous sample ranges;
        # the per-GPU average stays args.spp.  Not the constant-work weak scaling the default reports.
        shares, times = mg.speed_proportional_shares(dist, warm[-1].ms_trace + warm[-1].ms_shade, spp_total,
                                                     device="cpu" if rehearsal else "cuda")
        if shares is not None:
            prm = mg.shard_params(abi, spp_total, rank, world, mode="range", shares=shares, seed=0)
            step()                       # untimed: the new shares' plan and record buffers are set up here
            if pipe is not None:
                pipe.drain()
            balance = {"equal_split_kernel_ms": [round(t, 2) for t in times], "spp_shares": shares}
    import resource
    ru0, th0 = resource.getrusage(resource.RUSAGE_SELF), _thread_cpu()
    own["render_s"] = 0.0
    dt, stats = timed(step, args.steps)
    # N > 1: every rank's own render time per step (the renders alone, no reduce, no waiting): the spread between the node's GPUs —
    # an equal split waits for the slowest — apart from what the reduce and rank 0's copy-back cost
    rank_render_ms = None
    if dist is not None and multi and not args.dry_run:
        mine = torch.tensor([own["render_s"] / max(args.steps, 1) * 1e3], dtype=torch.float64, device="cpu" if rehearsal else dev)
        gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, mine)
        rank_render_ms = [round(float(x.item()), 3) for x in gathered]
    ru1, th1 = resource.getrusage(resource.RUSAGE_SELF), _thread_cpu()
    by_thread = sorted(((round(c - th0.get(t, (n, 0.0))[1], 3), n, t == os.getpid()) for t, (n, c) in th1.items()), reverse=True)
    # host CPU this rank's process spent inside the timed region (every thread: the library's loop, its waits, the runtime's helpers)
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    host_side = {"cpus_allowed": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": _cgroup_cpu_quota(),
                 "cpu_s_per_step": round(cpu_s / max(args.steps, 1), 5), "cpus_busy": round(cpu_s / max(dt, 1e-9), 3),
                 "wait": os.environ.get("MSK_WAIT", "library default"), "host_threads": os.environ.get("MSK_HOST_THREADS", "library default"),
                 "busiest_threads": [f"{n}{' (main)' if m else ''}: {c} s" for c, n, m in by_thread[:4] if c > 0],
                 "what": "user + system CPU time of rank 0's process over the timed region / its wall time"}

    # N > 1: the sharded config 4 / 5 runs — every rank takes part; rank 0 first puts the headline on stderr (below)
    sharded = {}
    run_sharded = (world > 1 or multi) and not args.no_other_configs
    if run_sharded and (rank != 0 or args.dry_run):
        sharded = sharded_configs(args, abi, hm, mg, ctx, dist, rank, world, rehearsal, fence)
    if rank == 0 and args.dry_run:
        ok = bool((film == float(world)).all())
        print(json.dumps({"metric": "Msamples/s (paths x spp) on cbox@512spp", "value": None, "unit": "Msamples/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "dry_run": True, "reduce_ok": ok,
                          "config": {"workload": "none (launch / rendezvous / reduce only)", "samples_per_step": WIDTH * HEIGHT * spp_total,
                                     "shard": args.shard if world > 1 else "none",
                                     "rccl_ranks": dist.get_world_size() if dist is not None else 1, "devices_seen": n_dev,
                                     "balanced": balance is not None, "reduce": "pipelined: two films in flight" if pipe is not None else "per step", **sharded}, "balance": balance}),
              flush=True)
    elif not args.dry_run:
        # the same K steps with the film left in HBM (every rank takes part: the steps hold collectives)
        dt_hbm, _ = timed(step_hbm, args.steps) if world == 1 else (None, None)       # (N = 1, also through --rccl-world1)
    if rank == 0 and not args.dry_run:
        import numpy as np
        n_gpus = max(world, members)
        samples_step = WIDTH * HEIGHT * spp_total              # all GPUs together
        value = samples_step * args.steps / dt / 1e6
        par = "single GPU" + (", through the N > 1 step: RCCL group of one rank, film reduce + pinned copy-back" if multi else "")
        if world > 1:
            par = f"{args.shard[:-1]}-shard x{world}, one process per GPU + " + ("gloo film reduce through host copies (rehearsal on one GPU)" if rehearsal else "RCCL film reduce")
        elif members > 1:
            par = f"sample-shard x{members} behind one context (msk_gpu_init(ids, {members})), films summed by k_film_sum over peer access"
        out = {
            "metric": "Msamples/s (paths x spp) on cbox@512spp", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            **({"rccl_world1": True} if (multi and world == 1) else {}),
            **({"rehearsal": True, "film_finite": bool(np.isfinite(host_film).all()), "film_weight_sum": float(host_film[..., 4].astype(np.float64).sum())} if rehearsal else {}),
            # (`config` holds scalars only: the driver's parser drops nested values; `workload` stays under 120 characters)
            "config": {"workload": f"cbox {WIDTH}x{HEIGHT} @ {args.spp} spp/GPU, diffuse, path (NEE+MIS, RR), counter RNG, incl. film resolve + film copy-back",
                       "timing_scope": "incl_copyback", "film_target": "pageable" if args.pageable_film else "pinned", "spp_total": spp_total,
                       "reduce": ("pipelined: two films in flight" if pipe is not None else "per step" if multi else "none"),
                       "parallelism": par, "shard": (args.shard if world > 1 else "samples" if members > 1 else "none"), "samples_per_step": samples_step,
                       "rccl_ranks": dist.get_world_size() if dist is not None else 1, "in_process_members": members,
                       "devices_seen": n_dev, "balanced": balance is not None,
                       **({"rank_render_ms_min": min(rank_render_ms), "rank_render_ms_max": max(rank_render_ms),
                           "step_minus_slowest_render_ms": round(dt / args.steps * 1e3 - max(rank_render_ms), 3)} if rank_render_ms else {})},
            # what the step's clock covers (integrator.cpp:43-78 is the reference's scope): since round 4 `value` includes the film's
            # copy-back to the host (rounds 1-3: film left in HBM = `value_film_in_hbm`, the figure to compare across rounds)
            "timing_scope": "incl_copyback", "balance": balance, "host_side": host_side, "rank_render_ms": rank_render_ms,
            "roofline": roofline(stats, args),
        }
        if dt_hbm is not None:
            out["ms_per_step_film_in_hbm"] = round(dt_hbm / args.steps * 1e3, 3)
            out["value_film_in_hbm"] = round(samples_step * args.steps / dt_hbm / 1e6, 2)
        if n_gpus == 1 and not rehearsal:
            out["roofline"]["stream_yardstick"] = stream_yardstick(out["roofline"])
        # the measured headline is safe on stderr before anything slower or riskier runs (CPU baselines, other configs)
        print("bench.py headline (extras follow on stdout): " + json.dumps(out), file=sys.stderr, flush=True)
        if run_sharded:
            sharded = sharded_configs(args, abi, hm, mg, ctx, dist, rank, world, rehearsal, fence)
        compact = {}                    # scalars that ride in `config`: the driver's parser keeps `config`, `roofline` and `cpu_baseline`
        if not args.no_cpu_baseline and n_gpus == 1:
            try:
                out["l2_vs_cpu"] = l2_vs_cpu(abi, hm, flat, host_film, prm)
                compact["l2_max"] = out["l2_vs_cpu"]["max"]
            except Exception as e:
                out["l2_vs_cpu"] = {"error": str(e)[:300]}
            # the reference CLI caps TBB at 8 threads (main.cpp:43-44): that run is `cpu_baseline` (+ config 1 under `config1`);
            # the same port on every hardware thread of this host rides along (SURVEY §8d asks for both)
            out["cpu_baseline"] = cpu_baseline(abi, hm, flat, min(8, os.cpu_count() or 1))
            compact["cpu_config1_msamples_per_s"] = out["cpu_baseline"]["config1"]["value"]
            # every hardware thread of this host: the full-size counter-RNG render l2_vs_cpu just timed (same port, same work
            # per sample but for the sampler).  The cores that can actually run are the cgroup's quota, not the thread count.
            l2 = out.get("l2_vs_cpu") or {}
            if l2.get("cpu_msamples_per_s"):
                quota = _cgroup_cpu_quota()
                out["cpu_baseline_all_threads"] = {"value": l2["cpu_msamples_per_s"], "unit": "Msamples/s",
                                                   "cores": min(l2["cpu_threads"], int(quota)) if quota else l2["cpu_threads"],
                                                   "threads": l2["cpu_threads"], "cgroup_cpu_quota": quota, "kind": "port",
                                                   "sample": l2["cpu"], "host_cpus": os.cpu_count()}
        else:
            out["cpu_baseline"] = None
        if n_gpus == 1 and not args.no_other_configs and not multi:
            out["other_configs"] = other_configs(abi, hm, ctx, skip_cpu=args.no_cpu_baseline)
            for e in out["other_configs"]:
                if "value" in e:
                    compact[e["tag"] + "_msamples_per_s"] = e["value"]
                if (e.get("l2_vs_cpu") or {}).get("max") is not None:
                    compact[e["tag"] + "_l2_max"] = e["l2_vs_cpu"]["max"]
        out["config"].update(compact)       # c3 / c5 / c4_1gpu _msamples_per_s, l2_max, cpu_config1_msamples_per_s
        out["config"].update(sharded)       # N > 1: c4_* / c5_* of the sharded config 4 / 5 runs
        if "cpu_baseline_all_threads" in out:
            out["config"]["cpu_all_threads_msamples_per_s"] = out["cpu_baseline_all_threads"]["value"]
        print(json.dumps(out), flush=True)
    if scene is not None:
        scene.close()
        ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
