#!/usr/bin/env python3
"""tools/rccl_selfcheck.py : the collectives bench.py's N > 1 path issues — `reduce` of a film-shaped fp32 tensor onto rank 0,
`all_reduce(MAX)` of the step time, `all_gather` of one double per rank (the balance option), `barrier` — through the RCCL back
end ("nccl" on ROCm) on THIS box's one GPU, as a process group of ONE rank.

What it is for: a GPU box of this pool has one GPU, so the two-rank rehearsals of bench.py run over gloo; RCCL itself — library
load, communicator setup with `device_id`, a collective on a HIP stream, the synchronisation multigpu.reduce_film relies on —
has never run on the image otherwise.  A group of one exercises all of that except the transport between peers (xGMI), which no
one-GPU box can.  Prints one JSON line; run ON THE GPU BOX:  python3 tools/rccl_selfcheck.py > gpurun_out/rccl_world1.json"""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29731")
import torch
import torch.distributed as dist

mg = importlib.import_module("misaki-render_amd.multigpu")
torch.cuda.set_device(0)
t0 = time.perf_counter()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
film = torch.rand((512, 512, 5), dtype=torch.float32, device="cuda")
ref = film.clone()
dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)          # what reduce_film issues when there is more than one rank
torch.cuda.current_stream().synchronize()
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
shares, times = mg.speed_proportional_shares(dist, 50.0, 512, device="cuda")
dist.barrier()
torch.cuda.synchronize()
t_first = time.perf_counter() - t0
t1 = time.perf_counter()
for _ in range(20):
    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
    torch.cuda.current_stream().synchronize()
per_reduce_ms = (time.perf_counter() - t1) / 20 * 1e3
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "torch": torch.__version__, "hip": torch.version.hip,
       "nccl_version": list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
       "reduce_identity": bool(torch.equal(film, ref)), "all_reduce_max": float(t.item()), "all_gather_times": times, "shares": shares,
       "init_plus_first_collectives_s": round(t_first, 3), "reduce_5MB_world1_ms": round(per_reduce_ms, 4),
       "device": torch.cuda.get_device_name(0), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
dist.destroy_process_group()
print(json.dumps(out))
sys.exit(0 if out["reduce_identity"] and out["all_reduce_max"] == 1.25 and times == [50.0] else 1)
