"""Wall time of msk_gpu_render (render + the film's copy-back into a caller-owned pageable array) against msk_gpu_render_device
(the film stays in HBM) on the bench workload: what the copy-back costs per step.  usage: copyback_time.py [steps]
(MSK_GPU_LIB selects another build of the library, as in tools/config_bench.py)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = abi.Context(0); sc = abi.Scene(ctx, hm.cbox_scene(512, 512)); prm = abi.render_params(spp=512)
host = np.zeros((512, 512, 5), np.float32); dev = torch.zeros((512, 512, 5), dtype=torch.float32, device="cuda")
for _ in range(2):
    sc.render(prm, out=host); sc.render_device(prm, dev.data_ptr())
res = {}
for name, fn in (("render", lambda: sc.render(prm, out=host)), ("render_device", lambda: sc.render_device(prm, dev.data_ptr())), ("render", lambda: sc.render(prm, out=host)),
                 ("render_device", lambda: sc.render_device(prm, dev.data_ptr()))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); res.setdefault(name, []).append((time.perf_counter() - t0) / steps * 1e3)
print({k: [round(x, 3) for x in v] for k, v in res.items()}, "copy-back ms/step:", round(min(res["render"]) - min(res["render_device"]), 3),
      "film equal:", bool(np.array_equal(host, dev.cpu().numpy())))
