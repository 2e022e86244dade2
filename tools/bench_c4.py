"""BASELINE config 4: cbox 1920x1080 @ 4096 spp on one GPU (8.5 G samples, ~170 GB of sample records in one pass)."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
w, h, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080, 4096)
flat = hm.cbox_scene(w, h)
ctx = abi.Context(0); sc = abi.Scene(ctx, flat)
for rep in range(2):      # the first render also allocates the record buffers (hipMalloc of ~170 GB takes seconds)
  t0 = time.time(); film, st = sc.render(abi.render_params(spp=spp)); dt = time.time() - t0
  print("render: %.2f s wall, device %.1f ms -> %.1f Msamples/s; trace %.1f shade %.1f resolve %.1f ms; passes=%d iterations=%d" % (
    dt, st.ms_total, st.samples / st.ms_total / 1e3, st.ms_trace, st.ms_shade, st.ms_resolve, st.passes, st.iterations))
img = hm.develop(film)
print("samples", st.samples, "== ", w * h * spp, "finite", bool(np.isfinite(film).all()), "mean rgb", img[..., :3].mean(axis=(0, 1)), "W/spp min/max", film[..., 4].min() / spp, film[..., 4].max() / spp)
