"""BASELINE config 5 stand-in: a teapot-class (~146 k triangle) rough-dielectric mesh inside the Cornell box,
1024x1024 @ 1024 spp per node = 128 spp per GPU on 8 GPUs (argv: size spp res).  Reports Msamples/s and the kernel split; checks a small crop against the oracle."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
res = int(sys.argv[3]) if len(sys.argv) > 3 else 270
meshes = hm.cbox_meshes()[:6]          # room + light, no boxes
blob = hm.blob_mesh("teapot_class", (278, 200, 280), 160, res, res, hm.WHITE, seed=7, bump=0.25)
blob.bsdf = {"type": "roughdielectric", "alpha": 0.1, "int_ior": 1.5, "ext_ior": 1.0}
t0 = time.time(); flat = hm.flatten(meshes + [blob], size, size); print("scene: %d triangles, flatten %.1f s" % (flat.desc.n_faces, time.time() - t0))
ctx = abi.Context(0); t0 = time.time(); sc = abi.Scene(ctx, flat); print("scene_create (upload + BVH) %.2f s" % (time.time() - t0))
prm = abi.render_params(spp=spp)
film = None
for i in range(2):
    t0 = time.time(); film, st = sc.render(prm); dt = time.time() - t0
    print("render %d: %.1f ms wall, device %.1f ms -> %.1f Msamples/s; trace %.1f shade %.1f resolve %.1f ms; L=%.2f iterations=%d" % (
        i, dt * 1e3, st.ms_total, st.samples / st.ms_total / 1e3, st.ms_trace, st.ms_shade, st.ms_resolve, st.segments / st.samples, st.iterations))
if "--check" in sys.argv:
    import oracle_binding
    o = oracle_binding.load().scene(flat)
    px = np.array([[size // 2, size // 2], [size // 2 + 37, size // 2 - 20], [size // 3, size // 2]], np.int32)
    p2 = abi.render_params(spp=16)
    a, _ = sc.sample_pixels(p2, px); b, _ = o.sample_pixels(p2, px)
    print("per-sample parity vs oracle:", np.array_equal(a.view(np.uint32), b.view(np.uint32)))
