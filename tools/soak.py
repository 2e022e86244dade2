"""Determinism soak: the same render N times (diffuse cbox and the metal+glass+environment scene), every film must be
bit-identical to the first, and the first to the oracle at a reduced sample count."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
import oracle_binding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ctx = abi.Context(0)
m = hm.cbox_meshes(); del m[3]
m[5].bsdf = {"type": "roughconductor", "alpha": 0.2, "eta": (0.2, 0.92, 1.1), "k": (3.9, 2.45, 2.14), "twosided": True}
blob = hm.blob_mesh("blob", (185, 240, 170), 75, 60, 60, hm.WHITE, seed=3); blob.bsdf = {"type": "roughdielectric", "alpha": 0.1, "int_ior": 1.5, "ext_ior": 1.0}
scenes = {"cbox": hm.cbox_scene(256, 256), "open box, metal, glass blob (7 k triangles), environment": hm.flatten(m + [blob], 192, 160, env={"radiance": (0.25, 0.4, 0.8)})}
for name, flat in scenes.items():
    sc = abi.Scene(ctx, flat)
    o = oracle_binding.load().scene(flat)
    small = abi.render_params(spp=4, seed=3)
    assert np.array_equal(sc.render(small)[0].view(np.uint32), o.render(small, threads=8)[0].view(np.uint32)), name
    prm = abi.render_params(spp=64, seed=1)
    first, st = sc.render(prm)
    t0 = time.time(); bad = 0
    for i in range(n):
        film, _ = sc.render(prm)
        bad += not np.array_equal(film.view(np.uint32), first.view(np.uint32))
    print("%s: %d renders of %d samples, %d differ from the first, %.1f s" % (name, n, st.samples, bad, time.time() - t0))
    assert bad == 0
    sc.close(); o.close()
print("soak ok")
