import importlib, importlib.util, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
spec = importlib.util.spec_from_file_location("fuzz_rays", os.path.join(ROOT, "tools", "fuzz_rays.py")); fr = importlib.util.module_from_spec(spec); spec.loader.exec_module(fr)
abi, hm, fz = fr.abi, fr.hm, fr.fz
import oracle_binding
s = int(sys.argv[1])
rng = np.random.RandomState(s); flat = fz.random_scene(rng); d = flat.desc
tris = np.array([[flat.vertices[d.meshes[m].first_vertex + i, :3] for i in flat.faces[f]] for m in range(d.n_meshes)
                 for f in range(d.meshes[m].first_face, d.meshes[m].first_face + d.meshes[m].face_count)], np.float32)
rays = fr.adversarial_rays(rng, tris, 20000)
o = oracle_binding.load().scene(flat)
hb = o.trace_closest(rays).copy(); ab = o.trace_any(rays).copy()
o.set_bvh(0); h0 = o.trace_closest(rays); a0 = o.trace_any(rays)
print("oracle bvh vs brute: closest differ", int((hb.view(np.uint32) != h0.view(np.uint32)).any(-1).sum()), "any differ", int((ab != a0).sum()))
try:
    ctx = abi.Context(0); g = abi.Scene(ctx, flat)
    hg = g.trace_closest(rays); ag = g.trace_any(rays)
    bad = np.nonzero((hg.view(np.uint32) != h0.view(np.uint32)).any(-1) | (ag != a0))[0]
    print("gpu vs brute differ on rays", bad.tolist(), "faces", d.n_faces)
    for k in bad[:3]:
        print(" ray", [float(x) for x in rays[k]], "gpu", hg[k], ag[k], "brute", h0[k], a0[k], "prim", h0[k, 3:].view(np.uint32))
        p = int(h0[k, 3:].view(np.uint32)[0]); print(" tri", tris[p].tolist())
except Exception as e:
    print("no gpu", e)
# float64 truth for the differing rays
try:
    for k in bad[:3]:
        p = int(h0[k, 3:].view(np.uint32)[0]); t3 = tris[p].astype(np.float64)
        o3, d3 = rays[k, :3].astype(np.float64), rays[k, 4:7].astype(np.float64)
        e1, e2 = t3[1] - t3[0], t3[2] - t3[0]; n = np.cross(e1, e2)
        den = d3 @ n; tt = ((t3[0] - o3) @ n) / den; q = o3 + tt * d3
        uv = np.linalg.lstsq(np.array([e1, e2]).T, q - t3[0], rcond=None)[0]
        print(" float64: t", tt, "uv", uv, "sin(angle to plane)", abs(den) / np.linalg.norm(n), "hit point fp32-t", (o3 + float(h0[k, 0]) * d3).tolist(), "bbox", t3.min(0).tolist(), t3.max(0).tolist())
except Exception as e:
    print(e)
