import importlib.util, importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
abi, hm = fz.abi, fz.hm
import oracle_binding
s = int(sys.argv[1])
rng = np.random.RandomState(s); flat = fz.random_scene(rng); kw = fz.random_params(rng)
d = flat.desc
print("meshes", d.n_meshes, "faces", d.n_faces, "emitters", [(d.emitters[i].type, d.emitters[i].mesh_id) for i in range(d.n_emitters)], "bsdf types", [d.bsdfs[i].type for i in range(d.n_bsdfs)], kw)
ctx = abi.Context(0); g = abi.Scene(ctx, flat); o = oracle_binding.load().scene(flat)
film, st = g.render(abi.render_params(**kw)); ref, rst = o.render(abi.render_params(**kw), threads=8)
bad = (film.view(np.uint32) != ref.view(np.uint32)).any(-1)
print("pixels differing:", int(bad.sum()), np.argwhere(bad)[:6].tolist(), "segments", st.segments, rst.segments)
ys, xs = np.nonzero(bad)
for y, x in list(zip(ys, xs))[:3]:
    print((y, x), film[y, x], ref[y, x])
# per-sample comparison on the differing pixels and their neighbours
px = np.array([[x + dx, y + dy] for y, x in list(zip(ys, xs))[:4] for dx in (-2, -1, 0, 1, 2) for dy in (-2, -1, 0, 1, 2) if 0 <= x + dx < 48 and 0 <= y + dy < 40], np.int32)
kw2 = {k: v for k, v in kw.items() if k != "block_size"}
a, pa = g.sample_pixels(abi.render_params(**kw2), px); b, pb = o.sample_pixels(abi.render_params(**kw2), px)
diff = (a.view(np.uint32) != b.view(np.uint32)).any(-1)
print("samples differing:", int(diff.sum()), "of", diff.size)
for i, sidx in np.argwhere(diff)[:5]:
    print(px[i], sidx, a[i, sidx], b[i, sidx])
