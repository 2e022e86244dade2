"""tools/dump_rays.py c3|c5 [spp] [iteration] — the GPU's own rays for the host-side scheduler model (tools/micro/sched_model.cpp):
renders the config with one loop on one stream and MSK_DUMP_RAYS set, so that the library writes the live slots of every 64th region
as the traversal launch of that iteration is about to read them; writes gpurun_out/<cfg>_rays.bin and <cfg>_pos.bin (9 floats per
triangle, scene-global order)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
it = sys.argv[3] if len(sys.argv) > 3 else "12"
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
os.environ.update(MSK_STREAMS="1", MSK_DUMP_RAYS=os.path.join(out, which + "_rays.bin"), MSK_DUMP_ITER=it, MSK_DUMP_STRIDE=os.environ.get("MSK_DUMP_STRIDE", "64"))
import numpy as np
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
flat = hm.bunny_class_scene(1024) if which == "c3" else hm.teapot_class_scene(1024)
pos = flat.vertices[:, :3]
tri = np.empty((flat.desc.n_faces, 9), np.float32)
for m in range(flat.desc.n_meshes):
    md = flat.desc.meshes[m]
    f = flat.faces[md.first_face:md.first_face + md.face_count].astype(np.int64) + md.first_vertex
    tri[md.first_face:md.first_face + md.face_count] = pos[f].reshape(-1, 9)
tri.tofile(os.path.join(out, which + "_pos.bin"))
ctx = abi.Context(0); sc = abi.Scene(ctx, flat)
film, st = sc.render(abi.render_params(spp=spp))
print(f"{which}: {st.samples} samples, {st.iterations} iterations; dumped iteration {it} ->", os.path.getsize(os.environ["MSK_DUMP_RAYS"]), "bytes")
