"""Driver for rocprofv3 runs of the mesh configs: tools/prof_mesh.py c3|c5 [spp] [reps] [size]
c3 = 70 k-triangle rough-conductor mesh in the Cornell room, c5 = 146 k-triangle rough-dielectric mesh (hostmirror's
bunny_class_scene / teapot_class_scene, what bench.py's other_configs renders).  Prints the render statistics."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
size = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
flat = hm.bunny_class_scene(size) if which == "c3" else hm.teapot_class_scene(size)
ctx = abi.Context(0); sc = abi.Scene(ctx, flat)
for _ in range(reps):
    film, st = sc.render(abi.render_params(spp=spp))
d = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.as_dict().items()}
d["config"] = which; d["triangles"] = int(flat.desc.n_faces); d["size"] = size; d["spp"] = spp
d["msamples_per_s"] = round(st.samples / st.ms_total / 1e3, 1)
print(d)
