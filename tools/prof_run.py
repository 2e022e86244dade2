"""Small driver for rocprofv3 runs: one render of the bench workload at a chosen spp."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
flat = hm.cbox_scene(512, 512)
ctx = abi.Context(0); sc = abi.Scene(ctx, flat)
for _ in range(reps):
    film, st = sc.render(abi.render_params(spp=spp))
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.as_dict().items()})
