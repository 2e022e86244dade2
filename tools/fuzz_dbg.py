import importlib.util, importlib, sys, os, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
abi, hm = fz.abi, fz.hm
import oracle_binding
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 20657
px = np.array([[int(sys.argv[2]), int(sys.argv[3])]], np.int32) if len(sys.argv) > 3 else np.array([[31, 23]], np.int32)
rng = np.random.RandomState(seed); flat = fz.random_scene(rng); kw = fz.random_params(rng)
orc = oracle_binding.load(); o = orc.scene(flat)
p = abi.render_params(spp=1, seed=kw["seed"], rr_depth=kw["rr_depth"], max_depth=kw["max_depth"], hide_emitters=kw["hide_emitters"])
orc.lib.msk_oracle_set_trace(1); b, _ = o.sample_pixels(p, px); orc.lib.msk_oracle_set_trace(0)
print("oracle", b)
if abi.__dict__.get("Context"):
    try:
        ctx = abi.Context(0); g = abi.Scene(ctx, flat)
        a, _ = g.sample_pixels(p, px); print("gpu   ", a)
        if len(sys.argv) > 4:   # a ray given as 8 hex words
            ray = np.array([float(h) for h in sys.argv[4:12]], np.float32).reshape(1, 8)
            print("ray", ray, "gpu any", g.trace_any(ray), "oracle any", o.trace_any(ray), "gpu closest", g.trace_closest(ray), "oracle closest", o.trace_closest(ray))
    except Exception as e:
        print("no gpu:", e)
