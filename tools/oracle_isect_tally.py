"""How far the oracle's two own rules are from a plain Moeller-Trumbore / first-found intersector: renders
BASELINE config 2, the config-3-class and the config-5-class scene on the CPU oracle and prints, per 1e9 rays, how many
Moeller-Trumbore-accepted hits the D10 bounds predicate rejects, how many rays it can have changed, and how many closest-hit
answers the "smallest t, then smallest prim" rule decided.  usage: tools/oracle_isect_tally.py [spp] [threads] [c2 c3 c5]
(test infrastructure: runs the oracle only; no GPU)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding
abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
threads = int(sys.argv[2]) if len(sys.argv) > 2 else len(os.sched_getaffinity(0))
which = sys.argv[3:] or ["c2", "c3", "c5"]
orc = oracle_binding.load()
scenes = {"c2": lambda: hm.cbox_scene(512, 512), "c3": lambda: hm.bunny_class_scene(1024), "c5": lambda: hm.teapot_class_scene(1024)}
for name in which:
    sc = orc.scene(scenes[name]())
    orc.isect_counters()
    t0 = time.time()
    _, st = sc.render(abi.render_params(spp=spp), threads)
    c = orc.isect_counters()
    sc.close()
    rays = c["closest_rays"] + c["any_rays"]
    per = lambda k: round(c[k] * 1e9 / max(rays, 1), 1)
    print(json.dumps({"config": name, "spp": spp, "samples": int(st.samples), "rays": rays, "seconds": round(time.time() - t0, 1), "counts": c,
                      "per_1e9_rays": {k: per(k) for k in ("d10_rejects", "closest_rays_d10_could_change", "any_rays_unoccluded_with_d10_reject",
                                                          "closest_rays_tie_decided", "equal_t_pairs")},
                      "d10_rejects_per_mt_accept": c["d10_rejects"] / max(c["mt_accepts"], 1)}), flush=True)
