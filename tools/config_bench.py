"""One script for the non-headline BASELINE configs (replaces round 3's bench_c3 / bench_c4 / bench_c5 / ab_c3 / ab_c5):

  tools/config_bench.py c3|c4|c5|cbox [--spp N] [--size S] [--reps R] [--check]
      renders the config R times in this process and prints one line per render
      (c3 = 70 k-triangle rough-conductor mesh, 1024^2 x 256 spp; c5 = 146 k-triangle rough-dielectric mesh, 1024^2 x 128 spp;
       c4 = cbox 1920x1080 x 4096 spp on one GPU; cbox = the headline scene 512^2 x 512 spp; c5d = c5's geometry with a diffuse
       mesh: with MSK_FORCE_GENERAL_SHADE=0/1 the same all-diffuse scene through either shading variant); --check compares three
      pixels' samples with the oracle bit for bit
  tools/config_bench.py c5 --reps 3 --ab "MSK_TREETOP=0" "MSK_TREETOP=128" "MSK_GPU_LIB=gpurun_scratch/libmsk_gpu_x.so"
      A/B: every quoted group of K=V is one configuration, run as its own process, interleaved R times; prints the minimum
      device time and the trace / shade / resolve sums per configuration
"""
import argparse, importlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
DEFAULTS = {"c3": (1024, 1024, 256), "c5": (1024, 1024, 128), "c5d": (1024, 1024, 128), "c4": (1920, 1080, 4096), "cbox": (512, 512, 512)}


def run_once(a):
    import numpy as np
    abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
    w, h, spp = DEFAULTS[a.config]
    if a.size: w = h = a.size
    if a.spp: spp = a.spp
    t0 = time.time()
    flat = hm.bunny_class_scene(w) if a.config == "c3" else hm.teapot_class_scene(w, diffuse=a.config == "c5d") if a.config in ("c5", "c5d") else hm.cbox_scene(w, h)
    t1 = time.time(); ctx = abi.Context(0); sc = abi.Scene(ctx, flat); t2 = time.time()
    out = {"config": a.config, "triangles": int(flat.desc.n_faces), "size": [w, h], "spp": spp, "flatten_s": round(t1 - t0, 2), "scene_create_s": round(t2 - t1, 3), "renders": []}
    for i in range(a.reps):
        t0 = time.time(); film, st = sc.render(abi.render_params(spp=spp)); dt = time.time() - t0
        r = {"wall_ms": round(dt * 1e3, 1), "ms_total": round(st.ms_total, 2), "msamples_per_s": round(st.samples / st.ms_total / 1e3, 1), "ms_trace": round(st.ms_trace, 1),
             "ms_shade": round(st.ms_shade, 1), "ms_resolve": round(st.ms_resolve, 2), "segments_per_sample": round(st.segments / st.samples, 3),
             "shadow_rays_per_sample": round(st.shadow_rays / st.samples, 3), "iterations": int(st.iterations), "passes": int(st.passes)}
        out["renders"].append(r)
        if not a.json: print("render %d: %s" % (i, r), flush=True)
    out["finite"] = bool(np.isfinite(film).all())
    if a.check:
        import oracle_binding
        o = oracle_binding.load().scene(flat)
        px = np.array([[w // 2, h // 2], [w // 2 + 37, h // 2 - 20], [w // 3, h // 2]], np.int32)
        p2 = abi.render_params(spp=16)
        x, _ = sc.sample_pixels(p2, px); y, _ = o.sample_pixels(p2, px)
        out["per_sample_parity"] = bool(np.array_equal(x.view(np.uint32), y.view(np.uint32)))
    print(json.dumps(out) if a.json else {k: v for k, v in out.items() if k != "renders"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=sorted(DEFAULTS))
    ap.add_argument("--spp", type=int, default=0); ap.add_argument("--size", type=int, default=0); ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--check", action="store_true"); ap.add_argument("--json", action="store_true")
    ap.add_argument("--ab", nargs="+", default=None); ap.add_argument("--inner", type=int, default=3, help="renders per process in an A/B (the minimum counts)")
    a = ap.parse_args()
    if not a.ab:
        return run_once(a)
    res = {c: [] for c in a.ab}
    for r in range(a.reps):
        for c in a.ab:
            env = dict(os.environ)
            for kv in c.split():
                k, v = kv.split("=", 1); env[k] = v
            cmd = [sys.executable, os.path.abspath(__file__), a.config, "--reps", str(a.inner), "--json"] + (["--spp", str(a.spp)] if a.spp else []) + (["--size", str(a.size)] if a.size else [])
            p = subprocess.run(cmd, env=env, capture_output=True, text=True)
            if p.returncode:
                print("FAILED:", c, p.stderr[-400:]); continue
            d = json.loads(p.stdout.strip().split("\n")[-1])
            res[c].append(min(d["renders"], key=lambda x: x["ms_total"]))
    for c in a.ab:
        if not res[c]: continue
        b = min(res[c], key=lambda x: x["ms_total"])
        print("%-64s device min %.1f ms (%.0f Msamples/s) | trace %.1f shade %.1f resolve %.1f | all: %s" % (
            c, b["ms_total"], b["msamples_per_s"], min(x["ms_trace"] for x in res[c]), min(x["ms_shade"] for x in res[c]), b["ms_resolve"],
            " ".join("%.1f" % x["ms_total"] for x in res[c])), flush=True)


if __name__ == "__main__":
    main()
