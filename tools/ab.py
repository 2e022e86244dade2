"""A/B harness: runs the bench workload (cbox 512^2) in-process under several env configurations, interleaved.
usage: ab.py spp reps "K=V K=V" "K=V ..." ...   (each quoted group = one configuration; a new Scene per run)"""
import importlib, os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    abi = importlib.import_module("misaki-render_amd.abi"); hm = importlib.import_module("misaki-render_amd.hostmirror")
    spp = int(sys.argv[2])
    ctx = abi.Context(0); sc = abi.Scene(ctx, hm.cbox_scene(512, 512))
    sc.render(abi.render_params(spp=8))
    best = None
    for _ in range(3):
        film, st = sc.render(abi.render_params(spp=spp))
        d = st.as_dict()
        if best is None or d["ms_total"] < best["ms_total"]:
            best = d
    print(json.dumps({k: round(best[k], 2) for k in ("ms_total", "ms_trace", "ms_shade", "ms_resolve")}))
    sys.exit(0)
spp, reps, cfgs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
res = {c: [] for c in cfgs}
for r in range(reps):
    for c in cfgs:
        env = dict(os.environ)
        for kv in c.split():
            k, v = kv.split("=")
            env[k] = v
        out = subprocess.run([sys.executable, __file__, "--one", spp], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
        res[c].append(json.loads(out))
for c in cfgs:
    tot = sorted(x["ms_total"] for x in res[c])
    print("%-70s total min %.2f med %.2f | trace %.2f shade %.2f resolve %.2f" % (
        c, tot[0], tot[len(tot) // 2], min(x["ms_trace"] for x in res[c]), min(x["ms_shade"] for x in res[c]), min(x["ms_resolve"] for x in res[c])))
