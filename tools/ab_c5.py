"""A/B on the config-5-class scene (see ab_c3.py)."""
import os, sys, subprocess, re
spp, reps, cfgs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
here = os.path.dirname(os.path.abspath(__file__))
res = {c: [] for c in cfgs}
for r in range(reps):
    for c in cfgs:
        env = dict(os.environ)
        for kv in c.split():
            k, v = kv.split("=")
            env[k] = v
        out = subprocess.run([sys.executable, os.path.join(here, "bench_c5.py"), "1024", spp, "270"], env=env, capture_output=True, text=True).stdout
        m = re.findall(r"device ([\d.]+) ms .*?trace ([\d.]+) shade ([\d.]+)", out)
        res[c].append(min((float(a), float(b), float(d)) for a, b, d in m))
for c in cfgs:
    print("%-60s device min %.1f | trace %.1f shade %.1f" % (c, min(x[0] for x in res[c]), min(x[1] for x in res[c]), min(x[2] for x in res[c])))
