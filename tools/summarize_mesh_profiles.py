#!/usr/bin/env python3
"""tools/summarize_mesh_profiles.py <tag> [c3 c5 ...]: turns gpurun_out/pm_<cfg>_* (tools/profile_mesh.sh) into
profiles/<tag>_<cfg>.json (+ the two kernel-stats csv files): per kernel launch counts and average durations (default four
loops and MSK_STREAMS=1), HBM bytes (read = 2*1024*FETCH_SIZE, write = 1024*WRITE_SIZE: MI355X_MICROARCH.md), SQ fractions of
SQ_WAVE_CYCLES, VALU instructions per ray, L2 hit rate, and the derived roofline figures."""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
cfgs = sys.argv[2:] or ["c3", "c5"]
out = os.path.join(root, "profiles")
HBM_PEAK, VALU_PEAK = 8000.0, 256 * 4 * 2.4 / 2.0


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from build_id import build_id          # hash of the GPU library's sources (bench.py flags `profile_stale` when it differs)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("msk::", "")


def latest(sub, pat):
    f = glob.glob(os.path.join(root, "gpurun_out", sub, "**", pat), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def stats_line(log):
    try:
        return eval(re.findall(r"^\{.*\}$", open(os.path.join(root, "gpurun_out", log)).read(), re.M)[-1])
    except Exception:
        return {}


def counters(sub):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    f = latest(sub, "*_counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            tot[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return tot


def durations(sub):
    """per kernel: launches and total ns of the kernel trace of a PMC pass (so bytes and time come from one run)"""
    d = collections.defaultdict(lambda: [0, 0.0])
    f = latest(sub, "*_kernel_trace.csv")
    if f:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            d[k][0] += 1
            d[k][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return d


head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
for cfg in cfgs:
    p = "pm_" + cfg
    res = {"_source": f"tools/profile_mesh.sh {cfg} on one MI355X; tree at or after commit {head}", "_build_id": build_id(), "kernels": {}}
    for sub, name in (("_kt", "kernel_stats"), ("_kt1", "kernel_stats_1stream")):
        f = latest(p + sub, "*_kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(out, f"{tag}_{cfg}_{name}.csv"))
            res["render" + sub] = stats_line(p + sub + ".log")
            for r in csv.DictReader(open(f)):
                k = short(r["Name"])
                if k.startswith("k_"):
                    res["kernels"].setdefault(k, {})["launches" + sub] = int(r["Calls"])
                    res["kernels"][k]["avg_us" + sub] = round(float(r["AverageNs"]) / 1e3, 1)
                    res["kernels"][k]["total_ms" + sub] = round(float(r["TotalDurationNs"]) / 1e6, 2)
    st = stats_line(p + "_fetch.log")
    res["render_pmc"] = st
    rays = st.get("segments", 0) + st.get("shadow_rays", 0)
    fetch, write, sq, sq2, l2 = counters(p + "_fetch"), counters(p + "_write"), counters(p + "_sq"), counters(p + "_sq2"), counters(p + "_l2")
    dur = durations(p + "_fetch")
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        e = res["kernels"].setdefault(k, {})
        rd, wr = 2 * 1024 * fetch[k].get("FETCH_SIZE", 0.0), 1024 * write[k].get("WRITE_SIZE", 0.0)
        n, ns = dur[k]
        e.update({"pmc_launches": n, "pmc_total_ms": round(ns / 1e6, 2), "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                  "hbm_gbs_pmc_run": round((rd + wr) / max(ns, 1), 1), "hbm_frac": round((rd + wr) / max(ns, 1) / HBM_PEAK, 4)})
        v = dict(sq[k]); v.update(sq2[k])
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            e["sq"] = {c + "/WAVE_CYCLES": round(x / wc, 4) for c, x in v.items() if c not in ("SQ_WAVE_CYCLES", "SQ_WAVES")}
            e["sq"]["SQ_WAVES"] = v.get("SQ_WAVES")
            e["valu_insts"] = v.get("SQ_INSTS_VALU")
            nsq, nssq = durations(p + "_sq")[k]
            if nssq:
                e["valu_ginst_per_s_sq_run"] = round(v.get("SQ_INSTS_VALU", 0.0) / nssq, 1)
                e["valu_frac"] = round(v.get("SQ_INSTS_VALU", 0.0) / nssq / VALU_PEAK, 4)
        if k.startswith("k_trace") and rays:
            e["valu_insts_per_ray"] = round(v.get("SQ_INSTS_VALU", 0.0) / rays, 2)
            e["hbm_bytes_per_ray"] = round((rd + wr) / rays, 1)
            e["grays_per_s_pmc_run"] = round(rays / max(ns, 1), 3)
        if k.startswith("k_shade") and st.get("segments"):
            e["valu_insts_per_segment"] = round(v.get("SQ_INSTS_VALU", 0.0) / st["segments"], 2)
            e["hbm_bytes_per_segment"] = round((rd + wr) / st["segments"], 1)
        h, m = l2[k].get("TCC_HIT_sum", 0.0), l2[k].get("TCC_MISS_sum", 0.0)
        if h + m:
            e["l2_hit_rate"] = round(h / (h + m), 4)
            e["l2_requests"] = l2[k].get("TCC_REQ_sum")
    json.dump(res, open(os.path.join(out, f"{tag}_{cfg}.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))
