#!/usr/bin/env python3
"""Turns rocprofv3 output under gpurun_out/ into the small summaries committed under profiles/.

  tools/summarize_profiles.py <round-tag>
reads  gpurun_out/prof_kt/**/**_kernel_stats.csv        (--kernel-trace --stats of bench.py)
       gpurun_out/prof_fetch/**/*_counter_collection.csv (--pmc FETCH_SIZE)
       gpurun_out/prof_write/**/*_counter_collection.csv (--pmc WRITE_SIZE)
       gpurun_out/prof_sq/**/*_counter_collection.csv    (--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... of tools/prof_run.py)
writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json, profiles/<tag>_sq.json and profiles/pmc_summary.json
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE is in KiB and, on gfx950, reports exactly half of
the bytes of wide coalesced reads -> bytes = 2 * 1024 * FETCH_SIZE; WRITE_SIZE (KiB) is exact.
"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
ks = glob.glob(os.path.join(root, "gpurun_out", "prof_kt", "**", "*_kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(max(ks, key=os.path.getmtime), os.path.join(out, f"{tag}_kernel_stats.csv"))
ks1 = glob.glob(os.path.join(root, "gpurun_out", "prof_kt1", "**", "*_kernel_stats.csv"), recursive=True)
if ks1:   # the same command under MSK_STREAMS=1: per-kernel durations without overlapping launches
    shutil.copy(max(ks1, key=os.path.getmtime), os.path.join(out, f"{tag}_kernel_stats_1stream.csv"))
for name in ("prof_kt.json", "prof_kt1.json"):      # the bench line printed inside each profiled run
    src = os.path.join(root, "gpurun_out", name)
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(out, f"{tag}_bench_under_{name.replace('prof_', 'rocprof_')}"))


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from build_id import build_id          # hash of the GPU library's sources (bench.py flags `profile_stale` when it differs)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("msk::", "")


def collect(sub, counter):
    tot, calls = collections.defaultdict(float), collections.defaultdict(int)
    files = glob.glob(os.path.join(root, "gpurun_out", sub, "**", "*_counter_collection.csv"), recursive=True)
    for f in ([max(files, key=os.path.getmtime)] if files else []):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            calls[k] += 1
    return tot, calls


fetch, fc = collect("prof_fetch", "FETCH_SIZE")
write, wc = collect("prof_write", "WRITE_SIZE")
summary = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    n = max(fc.get(k, 0), wc.get(k, 0), 1)
    rd = 2 * 1024 * fetch.get(k, 0.0)
    wr = 1024 * write.get(k, 0.0)
    summary[k] = {"launches": n, "fetch_size_kib": fetch.get(k, 0.0), "write_size_kib": write.get(k, 0.0),
                  "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": (rd + wr) / n,
                  "note": "read bytes = 2*1024*FETCH_SIZE (gfx950 half-count correction), write bytes = 1024*WRITE_SIZE"}
# wave-level VALU instructions per path segment (bench.py's roofline.valu): SQ_INSTS_VALU of the SQ pass over the segments
# that run traced (tools/prof_run.py prints its statistics)
import re
try:
    seg_sq = int(re.findall(r"'segments': (\d+)", open(os.path.join(root, "gpurun_out", "prof_sq.log")).read())[-1])
except Exception:
    seg_sq = 0
sq_files0 = glob.glob(os.path.join(root, "gpurun_out", "prof_sq", "**", "*_counter_collection.csv"), recursive=True)
if seg_sq and sq_files0:
    insts = collections.defaultdict(float)
    for r in csv.DictReader(open(max(sq_files0, key=os.path.getmtime))):
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            insts[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    for k, v in insts.items():
        if k in summary:
            summary[k]["valu_insts_per_segment"] = round(v / seg_sq, 3)
            summary[k]["valu_note"] = "SQ_INSTS_VALU (wave64 instructions, all launches of the SQ pass) / %d path segments of that run" % seg_sq
try:
    import subprocess
    head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    head = "?"
summary["_build_id"] = build_id()
summary["_source"] = f"profiles/{tag}_pmc.json (rocprofv3 PMC passes of tools/prof_run.py, single stream; tree at or after commit {head})"
json.dump(summary, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))

# SQ pass (one run, 8 counters): per-wave fractions of SQ_WAVE_CYCLES (all in quad-cycles, MI355X_MICROARCH.md)
sq_files = glob.glob(os.path.join(root, "gpurun_out", "prof_sq", "**", "*_counter_collection.csv"), recursive=True)
sq2_files = glob.glob(os.path.join(root, "gpurun_out", "prof_sq2", "**", "*_counter_collection.csv"), recursive=True)
if sq_files:
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in [max(sq_files, key=os.path.getmtime)] + ([max(sq2_files, key=os.path.getmtime)] if sq2_files else []):
        for r in csv.DictReader(open(f)):
            tot[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    sq = {}
    for k, v in sorted(tot.items()):
        if not k.startswith("k_"):
            continue
        wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        sq[k] = {"SQ_WAVE_CYCLES": v.get("SQ_WAVE_CYCLES", 0.0)}
        sq[k].update({c + "/WAVE_CYCLES": round(x / wc, 4) for c, x in v.items() if c != "SQ_WAVE_CYCLES"})
    json.dump(sq, open(os.path.join(out, f"{tag}_sq.json"), "w"), indent=1)
    print(json.dumps(sq, indent=1))
