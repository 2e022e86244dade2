#!/usr/bin/env python3
"""tools/cpu_contention.py — the bench step under a CPU budget, for every host-side wait strategy.

Eight ranks of one node share its CPU quota (the bench boxes: 16 CPUs), so a rank has about two.  This runs bench.py's N = 1
step in fresh child processes restricted (os.sched_setaffinity in the child, before anything touches a GPU: bench.py --cpus) to
16 / 4 / 2 / 1 CPUs for each MSK_WAIT mode and for 4 / 1 loop threads, and prints ms per step and the CPU seconds the process
burnt per step; then the two-rank rehearsal (both ranks on cuda:0, gloo) on 2 CPUs in total.  One child at a time; nothing is
restarted or replaced.  Output: a table on stdout + gpurun_out/cpu_contention.json.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cpus, wait, threads, extra=(), steps=10, warmup=2, more_env=None):
    env = dict(os.environ, MSK_WAIT=wait, MSK_HOST_THREADS=str(threads), **(more_env or {}))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline",
           "--no-other-configs", "--cpus", str(cpus)] + list(extra)
    t0 = time.time()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    line = next((l for l in reversed(p.stdout.splitlines()) if l.startswith("{")), None)
    if p.returncode != 0 or line is None:
        return {"error": (p.stderr or p.stdout)[-300:], "rc": p.returncode}
    j = json.loads(line)
    hs = j.get("host_side", {})
    return {"ms_per_step": j["ms_per_step"], "value": j["value"], "cpu_s_per_step": hs.get("cpu_s_per_step"), "cpus_busy": hs.get("cpus_busy"),
            "cpus_allowed": hs.get("cpus_allowed"), "busiest_threads": hs.get("busiest_threads"), "child_wall_s": round(time.time() - t0, 1)}


def experiments():
    """Who is the thread of the HIP / ROCr runtime that stays busy through the render whatever the library's wait does?
    The default (callback, one loop thread) under a few runtime settings, 16 CPUs."""
    out = []
    for env in ({}, {"MSK_TIMING_EVERY": "1000000"}, {"HSA_ENABLE_INTERRUPT": "0"}, {"ROC_ACTIVE_WAIT_TIMEOUT": "0"}, {"ROC_CPU_WAIT_FOR_SIGNAL": "0"},
                {"HSA_ENABLE_MWAITX": "1"}, {"AMD_DIRECT_DISPATCH": "0"}, {"MSK_SYNC_GROUP": "16"}, {"MSK_SYNC_GROUP": "32"},
                {"MSK_TIMING_EVERY": "1000000", "MSK_SYNC_GROUP": "32"}):
        for wait in ("callback", "sleep"):
            r = run(16, wait, 1, more_env=env)
            r.update(env=env, wait=wait)
            out.append(r)
            print(json.dumps(r), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cpu_contention_experiments.json"), "w"), indent=1)


def main():
    if "--experiments" in sys.argv:
        return experiments()
    quick = "--quick" in sys.argv
    cpu_counts = [16, 2] if quick else [16, 4, 2, 1]
    waits = ["poll", "callback"] if quick else ["poll", "sleep", "callback", "event", "sync"]
    out = {"rows": [], "rehearsal": []}
    print(f"{'cpus':>4} {'wait':>9} {'threads':>7} {'ms/step':>9} {'cpu s/step':>11} {'cpus busy':>9}", flush=True)
    for cpus in cpu_counts:
        for wait in waits:
            for threads in (4, 1):
                r = run(cpus, wait, threads)
                r.update(cpus=cpus, wait=wait, threads=threads)
                out["rows"].append(r)
                if "error" in r:
                    print(f"{cpus:>4} {wait:>9} {threads:>7} ERROR {r['error'][-120:]!r}", flush=True)
                else:
                    print(f"{cpus:>4} {wait:>9} {threads:>7} {r['ms_per_step']:>9.2f} {r['cpu_s_per_step']:>11.4f} {r['cpus_busy']:>9.2f}", flush=True)
                json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cpu_contention.json"), "w"), indent=1)
    # N ranks on one GPU sharing C CPUs: rank 0's line; N times the samples per step.  6 ranks (the most one card takes) of the
    # old default want 6 x 4.5 CPUs of a 16-CPU quota: the throttling case, with the GPU time-sliced six ways under it
    for ranks, cpus in ([(2, 2)] if quick else [(2, 16), (2, 2), (6, 16), (6, 6)]):
        for wait, threads in (("poll", 4), ("poll", 1), ("callback", 1), ("sleep", 1)):
            r = run(cpus, wait, threads, extra=("--gpus", str(ranks), "--rehearse-on-one-gpu"), steps=4 if ranks > 2 else 6)
            r.update(cpus=cpus, wait=wait, threads=threads, ranks=ranks)
            out["rehearsal"].append(r)
            print(f"{ranks} ranks on cuda:0, {cpus} CPUs in all, {wait}, {threads} thread(s): " + json.dumps(r), flush=True)
            json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cpu_contention.json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
