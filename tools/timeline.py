#!/usr/bin/env python3
"""tools/timeline.py <kernel_trace.csv> [step] : the shape of one bench step in a rocprofv3 kernel trace (tools/profile_all.sh's
`kt` pass).  A step = the dispatches between two k_film_put.  Prints, per millisecond of the step, how many k_shade_gen / k_trace_q
/ k_wavefront dispatches were RUNNING on average (sum of their overlap with the bucket / bucket length), and per queue the
durations of the shading launches in order — where the step's time goes once the pool starts to drain."""
import csv, sys, collections
path = sys.argv[1]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        k = "shade" if "k_shade_gen" in n else "trace" if "k_trace" in n else "wavefront" if "k_wavefront" in n else "resolve" if "k_resolve" in n else "film_put" if "k_film_put" in n else "reduce" if "k_reduce_ctl" in n else None
        if k: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, int(r["Queue_Id"]), int(r["Grid_Size_X"])))
rows.sort()
puts = [i for i, r in enumerate(rows) if r[2] == "film_put"]
lo = puts[want - 1] + 1 if want > 0 else 0
hi = puts[want]
step = rows[lo:hi + 1]
t0 = step[0][0]; t1 = step[-1][1]
print("step %d: %.3f ms, %d dispatches" % (want, (t1 - t0) / 1e6, len(step)))
nb = int((t1 - t0) / 1e6) + 1
occ = collections.defaultdict(lambda: [0.0] * nb)
for s, e, k, q, g in step:
    b0 = int((s - t0) / 1e6); b1 = int((e - t0) / 1e6)
    for b in range(b0, min(b1, nb - 1) + 1):
        a = max(s, t0 + b * 1000000); z = min(e, t0 + (b + 1) * 1000000)
        if z > a: occ[k][b] += (z - a) / 1e6
kinds = ["shade", "trace", "wavefront", "resolve", "reduce"]
print("ms   " + " ".join("%9s" % k for k in kinds) + "   (dispatches running, averaged over the millisecond)")
for b in range(nb):
    print("%3d  " % b + " ".join("%9.2f" % occ[k][b] for k in kinds))
qs = sorted({r[3] for r in step if r[2] == "shade"})
for q in qs:
    d = [(r[0], r[1], r[4]) for r in step if r[2] == "shade" and r[3] == q]
    print("queue %d: %d shade launches, grid %d; durations us: %s" % (q, len(d), d[0][2], " ".join("%d" % ((e - s) / 1000) for s, e, g in d)))
    gaps = [(d[i + 1][0] - d[i][0]) / 1000 for i in range(len(d) - 1)]
    print("   start-to-start us: " + " ".join("%d" % g for g in gaps))
