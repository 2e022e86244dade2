#!/usr/bin/env python3
"""tools/tail_timeline.py <kernel_trace.csv> [render] — the shape of one render of a mesh config in a rocprofv3 kernel trace
(rocprofv3 --kernel-trace -- python3 tools/prof_mesh.py c5 128 2): per stream the traversal launches in order, where their
durations fall below a fifth of the longest (the thin end of the pass: every sample started, Russian roulette's tail draining),
and how much of the render's wall time that window is."""
import collections, csv, sys
path = sys.argv[1]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    k = "shade" if "k_shade_gen" in n else "trace" if "k_trace" in n else "film" if "k_film_put" in n else "resolve" if "k_resolve" in n else \
        "wavefront" if "k_wavefront" in n else None
    if k:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, int(r["Queue_Id"])))
rows.sort()
puts = [i for i, r in enumerate(rows) if r[2] == "film"]
step = rows[(puts[want - 1] + 1) if want else 0:puts[want] + 1]
t0, t1 = step[0][0], step[-1][1]
res = [s for s, e, k, q in step if k == "resolve"]
byq = collections.defaultdict(lambda: collections.defaultdict(list))
for s, e, k, q in step:
    byq[q][k].append((s, e))
print("render %d: %.1f ms; the film replay starts at %.1f ms" % (want, (t1 - t0) / 1e6, (res[0] - t0) / 1e6))
tails = []
for q, kinds in sorted(byq.items()):
    tr = kinds.get("trace", [])
    if not tr:
        continue
    d = [(e - s) / 1e3 for s, e in tr]
    first = next((i for i in range(len(d)) if all(x < 0.2 * max(d) for x in d[i:])), len(d) - 1)
    tails.append(tr[first][0])
    sh = [(e - s) / 1e3 for s, e in kinds.get("shade", [])]
    print("  stream %d: %d traversal launches (%.1f ms in all, shading %.1f ms); thin from launch %d at %.1f ms, %.2f ms of traversal + %.2f ms of shading after that" %
          (q, len(d), sum(d) / 1e3, sum(sh) / 1e3, first, (tr[first][0] - t0) / 1e6, sum(d[first:]) / 1e3, sum(sh[first:]) / 1e3))
    if len(byq) == 1:
        print("    traversal us per iteration:", " ".join("%d" % x for x in d))
        print("    shading   us per iteration:", " ".join("%d" % x for x in sh))
if tails:
    w0 = min(tails)
    print("  thin end: %.1f ms .. %.1f ms = %.1f ms of the render's %.1f (%.0f %%)" % ((w0 - t0) / 1e6, (res[0] - t0) / 1e6, (res[0] - w0) / 1e6, (t1 - t0) / 1e6,
                                                                                  100.0 * (res[0] - w0) / (t1 - t0)))
