#!/usr/bin/env python3
"""Static per-source-line instruction profile of one kernel.
Build with line tables first:
  hipcc <flags of __graft_entry__.HIPCC_FLAGS> -gline-tables-only --save-temps -o /tmp/isa/lib.so misaki-render_amd/csrc/msk_gpu.hip
usage: isa_lines.py file.s kernel_substring [top_n] [--inlined]
Every instruction is charged to the source line of its most recent `.loc` (the innermost inlined frame), weighted by a
rough issue cost in quad-cycles: fp32 VALU 1, fp64 VALU 2, transcendental 4 (fp64 transcendental 8), LDS / VMEM / SALU 1.
Static counts: a loop body counts once, both sides of a branch count."""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
top_n = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
def cost(op):
    if op.startswith("v_"):
        trans = op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos"))
        if "f64" in op: return ("f64", 8 if trans else 2)
        if trans: return ("trans", 4)
        if op.startswith("v_div_"): return ("div", 1)
        return ("valu", 1)
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return ("wait", 0)
    if op.startswith("s_"): return ("salu", 1)
    if op.startswith("ds_"): return ("lds", 1)
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return ("vmem", 1)
    return ("other", 1)
per = collections.defaultdict(collections.Counter)
cur = ("?", 0)
tot = collections.Counter()
for i in range(start + 1, end):
    l = lines[i].strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    if not l or l.startswith((";", ".")) or l.endswith(":"): continue
    c, w = cost(l.split()[0])
    per[cur][c] += 1; per[cur]["cycles"] += w
    tot[c] += 1; tot["cycles"] += w
print(key, dict(tot))
src_cache = {}
def src(f, n):
    import os
    for d in ("misaki-render_amd/csrc", "include"):
        p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), d, f)
        if os.path.exists(p):
            if p not in src_cache: src_cache[p] = open(p).read().split("\n")
            return src_cache[p][n - 1].strip()[:90] if 0 < n <= len(src_cache[p]) else ""
    return ""
for (f, n), c in sorted(per.items(), key=lambda kv: -kv[1]["cycles"])[:top_n]:
    d = {k: v for k, v in c.items() if k != "cycles"}
    print(f"{c['cycles']:6d} {100.0 * c['cycles'] / tot['cycles']:5.1f}%  {f}:{n:<5d} {d}  | {src(f, n)}")
