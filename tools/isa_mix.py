#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc --save-temps .s file, per basic block.
usage: isa_mix.py file.s kernel_substring [--blocks]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
show_blocks = "--blocks" in sys.argv
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
def cls(op):
    if op.startswith("v_") and "f64" in op: return "valu_f64"
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")): return "valu_trans"
    if op.startswith(("v_div_", )): return "valu_div"
    if op.startswith("v_mul_lo") or op.startswith("v_mul_hi") or op.startswith("v_mad_u64") or op.startswith("v_mad_i64"): return "valu_imul"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    return "other"
blocks = []; cur = ["entry", collections.Counter(), start]
for i in range(start + 1, end + 1):
    l = lines[i].strip()
    if not l or l.startswith(";") or l.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur); cur = [l.split(":")[0], collections.Counter(), i]
        continue
    op = l.split()[0]
    cur[1][cls(op)] += 1
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        cur[1]["->" + l.split()[-1]] += 0
blocks.append(cur)
tot = collections.Counter()
for b in blocks: tot.update({k: v for k, v in b[1].items() if not k.startswith("->")})
print(key, "blocks", len(blocks), dict(tot))
if show_blocks:
    for b in blocks:
        c = {k: v for k, v in b[1].items() if not k.startswith("->")}
        tgt = [k for k in b[1] if k.startswith("->")]
        n = sum(c.values())
        if n >= 8: print(f"{b[0]:12s} line {b[2]:6d} n={n:5d} {c} {' '.join(tgt)}")
