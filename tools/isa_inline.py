#!/usr/bin/env python3
"""Static instruction profile of one kernel by SECTION of a chosen outer function, using the DWARF inline chains.
Build with line tables first (see isa_lines.py), then e.g.

  isa_inline.py /tmp/isa/msk_gpu-hip-amdgcn-amd-amdhsa-gfx950.out k_shade_genILb1ELb1 shade_region \
        958-1153:main 1154-1179:emit 1180-1195:compact 1196-1202:emit_tail 1203-1243:regen

Every instruction is charged to the line of `outer function` found in its inline chain (llvm-symbolizer --inlining) and
to the section that line falls in; within a section the innermost frames are listed by cost.  Cost model (quad-cycles
of issue): VALU 1 (fp64 too: full rate on gfx950), transcendental / 32-bit integer multiply 4, fp64 transcendental 8,
SALU / LDS / VMEM 1, waits 0."""
import collections, re, subprocess, sys
LLVM = "/opt/rocm/lib/llvm/bin/"
elf, key, outer = sys.argv[1], sys.argv[2], sys.argv[3]
sections = []
for a in sys.argv[4:]:
    rng, name = a.split(":")
    lo, hi = rng.split("-")
    sections.append((int(lo), int(hi), name))
syms = subprocess.run([LLVM + "llvm-readelf", "-s", "-W", elf], capture_output=True, text=True).stdout.split("\n")
sym = next(l.split()[-1] for l in syms if key in l and " FUNC " in l)
dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + sym, elf],
                     capture_output=True, text=True).stdout.split("\n")
ins = []
for l in dis:
    m = re.match(r"\s+(\S+)\s.*//\s*([0-9A-F]+):", l)
    if m: ins.append((int(m.group(2), 16), m.group(1)))
addr_in = "\n".join(hex(a) for a, _ in ins) + "\n"
symz = subprocess.run([LLVM + "llvm-symbolizer", "-e", elf, "--inlining", "-f", "-s"], input=addr_in, capture_output=True,
                      text=True).stdout.split("\n\n")
def cost(op):
    if op.startswith("v_"):
        trans = op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos"))
        if "f64" in op: return 8 if trans else 1
        if trans: return 4
        if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_lo_i32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32")): return 4
        return 1
    if op.startswith(("s_waitcnt", "s_nop")): return 0
    return 1
sec_tot = collections.Counter(); sec_inner = collections.defaultdict(collections.Counter)
sec_cls = collections.defaultdict(collections.Counter)
total = 0
for (addr, op), blk in zip(ins, symz):
    ls = [x for x in blk.split("\n") if x.strip()]
    fr = [(ls[k], ls[k + 1]) for k in range(0, len(ls) - 1, 2)]         # (function, file:line:col), innermost first
    sec = "other"
    for fn, loc in fr:
        if outer in fn:
            try: ln = int(loc.split(":")[1])
            except Exception: ln = 0
            for lo, hi, name in sections:
                if lo <= ln <= hi: sec = name
            break
    c = cost(op)
    total += c
    sec_tot[sec] += c
    inner_fn, inner_loc = fr[0] if fr else ("?", "?")
    short = re.sub(r"\(.*", "", inner_fn).replace("msk::", "")
    sec_inner[sec][short + " " + ":".join(inner_loc.split(":")[:2])] += c
    k = "f64" if "f64" in op else "trans" if cost(op) == 4 and op.startswith(("v_rcp", "v_sqrt", "v_rsq")) else "imul" if cost(op) == 4 else \
        "div" if op.startswith("v_div_") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else \
        "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else "salu"
    sec_cls[sec][k] += 1
print(f"{sym}: {len(ins)} instructions, {total} quad-cycles (static)")
for sec, c in sec_tot.most_common():
    print(f"\n== {sec}: {c} quad-cycles ({100.0 * c / total:.1f} %)  {dict(sec_cls[sec])}")
    for k, v in sec_inner[sec].most_common(int(__import__("os").environ.get("TOP", "14"))):
        print(f"   {v:5d}  {k}")
