"""Per-source-line instruction and LDS-wait counts of ONE inlined copy of the leap kernel's Newton loop (the common one: two slots, no dense code), from an assembly
listing with line tables (hipcc ... -gline-tables-only -S).  The copy is recognised by the call-site columns in the `.loc` inlined-at chains.
usage: python tools/diag/isa_hot_loop.py build/isa/v5.s [hand|lean]"""
import collections, re, sys
asm = sys.argv[1]
src = open("judo_amd/csrc/jh_engine_v5.hip").read().split("\n")
# call sites: newton_loop(std::false_type{}) inside `else if constexpr (SELF && NS == NSLOT)` and solve_step(NSLOT)
# which copy of the solver: "hand" (default) = the hand-capable copy, "lean" = the copy without the hand-contact code (JH_V5_HCSPLIT dispatch)
which = sys.argv[2] if len(sys.argv) > 2 else "hand"
pat = "done = solve_step(std::integral_constant<int, NSLOT>{}, std::false_type{})" if which == "lean" else "if (!done) solve_step(std::integral_constant<int, NSLOT>{}, std::integral_constant<bool, SELF>{})"
l_solve = next(i + 1 for i, l in enumerate(src) if pat in l)
if which == "lean":  # HC = false: the loop is instantiated by the last branch, `else newton_loop(std::false_type{});`
    l_loop = next(i + 1 for i, l in enumerate(src) if l.strip().startswith("else newton_loop(std::false_type{});"))
else:
    l_loop = next(i + 1 for i, l in enumerate(src) if "NS == NSLOT" in l and "newton_loop(std::false_type{})" in l)
c_loop = src[l_loop - 1].rindex("newton_loop(std::false_type{})") + 1
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and "k_leap_v5ILb0ELi4ELb1" in l)
end = next(i for i, l in enumerate(lines) if i > start and l.startswith("_ZN") and "k_leap_v5" in l)
hot = False; cur = None
ins = collections.Counter(); waits = collections.Counter(); cats = collections.defaultdict(collections.Counter)
def cat(op, rest):
    if op.startswith("scratch"): return "scratch"
    if "dpp" in rest or "quad_perm" in rest or "row_" in rest: return "dpp"
    if op.startswith("v_mov") or op.startswith("v_pk_mov") or op.startswith("v_cndmask"): return "mov/sel"
    if op.startswith("v_readlane") or op.startswith("v_writelane"): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    return "mem"
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)\s.*?; (\S+?):(\d+):\d+(.*)", l)
    if m:
        chain = m.group(4)
        hot = (f":{l_loop}:{c_loop} " in l or f":{l_loop}:{c_loop}]" in l or f"hip:{l_loop}:{c_loop}" in l) and f"hip:{l_solve}:" in l
        f = m.group(2).split("/")[-1]
        mm = re.findall(r"jh_engine_v5\.hip:(\d+):", l)
        main = int(m.group(3)) if f == "jh_engine_v5.hip" else (int(mm[0]) if mm else 0)
        cur = (main, f if f != "jh_engine_v5.hip" else "")
        continue
    if not hot: continue
    t = l.strip()
    if not t or t.startswith(".") or t.startswith(";") or t.endswith(":"): continue
    op, _, rest = t.partition(" ")
    ins[cur[0]] += 1; cats[cur[0]][cat(op, rest)] += 1
    if op.startswith("s_waitcnt") and "lgkmcnt(0)" in rest: waits[cur[0]] += 1
tot = sum(ins.values())
print(f"hot copy (newton_loop at line {l_loop}:{c_loop} via solve_step at {l_solve}): {tot} instructions, {sum(waits.values())} s_waitcnt lgkmcnt(0)")
allc = collections.Counter()
for c in cats.values(): allc += c
print("mix:", dict(allc))
print("top source lines:")
for ln, n in ins.most_common(45):
    print(f"  {ln:5d} {n:5d} ({100 * n / tot:4.1f} %) waits {waits[ln]:3d} {dict(cats[ln])}  | {src[ln - 1].strip()[:110] if 0 < ln <= len(src) else ''}")
