"""Exec-mask regions (s_*_saveexec) of the Spot tree kernel k_tree_v4<true> by the line of the kernel body they were inlined into (the outermost jh_engine_v4.hip line of the
.loc inlined-at chain) and by the innermost source line.  usage: python tools/diag/isa_exec_v4.py [build/isa/v4.s]  (listing: hipcc ... -gline-tables-only -S, see tools/diag/isa_v5.sh)"""
import collections, re, sys
asm = sys.argv[1] if len(sys.argv) > 1 else "build/isa/v4.s"
src = open("judo_amd/csrc/jh_engine_v4.hip").read().split("\n")
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and "k_tree_v4ILb1" in l)
end = next((i for i, l in enumerate(lines) if i > start and l.startswith("_ZN") and "k_tree_v4" in l), len(lines))
cur = (0, (0, ""))
tot, ex, exin = collections.Counter(), collections.Counter(), collections.Counter()
n = nex = 0
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)\s.*?; (\S+?):(\d+):\d+(.*)", l)
    if m:
        f = m.group(2).split("/")[-1]
        mm = re.findall(r"jh_engine_v4\.hip:(\d+):", l)
        outer = int(mm[-1]) if mm else (int(m.group(3)) if f == "jh_engine_v4.hip" else 0)
        cur = (outer, (int(m.group(3)), f))
        continue
    t = l.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    op = t.split(" ")[0]; n += 1; tot[cur[0]] += 1
    if "saveexec" in op:
        nex += 1; ex[cur[0]] += 1; exin[cur[1]] += 1
print(f"k_tree_v4<true>: {n} instructions, {nex} s_*_saveexec")
print("by the kernel-body line they were inlined into:")
for ln, c in ex.most_common(40):
    print(f"  {c:3d} of {tot[ln]:5d}  :{ln} | {src[ln - 1].strip()[:150] if 0 < ln <= len(src) else ''}")
print("by innermost source line:")
for (ln, f), c in exin.most_common(25):
    print(f"  {c:3d}  {f}:{ln}")
