"""Exec-mask traffic of the leap kernel's Newton loop by source line: s_*_saveexec / writes of exec / s_cbranch per line of the hot copy (tools/diag/isa_hot_loop.py's
attribution).  Short-circuit `||` / `&&` and small `if` bodies that the compiler did not if-convert show up here.  usage: python tools/diag/isa_exec_by_line.py build/isa/v5.s [lean|hand]"""
import collections, re, sys
asm = sys.argv[1]; which = sys.argv[2] if len(sys.argv) > 2 else "lean"
src = open("judo_amd/csrc/jh_engine_v5.hip").read().split("\n")
pat = "done = solve_step(std::integral_constant<int, NSLOT>{}, std::false_type{})" if which == "lean" else "if (!done) solve_step(std::integral_constant<int, NSLOT>{}, std::integral_constant<bool, SELF>{})"
l_solve = next(i + 1 for i, l in enumerate(src) if pat in l)
l_loop = next(i + 1 for i, l in enumerate(src) if l.strip().startswith("else newton_loop(std::false_type{});")) if which == "lean" else next(i + 1 for i, l in enumerate(src) if "NS == NSLOT" in l and "newton_loop(std::false_type{})" in l)
c_loop = src[l_loop - 1].rindex("newton_loop(std::false_type{})") + 1
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and "k_leap_v5ILb0ELi4ELb1" in l)
end = next(i for i, l in enumerate(lines) if i > start and l.startswith("_ZN") and "k_leap_v5" in l)
hot = False; cur = None
ex = collections.Counter(); br = collections.Counter(); salu = collections.Counter(); tot = collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)\s.*?; (\S+?):(\d+):\d+(.*)", l)
    if m:
        hot = (f":{l_loop}:{c_loop} " in l or f":{l_loop}:{c_loop}]" in l or f"hip:{l_loop}:{c_loop}" in l) and f"hip:{l_solve}:" in l
        f = m.group(2).split("/")[-1]
        chain = re.findall(r"(\w+\.h(?:ip)?):(\d+):", l)
        cur = (f, int(m.group(3)), tuple(int(b) for a, b in chain if a == "jh_engine_v5.hip"))
        continue
    if not hot: continue
    t = l.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"): continue
    op = t.split(" ")[0]
    key = (cur[0], cur[1])
    tot[key] += 1
    if "saveexec" in op or re.search(r"\bexec\b", t.split(" ", 1)[1] if " " in t else "") and op.startswith("s_") and not op.startswith("s_cbranch"): ex[key] += 1
    if op.startswith("s_cbranch") or op == "s_branch": br[key] += 1
    if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")): salu[key] += 1
print(f"{which} copy: {sum(tot.values())} instructions, {sum(salu.values())} SALU of which {sum(ex.values())} touch exec, {sum(br.values())} branches")
def text(f, n):
    try: return (open("judo_amd/csrc/" + f).read().split("\n")[n - 1]).strip()[:120]
    except Exception: return ""
for key, n in sorted(salu.items(), key=lambda kv: -kv[1])[:40]:
    print(f"  {key[0]}:{key[1]:5d}  salu {n:3d} exec {ex[key]:3d} branches {br[key]:2d} of {tot[key]:4d}  | {text(*key)}")
