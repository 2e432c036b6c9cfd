"""Exec-mask traffic of the WHOLE leap kernel by innermost source line (outside the Newton loop too): SALU / exec writes / branches per line.  usage: isa_exec_whole.py build/isa/v5.s [lo hi]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 10 ** 9)
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and "k_leap_v5ILb0ELi4ELb1" in l)
end = next(i for i, l in enumerate(lines) if i > start and l.startswith("_ZN") and "k_leap_v5" in l)
cur = None; salu = collections.Counter(); tot = collections.Counter(); br = collections.Counter(); ex = collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)\s.*?; (\S+?):(\d+):\d+(.*)", l)
    if m:
        f = m.group(2).split("/")[-1]; mm = [int(x) for x in re.findall(r"jh_engine_v5\.hip:(\d+):", l)]
        outer = mm[-1] if mm else (int(m.group(3)) if f == "jh_engine_v5.hip" else 0)  # outermost v5 line of the inlining chain
        inner = int(m.group(3)) if f == "jh_engine_v5.hip" else (mm[0] if mm else 0)
        cur = (f, int(m.group(3)), inner, outer); continue
    t = l.strip()
    if cur is None or not t or t.startswith((".", ";")) or t.endswith(":"): continue
    if not (lo <= cur[2] < hi): continue
    op = t.split(" ")[0]; key = (cur[0], cur[1])
    tot[key] += 1
    if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")): salu[key] += 1
    if "saveexec" in op or (op.startswith("s_") and re.search(r"\bexec\b", t)): ex[key] += 1
    if op.startswith("s_cbranch") or op == "s_branch": br[key] += 1
print(f"lines [{lo},{hi}): {sum(tot.values())} instructions, {sum(salu.values())} SALU, {sum(ex.values())} exec, {sum(br.values())} branches")
def text(f, n):
    try: return open("judo_amd/csrc/" + f).read().split("\n")[n - 1].strip()[:115]
    except Exception: return ""
for key, n in sorted(salu.items(), key=lambda kv: -kv[1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    print(f"  {key[0]}:{key[1]:5d} salu {n:3d} exec {ex[key]:3d} br {br[key]:2d} of {tot[key]:4d} | {text(*key)}")
