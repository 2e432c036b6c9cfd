"""SALU / exec-mask / branch counts by source line for one kernel of an assembly listing with line tables.  usage: isa_exec_any.py <file.s> <kernel symbol substring> <source file name> [top]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n"); sym = sys.argv[2]; srcname = sys.argv[3]; top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
islabel = lambda l: l.startswith("_Z") and re.match(r"^\S+:(\s|$)", l) is not None
start = next(i for i, l in enumerate(lines) if islabel(l) and sym in l)
end = next((i for i, l in enumerate(lines) if i > start and islabel(l)), len(lines))
cur = None; salu = collections.Counter(); tot = collections.Counter(); br = collections.Counter(); ex = collections.Counter(); cls = collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)\s.*?; (\S+?):(\d+):\d+(.*)", l)
    if m: cur = (m.group(2).split("/")[-1], int(m.group(3))); continue
    t = l.strip()
    if cur is None or not t or t.startswith((".", ";")) or t.endswith(":"): continue
    op = t.split(" ")[0]; tot[cur] += 1
    cls["valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")) else "lds" if op.startswith("ds_") else "wait" if op.startswith("s_") else "mem"] += 1
    if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")): salu[cur] += 1
    if "saveexec" in op or (op.startswith("s_") and re.search(r"\bexec\b", t)): ex[cur] += 1
    if op.startswith("s_cbranch") or op == "s_branch": br[cur] += 1
print(f"{sym}: {sum(tot.values())} instructions {dict(cls)}; exec writes {sum(ex.values())}, branches {sum(br.values())}")
def text(f, n):
    try: return open("judo_amd/csrc/" + f).read().split("\n")[n - 1].strip()[:118]
    except Exception: return ""
for key, n in sorted(salu.items(), key=lambda kv: -kv[1])[:top]:
    print(f"  {key[0]}:{key[1]:5d} salu {n:3d} exec {ex[key]:3d} br {br[key]:2d} of {tot[key]:4d} | {text(*key)}")
