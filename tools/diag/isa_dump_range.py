"""Listing-order dump of the instructions of one kernel whose OUTERMOST jh_engine_v5.hip (or given file) source line lies in [a, b] (assembly with line tables).
usage: python tools/diag/isa_dump_range.py build/isa/v5.s k_leap_v5ILb0ELi4ELb1 jh_engine_v5.hip 780 960 > out.txt"""
import re, sys
asm, key, fname, a, b = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
lines = open(asm).read().split("\n")
islabel = lambda l: l.startswith("_Z") and re.match(r"^\S+:(\s|$)", l) is not None
start = next(i for i, l in enumerate(lines) if islabel(l) and key in l)
end = next((i for i, l in enumerate(lines) if i > start and islabel(l)), len(lines))
hot = False; tag = ""
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)\s.*?; (\S+?):(\d+):\d+(.*)", l)
    if m:
        f = m.group(2).split("/")[-1]; mm = re.findall(re.escape(fname) + r":(\d+):", l)
        outer = int(mm[-1]) if mm else (int(m.group(3)) if f == fname else 0)
        hot = a <= outer <= b
        tag = (f"{f}:" if f != fname else ":") + m.group(3) + (f"<{outer}" if (f != fname or int(m.group(3)) != outer) else "")
        continue
    t = l.strip()
    if not t or t.startswith((".", ";")): continue
    if t.endswith(":") and hot: print(t); continue
    if hot: print(f"  {t.split(';')[0].rstrip():<78s} {tag}")
