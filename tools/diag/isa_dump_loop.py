"""Listing-order dump of ONE inlined copy of the leap kernel's Newton loop (see isa_hot_loop.py for how the copy is recognised), each instruction tagged with its source line.
usage: python tools/diag/isa_dump_loop.py build/isa/v5.s [hand|lean] > loop.txt"""
import re, sys
asm = sys.argv[1]; which = sys.argv[2] if len(sys.argv) > 2 else "hand"
src = open("judo_amd/csrc/jh_engine_v5.hip").read().split("\n")
pat = "done = solve_step(std::integral_constant<int, NSLOT>{}, std::false_type{})" if which == "lean" else "if (!done) solve_step(std::integral_constant<int, NSLOT>{}, std::integral_constant<bool, SELF>{})"
l_solve = next(i + 1 for i, l in enumerate(src) if pat in l)
if which == "lean": l_loop = next(i + 1 for i, l in enumerate(src) if l.strip().startswith("else newton_loop(std::false_type{});"))
else: l_loop = next(i + 1 for i, l in enumerate(src) if "NS == NSLOT" in l and "newton_loop(std::false_type{})" in l)
c_loop = src[l_loop - 1].rindex("newton_loop(std::false_type{})") + 1
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and "k_leap_v5ILb0ELi4ELb1" in l)
end = next(i for i, l in enumerate(lines) if i > start and l.startswith("_ZN") and "k_leap_v5" in l)
hot = False; tag = ""
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)\s.*?; (\S+?):(\d+):\d+(.*)", l)
    if m:
        hot = (f":{l_loop}:{c_loop} " in l or f":{l_loop}:{c_loop}]" in l or f"hip:{l_loop}:{c_loop}" in l) and f"hip:{l_solve}:" in l
        f = m.group(2).split("/")[-1]; mm = re.findall(r"jh_engine_v5\.hip:(\d+):", l)
        tag = f"{f.replace('jh_engine_v5.hip','')}:{m.group(3)}" + (f"<{mm[0]}" if f != "jh_engine_v5.hip" and mm else "")
        continue
    t = l.strip()
    if not t or t.startswith((".", ";")): continue
    if t.endswith(":") and hot: print(t); continue
    if hot: print(f"  {t.split(';')[0].rstrip():<70s} {tag}")
