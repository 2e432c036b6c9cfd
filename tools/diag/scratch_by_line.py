"""Static attribution of register-spill traffic (scratch_load / scratch_store, v_accvgpr moves, v_readlane / v_writelane) to source lines for one kernel in an
assembly listing built with -gline-tables-only.
usage: python tools/diag/scratch_by_line.py build/isa/v5.s <kernel-substring> [source.hip]"""
import collections, re, sys
asm, pat = sys.argv[1], sys.argv[2]
src = open(sys.argv[3]).read().split("\n") if len(sys.argv) > 3 else None
lines = open(asm).read().split("\n")
heads = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
start = next(i for i in heads if pat in lines[i])
end = next((i for i in heads if i > start), len(lines))
files = {}
cur = None
cnt = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r"\s+(scratch_load|scratch_store|v_accvgpr_read|v_accvgpr_write|v_readlane|v_writelane)", l)
    if m: cnt[cur][m.group(1)] += 1
tot = collections.Counter()
for c in cnt.values(): tot.update(c)
print("totals:", dict(tot))
byfile = collections.Counter()
for (f, ln), c in cnt.items(): byfile[f] += c["scratch_load"] + c["scratch_store"]
main = max(byfile, key=byfile.get) if byfile else None
rows = sorted(((k, c) for k, c in cnt.items() if c["scratch_load"] + c["scratch_store"] > 0), key=lambda kc: (kc[0][0] != main, kc[0][1]))
for (f, ln), c in rows:
    text = src[ln - 1].strip()[:110] if (src and f == main and 0 < ln <= len(src)) else ""
    print(f"file {f} line {ln:5d}: load {c['scratch_load']:4d} store {c['scratch_store']:4d}   {text}")
