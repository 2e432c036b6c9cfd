"""Summarise a rocprofv3 PC-sampling CSV: samples per source line (Instruction_Comment) and per instruction class of the leap kernel.  usage: pcs_summary.py <pc_sampling csv>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "samples; columns:", list(rows[0].keys()) if rows else None)
ic = next((k for k in rows[0] if k.lower().startswith("instruction") and "comment" not in k.lower()), None); cc = next((k for k in rows[0] if "comment" in k.lower()), None)
byline = collections.Counter(); byop = collections.Counter(); byfile = collections.Counter()
for r in rows:
    ins = (r.get(ic) or "").strip(); com = (r.get(cc) or "").strip()
    byop[ins.split(" ")[0] if ins else "?"] += 1
    m = re.search(r"([A-Za-z0-9_./-]+):(\d+)", com)
    if m: byline[(m.group(1).split("/")[-1], int(m.group(2)))] += 1; byfile[m.group(1).split("/")[-1]] += 1
    else: byline[("?", 0)] += 1
n = len(rows)
print("by file:", byfile.most_common(8))
print("top opcodes:", [(k, round(100 * v / n, 2)) for k, v in byop.most_common(40)])
print("top source lines (% of samples):")
for (f, l), v in byline.most_common(150): print(f"  {f}:{l}  {100 * v / n:.2f}")
# per 25-line bucket of the main kernel file
b = collections.Counter()
for (f, l), v in byline.items():
    if f.startswith("jh_engine_v5"): b[l // 25 * 25] += v
print("jh_engine_v5.hip per 25-line bucket:")
for k in sorted(b): print(f"  {k:5d}  {100 * b[k] / n:.2f}")
