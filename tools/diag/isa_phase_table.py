"""Static size of one Newton iteration of the leap kernel by PHASE (source-line ranges of jh_engine_v5.hip), from tools/diag/isa_hot_loop.py's per-line attribution.
usage: python tools/diag/isa_phase_table.py build/isa/v5.s [lean|hand]"""
import collections, re, subprocess, sys
which = sys.argv[2] if len(sys.argv) > 2 else "lean"
src = open("judo_amd/csrc/jh_engine_v5.hip").read().split("\n")
def line_of(pat, after=0): return next(i + 1 for i, l in enumerate(src) if i + 1 > after and pat in l)
marks = [("loop head, dof rows, Hessian init", line_of("for (int it = 0; it < cap && __any(act); it++)")),
         ("contact pass: cone, A, cube block", line_of("for (int k = 0; k < NS; k++) if (sl[k].la >= 0) {", line_of("float gcp[6]"))),
         ("contact pass: chain part (columns, g, Hbb, Hcb atomics)", line_of("if (t.lb > 0) {", line_of("float gcp[6]"))),
         ("reduce-scatters, Hcc store", line_of("float gcl = 0.f;")),
         ("convergence test", line_of("// ---- (2) convergence")),
         ("chain blocks: factor, Y, zb (+ staged elimination)", line_of("float L[10], Linv[4], Ya[NLK]")),
         ("Schur complement + 6x6 + back-substitution", line_of("// Schur complement: Hcc[q][r] -=")),
         ("dense path", line_of("// ---- (4b) dense path")),
         ("line search set-up (M p, J p)", line_of("// ---- (5) exact line search along p")),
         ("line search loop (one evaluation)", line_of("float lo = 0.f, hi = -1.f, alpha = 1.f")),
         ("step", line_of("// ---- (6) step")),
         ("end", line_of("// (the rare slot-count copy instantiates"))]
out = subprocess.run([sys.executable, "tools/diag/isa_hot_loop.py", sys.argv[1], which], capture_output=True, text=True).stdout
# re-run the attribution with all lines: import the module's logic by exec is messy; parse a full dump instead
import importlib.util, io, contextlib
code = open("tools/diag/isa_hot_loop.py").read().replace("ins.most_common(45)", "ins.most_common(100000)")
buf = io.StringIO()
sys.argv = ["isa_hot_loop.py", sys.argv[1], which]
with contextlib.redirect_stdout(buf): exec(compile(code, "isa_hot_loop.py", "exec"), {"__name__": "__main__"})
rows = []
for l in buf.getvalue().split("\n"):
    m = re.match(r"\s+(\d+)\s+(\d+) \(.*?\) waits\s+(\d+) (\{.*?\})\s+\|", l)
    if m: rows.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), eval(m.group(4))))
print(buf.getvalue().split("\n")[0]); print(buf.getvalue().split("\n")[1])
helpers = collections.Counter()
tot = sum(r[1] for r in rows)
for i, (name, lo) in enumerate(marks[:-1]):
    hi = marks[i + 1][1]
    sel = [r for r in rows if lo <= r[0] < hi]
    c = collections.Counter()
    for r in sel: c.update(r[3])
    print(f"  {name:58s} lines {lo}-{hi - 1}: {sum(r[1] for r in sel):5d} ({100 * sum(r[1] for r in sel) / tot:4.1f} %)  lgkm waits {sum(r[2] for r in sel):2d}  {dict(c)}")
other = [r for r in rows if not (marks[0][1] <= r[0] < marks[-1][1])]
c = collections.Counter()
for r in other: c.update(r[3])
print(f"  {'helpers attributed to their own lines (link_c3, slot_Jx, cone_dir, chol4, ...)':58s}: {sum(r[1] for r in other):5d} ({100 * sum(r[1] for r in other) / tot:4.1f} %)  lgkm waits {sum(r[2] for r in other):2d}  {dict(c)}")
for r in sorted(other, key=lambda r: -r[1])[:14]: print(f"      {r[0]:5d} {r[1]:4d} waits {r[2]}  | {src[r[0] - 1].strip()[:100]}")
