"""Set the stated tolerances of the GPU parity tests from measurements: every np.testing.assert_allclose / bounded(...) site recorded in gpurun_out/test_margins.jsonl
(JUDO_RECORD_MARGINS=1 pytest -m gpu) whose observed error used less than a fifth of its tolerance gets the tolerance 5 x observed, rounded UP to {1, 1.5, 2, 3, 5, 7} x 10^k.
Only tightens; sites that observed exactly zero, sites whose tolerance differs from call to call, and lines it cannot parse unambiguously are listed and left alone.
usage: python tools/diag/tighten_tolerances.py [margins.jsonl] [--apply]"""
import collections, json, math, re, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
f = args[0] if args else "gpurun_out/test_margins.jsonl"
apply = "--apply" in sys.argv
FACTOR = 5.0
def round_up(x):
    if x <= 0: return 0.0
    k = math.floor(math.log10(x)); m = x / 10 ** k
    for c in (1, 1.5, 2, 3, 5, 7, 10):
        if m <= c * (1 + 1e-12): return c * 10 ** k
ac = collections.defaultdict(lambda: dict(used=0.0, tols=set())); bd = collections.defaultdict(lambda: dict(obs=0.0, bounds=set()))
for l in open(f):
    r = json.loads(l)
    if r.get("test") == "allclose":
        d = ac[r["site"]]; d["used"] = max(d["used"], r["used"]); d["tols"].add((r["rtol"], r["atol"]))
    elif r.get("test") == "bounded":
        d = bd[(r["site"], r["what"])]; d["obs"] = max(d["obs"], r["observed"]); d["bounds"].add(r["bound"])
num = r"[0-9]+(?:\.[0-9]*)?(?:e-?[0-9]+)?"
def fmt(x): return f"{x:.3g}".replace("e-0", "e-").replace("e+0", "e")
edits = collections.defaultdict(dict); skipped = []
for site, d in ac.items():
    fn, ln = site.split(":"); ln = int(ln)
    if len(d["tols"]) != 1: skipped.append((site, "tolerance differs between calls")); continue
    if d["used"] == 0: skipped.append((site, "observed exactly zero")); continue
    if d["used"] * FACTOR >= 1: continue
    rt0, at0 = next(iter(d["tols"]))
    if rt0 <= 1e-8 and at0 <= 1e-8: skipped.append((site, "fp64 host comparison (stated tolerance <= 1e-8): not a kernel tolerance")); continue
    scale = round_up(d["used"] * FACTOR)
    edits[fn][ln] = ("allclose", scale, d)
for (site, what), d in bd.items():
    fn, ln = site.split(":"); ln = int(ln)
    if len(d["bounds"]) != 1: skipped.append((site, "bound differs between calls")); continue
    if d["obs"] == 0: skipped.append((site + " " + what, "observed exactly zero")); continue
    b = next(iter(d["bounds"]))
    if d["obs"] * FACTOR >= b: continue
    edits[fn].setdefault(ln, ("bounded", {}, None))[1][what] = (b, round_up(d["obs"] * FACTOR))
n = 0
for fn, byline in sorted(edits.items()):
    path = "tests/" + fn; src = open(path).read().split("\n")
    for ln, e in sorted(byline.items()):
        line = src[ln - 1]; new = line
        if e[0] == "allclose":
            rt = re.findall(rf"rtol=({num})", line); at = re.findall(rf"atol=({num})", line)
            if len(rt) > 1 or len(at) > 1 or (not rt and not at) or line.count("assert_allclose") != 1: skipped.append((f"{fn}:{ln}", "cannot parse the tolerances of this line")); continue
            if rt and float(rt[0]) > 0: new = re.sub(rf"rtol={num}", "rtol=" + fmt(float(rt[0]) * e[1]), new)
            if at and float(at[0]) > 0: new = re.sub(rf"atol={num}", "atol=" + fmt(float(at[0]) * e[1]), new)
        else:
            for what, (b, nb) in e[1].items():
                pat = 'bounded("' + what + '", '
                i = new.find(pat)
                if i < 0 or new.count(pat) != 1: skipped.append((f"{fn}:{ln} {what}", "cannot find the bounded(...) call")); continue
                # the bound is the last argument of this call: find the matching parenthesis
                j = i + len("bounded("); depth = 1
                while depth: depth += {"(": 1, ")": -1}.get(new[j], 0); j += 1
                call = new[i:j]; k = call.rfind(", ")
                if not re.fullmatch(num, call[k + 2:-1]): skipped.append((f"{fn}:{ln} {what}", "bound is not a literal")); continue
                new = new[:i] + call[:k + 2] + fmt(nb) + ")" + new[j:]
        if new != line:
            n += 1; print(f"{fn}:{ln}\n  - {line.strip()[:220]}\n  + {new.strip()[:220]}")
            src[ln - 1] = new
    if apply: open(path, "w").write("\n".join(src))
print(f"# {n} lines {'rewritten' if apply else 'would change'}; left alone:")
for s in skipped: print("#  ", *s)
