"""Stated tolerance against observed error of every recorded parity assertion (gpurun_out/test_margins.jsonl, written by `JUDO_RECORD_MARGINS=1 pytest -m gpu`:
tests/conftest.py wraps np.testing.assert_allclose and provides `bounded`).  Prints, per call site, the worst fraction of the tolerance that was used over the run; a
fraction below 0.2 means the stated tolerance is more than 5x the observed error.  usage: python tools/diag/margin_report.py [file]"""
import collections, json, sys
f = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/test_margins.jsonl"
ac = collections.defaultdict(lambda: dict(used=0.0, max_abs=0.0, n=0)); bd = collections.defaultdict(lambda: dict(obs=0.0, n=0))
for l in open(f):
    r = json.loads(l)
    if r.get("test") == "allclose":
        d = ac[(r["site"], r["rtol"], r["atol"])]; d["used"] = max(d["used"], r["used"]); d["max_abs"] = max(d["max_abs"], r["max_abs"]); d["n"] += 1
    elif r.get("test") == "bounded":
        d = bd[(r["site"], r["what"], r["bound"])]; d["obs"] = max(d["obs"], r["observed"]); d["n"] += 1
print("# assert_allclose sites: fraction of the tolerance used (max over calls), largest |diff|, rtol, atol")
for (site, rtol, atol), d in sorted(ac.items(), key=lambda kv: kv[1]["used"]):
    print(f"{site:32s} used {d['used']:9.3g}  max|diff| {d['max_abs']:9.3g}  rtol {rtol:g} atol {atol:g}  calls {d['n']}")
print("# bounded sites: observed / bound")
for (site, what, bound), d in sorted(bd.items(), key=lambda kv: kv[1]["obs"] / kv[0][2] if kv[0][2] else 0):
    print(f"{site:32s} {d['obs'] / bound if bound else float('nan'):9.3g}  observed {d['obs']:9.3g}  bound {bound:g}  calls {d['n']}  | {what}")
