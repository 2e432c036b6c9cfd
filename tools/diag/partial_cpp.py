"""Partial evaluation of preprocessor conditionals: the branches decided by the given macro values are resolved (directive lines dropped, dead branches removed), every other
conditional stays as written.  usage: python tools/diag/partial_cpp.py file NAME=value ... [-UNAME ...] > out"""
import re, sys
path = sys.argv[1]
known = {}
for a in sys.argv[2:]:
    if a.startswith("-U"): known[a[2:]] = None
    else: k, v = a.split("="); known[k] = int(v)
def evaluate(expr):
    e = re.sub(r"//.*$", "", expr).strip()
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: ("1" if known.get(m.group(1)) is not None else "0") if m.group(1) in known else m.group(0), e)
    def sub(m):
        n = m.group(0)
        if n in known: return str(known[n] if known[n] is not None else 0)
        return n
    e2 = re.sub(r"\b[A-Za-z_]\w*\b", sub, e)
    if re.search(r"[A-Za-z_]", e2): return None
    py = e2.replace("&&", " and ").replace("||", " or ")
    py = re.sub(r"!(?!=)", " not ", py)
    return bool(eval(py))
lines = open(path).read().split("\n")
def parse(i, out):
    """copy lines from i until an #elif/#else/#endif of the enclosing group; returns the index of that directive"""
    while i < len(lines):
        l = lines[i]; m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", l)
        if not m: out.append(l); i += 1; continue
        d, rest = m.group(1), m.group(2)
        if d in ("elif", "else", "endif"): return i
        # a group starts here
        branches = []  # (directive line, value or None, body lines)
        cur_line = l
        if d == "if": val = evaluate(rest)
        else:
            name = re.sub(r"//.*$", "", rest).strip()
            val = None if name not in known else ((known[name] is not None) == (d == "ifdef"))
        i += 1
        while True:
            body = []; i = parse(i, body)
            branches.append((cur_line, val, body))
            m2 = re.match(r"\s*#\s*(elif|else|endif)\b(.*)", lines[i]); cur_line = lines[i]
            if m2.group(1) == "endif": endl = lines[i]; i += 1; break
            val = evaluate(m2.group(2)) if m2.group(1) == "elif" else True
            if m2.group(1) == "else": val = "else"
            i += 1
        if branches[0][1] is None:  # undecided: keep verbatim
            for bl, _, body in branches: out.append(bl); out.extend(body)
            out.append(endl)
        else:
            chosen = None
            for bl, v, body in branches:
                if v == "else" or v is True: chosen = body; break
                assert v is not None, "undecided #elif after a decided #if: " + bl
            if chosen is not None: out.extend(chosen)
    return i
res = []; parse(0, res)
sys.stdout.write("\n".join(res))
