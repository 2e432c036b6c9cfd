#!/bin/bash
# GPU box: kernel + copy timeline of cartpole plan steps (rocprofv3 kernel + memory-copy trace) -> start offsets and durations of one steady-state step.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o x -- python $GRAFT_REPO_ROOT/bench.py --task cartpole --no-cpu-baseline --steps 20 --warmup 3 > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob("/tmp/tl/**/*.db", recursive=True)[0]
con = sqlite3.connect(db); cur = con.cursor()
ev = [(s, e, n[:60]) for n, s, e in cur.execute("select name, start, end from kernels")]
try:
    ev += [(s, e, "COPY " + str(n)) for n, s, e in cur.execute("select name, start, end from memory_copies")]
except Exception as ex:
    print("no memory_copies view:", ex)
ev.sort()
tail = ev[-40:]
t0 = tail[0][0]
for s, e, n in tail:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n}")
PY
