#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite) outputs as text: per-kernel stats and PMC counters.
usage: rocpd_summary.py <results.db> [...]  > profiles/<name>.txt"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if "distribution_elementwise_grid_stride_kernel" in name:
        return "at::native::distribution_elementwise_grid_stride_kernel<normal> (torch.randn)"
    return name if len(name) < 160 else name[:157] + "..."


for db in sys.argv[1:]:
    con = sqlite3.connect(db)
    cur = con.cursor()
    print(f"== {db}")
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    if rows:
        print(f"{'kernel':<100} {'calls':>6} {'total_us':>16} {'avg_us':>14} {'pct':>7}")
        for n, c, t, a, p in rows:
            print(f"{short(n):<100} {c:>6} {t:>16.1f} {a:>14.1f} {p:>7.3f}")
    if rows:  # the dominant kernel, dispatch by dispatch (the bench times its last K launches: warm-up launches come first)
        top = max(rows, key=lambda r: r[2])[0]
        try:
            d = [r[0] / 1e3 for r in cur.execute("select duration from kernels where name = ? order by start", (top,))]
            print(f"dispatches of {short(top)[:60]} [us]: " + " ".join(f"{x:.0f}" for x in d))
            for k in (3, 20):
                if len(d) > k:
                    print(f"  mean of the last {k}: {sum(d[-k:]) / k:.1f} us")
        except sqlite3.OperationalError:
            pass
    try:
        rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(vgpr_count), avg(accum_vgpr_count), avg(sgpr_count), avg(scratch_size), avg(lds_block_size) from counters_collection group by kernel_name, counter_name order by avg(value) desc"))
    except sqlite3.OperationalError:
        rows = []
    if rows:
        print(f"{'kernel':<100} {'counter':>12} {'n':>3} {'avg':>16} {'min':>16} {'max':>16}  vgpr/agpr/sgpr/scratch/lds")
        for n, cn, k, a, mn, mx, vg, ag, sg, sc, lds in rows:
            print(f"{short(n):<100} {cn:>12} {k:>3} {a:>16.1f} {mn:>16.1f} {mx:>16.1f}  {vg:.0f}/{ag:.0f}/{sg:.0f}/{sc:.0f}/{lds:.0f}")
    print()
