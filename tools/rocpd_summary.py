#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace and/or --pmc run) as plain text.

    python tools/rocpd_summary.py gpurun_out/prof/kt_results.db [--filter k_render]

Prints per-kernel call count / total / average / min / max duration (what `--stats` reports) and, if
counters were collected, the per-kernel average of every counter.
"""
import argparse
import sqlite3
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--filter", default="")
    ap.add_argument("--skip", type=int, default=0, help="ignore the first N dispatches of each kernel")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end, dispatch_id from kernels order by start").fetchall()
    per = defaultdict(list)
    for n, s, e, d in rows:
        per[n].append((e - s, d))
    total = sum(x for v in per.values() for x, _ in v[a.skip:])
    print(f"{'kernel':60s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for n, v in sorted(per.items(), key=lambda kv: -sum(x for x, _ in kv[1])):
        if a.filter and a.filter not in n:
            continue
        d = [x for x, _ in v[a.skip:]] or [0]
        short = n.replace("(anonymous namespace)::", "")[:60]
        print(f"{short:60s} {len(d):6d} {sum(d)/1e3:11.1f} {sum(d)/len(d)/1e3:9.2f} {min(d)/1e3:9.2f} {max(d)/1e3:9.2f} "
              f"{100.0*sum(d)/max(total,1):6.1f}")
    try:
        pm = db.execute("select * from pmc_events limit 1").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        pcols = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
        kcol = "name" if "name" in pcols else ("kernel_name" if "kernel_name" in pcols else None)
        ccol = "counter_name" if "counter_name" in pcols else "pmc_name"
        vcol = "value" if "value" in pcols else "counter_value"
        q = f"select {kcol}, {ccol}, avg({vcol}), count(*) from pmc_events group by {kcol}, {ccol}"
        print("\ncounters (average per dispatch):")
        cur = None
        for n, c, v, k in db.execute(q):
            if a.filter and a.filter not in (n or ""):
                continue
            if n != cur:
                print(" ", (n or "?").replace("(anonymous namespace)::", "")[:90])
                cur = n
            print(f"      {c:28s} {v:16.1f}   (n={k})")


if __name__ == "__main__":
    main()
