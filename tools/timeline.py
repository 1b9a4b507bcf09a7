#!/usr/bin/env python
"""Per-step timeline from a rocprofv3 kernel-trace database: kernel durations and the idle gaps between consecutive
dispatches (development aid).   python tools/timeline.py kt_results.db [first_kernel_substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "k_classify"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
steps, cur = [], []
for n, s, e in rows:
    if first in n and cur:
        steps.append(cur)
        cur = []
    cur.append((n.replace("(anonymous namespace)::", "")[:28], s, e))
steps.append(cur)
steps = [st for st in steps if first in st[0][0]][5:-2]
acc = {}
for st in steps:
    for i, (n, s, e) in enumerate(st):
        gap = (s - st[i - 1][2]) if i else 0
        a = acc.setdefault((i, n), [0.0, 0.0, 0])
        a[0] += (e - s) / 1e3
        a[1] += gap / 1e3
        a[2] += 1
tot = 0.0
for (i, n), (d, g, k) in sorted(acc.items()):
    print(f"{i:2d} {n:30s} dur {d / k:7.2f} us   gap before {g / k:6.2f} us   (n={k})")
    tot += (d + g) / k
span = sum((st[-1][2] - st[0][1]) / 1e3 for st in steps) / len(steps)
print(f"step span (first start -> last end): {span:.2f} us over {len(steps)} steps")
