"""Default kernel choice against the best forced combination: reads tools/policy_sweep.sh output, prints one row per
(distribution, shape): default step time, best combination and its time, regret = default / best - 1, and per stage the best."""
import re
import sys
from collections import defaultdict

rows = defaultdict(dict)
pat = re.compile(r"^(dist\d+) (\S+) (\S+) : .*plan ([\d.]+) us\s+fwd ([\d.]+) us\s+bwd ([\d.]+) us.*reach=(\d+),(\d+) maxcell=(\d+)")
for line in open(sys.argv[1]):
    m = pat.match(line)
    if m:
        d, shape, combo, p, f, b, rx, ry, mc = m.groups()
        rows[(d, shape)][combo] = (float(p), float(f), float(b), int(rx), int(ry), int(mc))
names = {"dist0": "SURVEY 8(d)", "dist1": "small", "dist2": "saturated", "dist3": "log-uniform", "dist4": "needles", "dist5": "bimodal", "dist6": "jitter 3 LR px"}
print(f"{'distribution':14s} {'shape':6s} {'reach':>7s} {'default us':>10s} (plan/fwd/bwd)        {'best combination':24s} {'us':>8s} {'regret':>7s}")
worst = 0.0
for (d, shape), r in sorted(rows.items()):
    if "default" not in r:
        continue
    dp, df, db, rx, ry, mc = r["default"]
    tot = {k: v[0] + v[1] + v[2] for k, v in r.items()}
    best = min((k for k in tot if k != "default"), key=lambda k: tot[k], default=None)
    if best is None:
        continue
    reg = tot["default"] / tot[best] - 1.0
    worst = max(worst, reg)
    print(f"{names.get(d, d):14s} {shape:6s} {rx:3d},{ry:3d} {tot['default']:10.1f} ({dp:6.1f}/{df:7.1f}/{db:7.1f})  {best:24s} {tot[best]:8.1f} {100 * reg:6.1f}%")
print(f"worst regret {100 * worst:.1f}%")
