#!/bin/bash
# inference (forward-only plans): the default against {8x16 | 16x16 forward} x {lists | search}
mkdir -p gpurun_out/r05ac
for sh in "x4_1024 256 256 4 0.1 0 20 1 70" "x4d16_512 128 128 4 0.1 0 20 16 70" "x4d16_1024 256 256 4 0.1 0 10 16 70" "x2_1024 512 512 2 0.1 0 20 1 70" "x3_1536 512 512 3 0.1 0 10 1 70" "x6_3072 512 512 6 0.1 0 10 1 70" "x8_3072 384 384 8 0.1 0 10 1 70" "x12_6144 512 512 12 0.1 0 5 1 70" "x8d16_2048 256 256 8 0.1 0 5 16 70" "x4_512 128 128 4 0.1 0 20 1 70"; do
  set -- $sh; name=$1; shift
  for dist in 0 1 3; do
    echo -n "dist$dist $name default : "; MB_DIST=$dist tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/  bwd.*reach/ reach/; s/ maxcell.*//'
    for w in 0 1; do for l in 0 1; do
      echo -n "dist$dist $name wide$w-lists$l : "; MB_DIST=$dist GSASR_SPLAT_DEV=1 GSASR_SPLAT_FWD_WIDE=$w GSASR_SPLAT_LISTS=$l tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/  bwd.*reach/ reach/; s/ maxcell.*//'
    done; done
  done
done | tee gpurun_out/r05ac/inference_sweep.txt
python - <<'P' | tee gpurun_out/r05ac/inference_regret.txt
import re
from collections import defaultdict
rows=defaultdict(dict)
for l in open('gpurun_out/r05ac/inference_sweep.txt'):
    m=re.match(r"^(dist\d) (\S+) (\S+) : plan ([\d.]+) us\s+fwd ([\d.]+) us", l)
    if m: rows[(m.group(2),m.group(1))][m.group(3)]=(float(m.group(4)),float(m.group(5)))
print(f"{'shape':12s} {'dist':6s} {'default plan+fwd':>17s}   {'best':14s} {'us':>8s} {'regret':>7s}")
for (s,d),r in sorted(rows.items()):
    if 'default' not in r: continue
    t={k:v[0]+v[1] for k,v in r.items()}
    best=min((k for k in t if k!='default'), key=lambda k:t[k])
    print(f"{s:12s} {d:6s} {t['default']:8.1f} ({r['default'][0]:.1f}+{r['default'][1]:.1f})   {best:14s} {t[best]:8.1f} {100*(t['default']/t[best]-1):6.1f}%")
P
