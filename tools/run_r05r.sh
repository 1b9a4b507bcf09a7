#!/bin/bash
# tile-stationary backward from lists at GSASR's real density: where does its time go?
E1="GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD=tile GSASR_SPLAT_LISTS=1"
bash tools/prof_mb.sh c2x16_tile "$E1" "256 256 4 0.1 0 10 16 6" k_ > /dev/null 2>&1
bash tools/prof_mb.sh c2x16_gauss "GSASR_SPLAT_DEV=1" "256 256 4 0.1 0 10 16 6" k_ > /dev/null 2>&1
bash tools/prof_mb.sh c5_tile "$E1" "192 192 4 0.5 0 10 16 6" k_ > /dev/null 2>&1
for t in c2x16_tile c2x16_gauss c5_tile; do echo "=== $t"; cat gpurun_out/prof_$t/mb.txt | cut -c1-150; grep -v "^$" gpurun_out/prof_$t/kernel_stats.txt | head -12 | cut -c1-160; cat gpurun_out/prof_$t/pmc_sq.txt gpurun_out/prof_$t/pmc_lds.txt | cut -c1-200; done
