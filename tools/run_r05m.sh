#!/bin/bash
# round 5, run 13: backward what-ifs (pair-planar gradient, no sweep, no reduction)
mkdir -p gpurun_out/r05m
for v in base pairplanar nosweep noreduce nosweep_noreduce noload_nomath base; do
  bash tools/mbset.sh tools/bin/mb_$v $v
done > gpurun_out/r05m/whatif_bwd.txt 2>&1
cat gpurun_out/r05m/whatif_bwd.txt | cut -c1-120
