#!/bin/bash
# Run a micro-benchmark binary over the BASELINE-shaped workloads:  bash tools/mbset.sh tools/bin/mb [label]
B=${1:-tools/bin/mb}; L=${2:-$(basename $B)}
run() { echo -n "$L $1: "; shift; timeout 120 $B "$@" | tail -1; }
run C2    256 256 4 0.1 0 30 1 6
run C2x16 256 256 4 0.1 0 20 16 6
run C3    512 512 12 0.1 0 10 1 6
run C4    1024 1024 8 0.1 0 5 1 6
run C5    192 192 4 0.5 0 20 16 6
