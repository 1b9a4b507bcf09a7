#!/bin/bash
# Build the micro-benchmark with exactly the library's compiler flags (gsasr_amd/build.py HIPCC_FLAGS).
#   bash tools/build_mb.sh                 -> tools/bin/mb
#   bash tools/build_mb.sh half -DGSASR_SRC='"/tmp/half.hip"'   -> tools/bin/mb_half (variant source)
set -e
cd "$(dirname "$0")/.."
name=mb
if [ $# -gt 0 ] && [[ "$1" != -* ]]; then name=mb_$1; shift; fi
mkdir -p tools/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -fvisibility=hidden -Wno-unused-function \
      -Iinclude -Igsasr_amd/csrc "$@" tools/mb.hip -o tools/bin/$name
echo tools/bin/$name
