#!/bin/bash
# Instruction-fetch evidence for the render kernels: SQC instruction-cache requests / hits / misses and
# the SQ's instruction-fetch counters, per kernel, over the micro-benchmark (counters only + kernel trace, one pass per group).
#   bash tools/icache_probe.sh "<mb args>" [binary]   -> stdout (tee it into gpurun_out/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=${1:-"256 256 4 0.1 0 20 1 6"}
BIN=${2:-$R/tools/bin/mb}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
  "SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  echo "== $grp   ($BIN $ARGS)"
  timeout 120 rocprofv3 --pmc $grp --kernel-trace -d /tmp/ic$i -o p -- $BIN $ARGS > /dev/null 2> /tmp/ic$i.err
  [ -f /tmp/ic$i/p_results.db ] && python $R/tools/rocpd_summary.py /tmp/ic$i/p_results.db --filter k_render | sed -n '/counters/,$p' || { echo "(pass failed or timed out)"; tail -3 /tmp/ic$i.err; }
  rm -rf /tmp/ic$i
done
