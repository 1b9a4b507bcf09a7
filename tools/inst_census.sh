#!/bin/bash
# Instruction census of the backward (and forward) kernels: wave-level instruction counts by type, per launch.
#   bash tools/inst_census.sh "<mb args>"   -> gpurun_out/inst_census.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=${1:-"256 256 4 0.1 0 20 1 134"}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/inst_census.txt
: > $OUT
i=0
for grp in \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
  "SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_EXP_GDS SQ_WAVES" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
  "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VALU_TRANS" ; do
  i=$((i+1))
  echo "== $grp" >> $OUT
  timeout 120 rocprofv3 --pmc $grp --kernel-trace -d /tmp/ic$i -o p -- $R/tools/bin/mb $ARGS > /dev/null 2> /tmp/ic$i.err
  [ -f /tmp/ic$i/p_results.db ] && python $R/tools/rocpd_summary.py /tmp/ic$i/p_results.db --filter k_render | sed -n '/counters/,$p' >> $OUT || { echo "(pass failed or timed out)" >> $OUT; tail -3 /tmp/ic$i.err >> $OUT; }
done
cat $OUT
