#!/bin/bash
# rocprofv3 counter passes over the microbenchmark (one pass per group; counters only + kernel trace):
#   bash tools/pmc_probe.sh "<mb args>" "<kernel filter>"  -> gpurun_out/pmc_probe.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=${1:-"256 256 4 0.1 0 20 1 6"}
FILT=${2:-k_render}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_probe.txt
: > $OUT
i=0
for grp in \
  "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
  "SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" ; do
  i=$((i+1))
  echo "== $grp" >> $OUT
  timeout 90 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pp$i -o p -- $R/tools/bin/mb $ARGS > /dev/null 2> /tmp/pp$i.err
  [ -f /tmp/pp$i/p_results.db ] && python $R/tools/rocpd_summary.py /tmp/pp$i/p_results.db --filter $FILT | sed -n '/counters/,$p' >> $OUT || echo "(pass failed or timed out)" >> $OUT
done
cat $OUT
