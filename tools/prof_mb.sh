#!/bin/bash
# kernel-trace stats + SQ counters of the micro-benchmark's kernels (gpurun from the repo root):
#   bash tools/prof_mb.sh <tag> "<env>" "<mb args>" [kernel filter]
TAG=$1; ENVS=$2; ARGS=${3:-"256 256 4 0.1 0 20 1 6"}; FILT=${4:-k_}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env $ENVS rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o kt -- $R/tools/bin/mb $ARGS > $OUT/mb.txt 2> $OUT/kt.err
python $R/tools/rocpd_summary.py /tmp/kt_$TAG/kt_results.db --skip 2 > $OUT/kernel_stats.txt
env $ENVS rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS \
    --kernel-trace -d /tmp/pmc1_$TAG -o p -- $R/tools/bin/mb $ARGS > /dev/null 2> $OUT/pmc1.err
python $R/tools/rocpd_summary.py /tmp/pmc1_$TAG/p_results.db --filter $FILT | sed -n '/counters/,$p' > $OUT/pmc_sq.txt
env $ENVS rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU \
    --kernel-trace -d /tmp/pmc2_$TAG -o p -- $R/tools/bin/mb $ARGS > /dev/null 2> $OUT/pmc2.err
python $R/tools/rocpd_summary.py /tmp/pmc2_$TAG/p_results.db --filter $FILT | sed -n '/counters/,$p' > $OUT/pmc_lds.txt
cat $OUT/mb.txt; cat $OUT/kernel_stats.txt; cat $OUT/pmc_sq.txt $OUT/pmc_lds.txt
