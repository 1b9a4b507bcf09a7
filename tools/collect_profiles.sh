#!/bin/bash
# Collect the rocprofv3 evidence for a round on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r01
# Writes text summaries under gpurun_out/profiles_<tag>/ ; copy the ones to keep into profiles/.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph"
# 1. kernel trace + stats of the bench command (same command as the bench line, eager launches so every
#    kernel is a separate dispatch)
rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o kt -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/kt.err
python $R/tools/rocpd_summary.py /tmp/kt_$TAG/kt_results.db --skip 2 > $OUT/kernel_stats.txt
# 2. PMC passes (counters only, no other tracing domains): SQ activity, then HBM bytes in separate passes
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU \
    --kernel-trace -d /tmp/pmc1_$TAG -o p -- $BENCH > /dev/null 2> $OUT/pmc1.err
python $R/tools/rocpd_summary.py /tmp/pmc1_$TAG/p_results.db --filter k_ | sed -n '/counters/,$p' > $OUT/pmc_sq.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc2_$TAG -o p -- $BENCH > /dev/null 2> $OUT/pmc2.err
python $R/tools/rocpd_summary.py /tmp/pmc2_$TAG/p_results.db --filter k_ | sed -n '/counters/,$p' > $OUT/pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc3_$TAG -o p -- $BENCH > /dev/null 2> $OUT/pmc3.err
python $R/tools/rocpd_summary.py /tmp/pmc3_$TAG/p_results.db --filter k_ | sed -n '/counters/,$p' > $OUT/pmc_write.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d /tmp/pmc4_$TAG -o p -- $BENCH > /dev/null 2> $OUT/pmc4.err
python $R/tools/rocpd_summary.py /tmp/pmc4_$TAG/p_results.db --filter k_ | sed -n '/counters/,$p' > $OUT/pmc_l2.txt
# 3. the plain bench line (hipGraph, with cpu_baseline)
cd $R && python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls -la $OUT
cat $OUT/kernel_stats.txt $OUT/pmc_fetch.txt $OUT/pmc_write.txt
tail -c 2500 $OUT/bench.json
# 4. the other BASELINE configs and variants on this GPU (bench lines only)
cd $R
python bench.py --no-cpu-baseline --config c3 --steps 10 --warmup 3 > $OUT/bench_c3.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --config c4 --steps 5 --warmup 2 > $OUT/bench_c4.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --dmax 0.5 > $OUT/bench_c2_dmax0p5.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --dmax -1 > $OUT/bench_c2_unbounded.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --cutoff 104 > $OUT/bench_c2_tau104.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --cutoff 32 > $OUT/bench_c2_tau32.json 2>> $OUT/bench.err
python tools/e2e_time.py > $OUT/e2e_time.txt 2>&1
for f in $OUT/bench_c*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})"; done
cat $OUT/e2e_time.txt | tail -5
python bench.py --no-cpu-baseline --config c5 > $OUT/bench_c5.json 2>> $OUT/bench.err
python tools/e2e_batch_time.py > $OUT/e2e_batch_time.txt 2>&1
tail -c 400 $OUT/bench_c5.json; tail -2 $OUT/e2e_batch_time.txt
# 5. the sampled-pixel path: bench line, host-API timing, per-kernel device times
python bench.py --no-cpu-baseline --config c5s > $OUT/bench_c5s.json 2>> $OUT/bench.err
python tools/sample_time.py > $OUT/sample_time.txt 2>&1
(cd /tmp && SAMPLE_TIME_HOST=0 rocprofv3 --kernel-trace --stats -d /tmp/kts_$TAG -o kt -- python $R/tools/sample_time.py > /dev/null 2> $OUT/kts.err
 python $R/tools/rocpd_summary.py /tmp/kts_$TAG/kt_results.db --skip 2 > $OUT/sampled_kernel_stats.txt)
tail -c 600 $OUT/bench_c5s.json; cat $OUT/sample_time.txt | tail -7; head -14 $OUT/sampled_kernel_stats.txt
