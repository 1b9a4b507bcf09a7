#!/bin/bash
# Collect the rocprofv3 evidence for a round on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02
# Writes text summaries under gpurun_out/profiles_<tag>/ ; copy the ones to keep into profiles/.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GSASR_SPLAT_DEV=1   # the GSASR_SPLAT_* A/B switches below are read only with it
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-extras --no-live-pmc"
sq="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU"
pass() {   # pass <name> "<counters>" <command...>: one counters-only run -> $OUT/<name>.txt
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/${name}_$TAG -o p -- "$@" > /dev/null 2> $OUT/$name.err
  python $R/tools/rocpd_summary.py /tmp/${name}_$TAG/p_results.db --filter k_ | sed -n '/counters/,$p' > $OUT/$name.txt
}
# 1. kernel trace + stats of the bench command (same command as the bench line, eager launches so every
#    kernel is a separate dispatch)
rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o kt -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/kt.err
python $R/tools/rocpd_summary.py /tmp/kt_$TAG/kt_results.db --skip 2 > $OUT/kernel_stats.txt
# 2. PMC passes (counters only, no other tracing domains): SQ activity, then HBM bytes in separate passes
pass pmc_sq "$sq" $BENCH
pass pmc_fetch "FETCH_SIZE" $BENCH
pass pmc_write "WRITE_SIZE" $BENCH
pass pmc_l2 "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" $BENCH
# 3. the same for the exact-semantics run (tau = 104: the reference's set of non-zero fp32 terms) and for the
#    tile-stationary backward (GSASR_SPLAT_BWD=tile: development switch of the library)
rocprofv3 --kernel-trace --stats -d /tmp/kt104_$TAG -o kt -- $BENCH --cutoff 104 > /dev/null 2> $OUT/kt104.err
python $R/tools/rocpd_summary.py /tmp/kt104_$TAG/kt_results.db --skip 2 > $OUT/kernel_stats_tau104.txt
export GSASR_SPLAT_BWD=tile
rocprofv3 --kernel-trace --stats -d /tmp/ktt_$TAG -o kt -- $BENCH > /dev/null 2> $OUT/ktt.err
python $R/tools/rocpd_summary.py /tmp/ktt_$TAG/kt_results.db --skip 2 > $OUT/kernel_stats_tile_bwd.txt
pass pmc_sq_tile_bwd "$sq" $BENCH
pass pmc_l2_tile_bwd "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" $BENCH
pass pmc_fetch_tile_bwd "FETCH_SIZE" $BENCH
pass pmc_write_tile_bwd "WRITE_SIZE" $BENCH
rocprofv3 --kernel-trace --stats -d /tmp/ktt4_$TAG -o kt -- $BENCH --config c4 --steps 5 --warmup 2 > /dev/null 2> $OUT/ktt4.err
python $R/tools/rocpd_summary.py /tmp/ktt4_$TAG/kt_results.db --skip 1 > $OUT/kernel_stats_c4_tile_bwd.txt
export GSASR_SPLAT_BWD=gaussian
rocprofv3 --kernel-trace --stats -d /tmp/ktg4_$TAG -o kt -- $BENCH --config c4 --steps 5 --warmup 2 > /dev/null 2> $OUT/ktg4.err
python $R/tools/rocpd_summary.py /tmp/ktg4_$TAG/kt_results.db --skip 1 > $OUT/kernel_stats_c4_gaussian_bwd.txt
unset GSASR_SPLAT_BWD
rocprofv3 --kernel-trace --stats -d /tmp/kt4_$TAG -o kt -- $BENCH --config c4 --steps 5 --warmup 2 > /dev/null 2> $OUT/kt4.err
python $R/tools/rocpd_summary.py /tmp/kt4_$TAG/kt_results.db --skip 1 > $OUT/kernel_stats_c4.txt
# 3b. kernel traces of the other BASELINE configs (config 3: forward only; config 5: the batched canvas)
if [ -z "${ONLY_EXTRA_TRACES:-}" ] || true; then
rocprofv3 --kernel-trace --stats -d /tmp/kt3_$TAG -o kt -- $BENCH --config c3 --steps 10 --warmup 3 > /dev/null 2> $OUT/kt3.err
python $R/tools/rocpd_summary.py /tmp/kt3_$TAG/kt_results.db --skip 1 > $OUT/kernel_stats_c3.txt
rocprofv3 --kernel-trace --stats -d /tmp/kt5_$TAG -o kt -- $BENCH --config c5 > /dev/null 2> $OUT/kt5.err
python $R/tools/rocpd_summary.py /tmp/kt5_$TAG/kt_results.db --skip 1 > $OUT/kernel_stats_c5.txt
fi
# 3c. GSASR's real Gaussian density at inference size (16 per LR pixel)
rocprofv3 --kernel-trace --stats -d /tmp/kt216_$TAG -o kt -- $BENCH --config c2x16 --steps 10 --warmup 3 > /dev/null 2> $OUT/kt216.err
python $R/tools/rocpd_summary.py /tmp/kt216_$TAG/kt_results.db --skip 1 > $OUT/kernel_stats_c2x16.txt
# 3d. HBM counter passes of the x12, x8 and dense legs (forward / backward / gather bytes per launch -> pmc_latest.json "configs")
for cfg in c3 c4 c2x16; do
  pass pmc_fetch_$cfg "FETCH_SIZE" $BENCH --config $cfg --steps 5 --warmup 2
  pass pmc_write_$cfg "WRITE_SIZE" $BENCH --config $cfg --steps 5 --warmup 2
  pass pmc_sq_$cfg "$sq" $BENCH --config $cfg --steps 5 --warmup 2
done
# 4. the plain bench line (with exact / dropin / cpu_baseline / the x12, x8 and 16-per-LR-pixel legs)
cd $R && python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/kernel_stats.txt $OUT/pmc_fetch.txt $OUT/pmc_write.txt
tail -c 3000 $OUT/bench.json
# 5. the other BASELINE configs and variants on this GPU (bench lines only)
python bench.py --no-cpu-baseline --config c3 --steps 10 --warmup 3 > $OUT/bench_c3.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --config c4 --steps 5 --warmup 2 > $OUT/bench_c4.json 2>> $OUT/bench.err
GSASR_SPLAT_BWD=tile python bench.py --no-cpu-baseline --config c4 --steps 5 --warmup 2 > $OUT/bench_c4_tile_bwd.json 2>> $OUT/bench.err
GSASR_SPLAT_BWD=gaussian python bench.py --no-cpu-baseline --config c4 --steps 5 --warmup 2 > $OUT/bench_c4_gaussian_bwd.json 2>> $OUT/bench.err
GSASR_SPLAT_BWD=tile python bench.py --no-cpu-baseline --no-extras > $OUT/bench_c2_tile_bwd.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --no-extras --dmax 0.5 > $OUT/bench_c2_dmax0p5.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --no-extras --dmax -1 > $OUT/bench_c2_unbounded.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --no-extras --cutoff 104 > $OUT/bench_c2_tau104.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --no-extras --cutoff -1 --steps 10 --warmup 3 > $OUT/bench_c2_nocull.json 2>> $OUT/bench.err
python tools/e2e_time.py > $OUT/e2e_time.txt 2>&1
for f in $OUT/bench_c*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})"; done
cat $OUT/e2e_time.txt | grep fused
python bench.py --no-cpu-baseline --config c5 > $OUT/bench_c5.json 2>> $OUT/bench.err
python bench.py --config c5e2e --steps 20 --warmup 5 > $OUT/bench_c5e2e.json 2>> $OUT/bench.err
python tools/e2e_batch_time.py > $OUT/e2e_batch_time.txt 2>&1
tail -c 400 $OUT/bench_c5.json; tail -2 $OUT/e2e_batch_time.txt
# 6. the sampled-pixel path: bench line, host-API timing, per-kernel device times
python bench.py --no-cpu-baseline --config c5s > $OUT/bench_c5s.json 2>> $OUT/bench.err
python tools/sample_time.py > $OUT/sample_time.txt 2>&1
tail -c 600 $OUT/bench_c5s.json; cat $OUT/sample_time.txt | tail -7
# 7. multi-rank code path self-test at world size 1 (RCCL init, band exchange with itself)
python bench.py --no-cpu-baseline --force-dist --steps 10 --warmup 3 > $OUT/bench_force_dist.json 2>> $OUT/bench.err
tail -c 700 $OUT/bench_force_dist.json
# 8. host path: the reference's calling convention and PyTorch's own floor under it; gradient error vs conditioning
python tools/dropin_profile.py > $OUT/dropin_profile.txt 2>&1
python tools/autograd_floor.py > $OUT/autograd_floor.txt 2>&1
python tools/rho_conditioning.py > $OUT/rho_conditioning.txt 2>&1
python bench.py --no-cpu-baseline --config c2x16 > $OUT/bench_c2x16.json 2>> $OUT/bench.err
cat $OUT/dropin_profile.txt $OUT/autograd_floor.txt | grep -v amdgpu.ids

# 9. round 5: the reference's published workload, the tile lists against the search, what-if builds, hipGraph replay against eager
cd /tmp
for c in 0 -1; do
  rocprofv3 --kernel-trace --stats -d /tmp/pub${c}_$TAG -o kt -- python $R/tools/published_run.py --cutoff $c 2> /dev/null | grep "ms per plan" > $OUT/published_run_$c.txt
  python $R/tools/rocpd_summary.py /tmp/pub${c}_$TAG/kt_results.db --skip 2 > $OUT/kernel_stats_published_$c.txt
done
pass pmc_sq_published "$sq" python $R/tools/published_run.py --cutoff 0 --calls 3
cd $R
for cfg in c2 c2x16 c5; do
  python bench.py --no-cpu-baseline --no-extras --no-live-pmc --no-graph --config $cfg > $OUT/bench_${cfg}_default.json 2>> $OUT/bench.err
  GSASR_SPLAT_LISTS=0 python bench.py --no-cpu-baseline --no-extras --no-live-pmc --no-graph --config $cfg > $OUT/bench_${cfg}_search.json 2>> $OUT/bench.err
  GSASR_SPLAT_LISTS=1 python bench.py --no-cpu-baseline --no-extras --no-live-pmc --no-graph --config $cfg > $OUT/bench_${cfg}_lists.json 2>> $OUT/bench.err
done
for f in $OUT/bench_c*_default.json $OUT/bench_c*_search.json $OUT/bench_c*_lists.json; do echo -n "$(basename $f) "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})"; done > $OUT/lists_ab.txt
cat $OUT/lists_ab.txt
bash tools/whatif.sh > $OUT/whatif.txt 2>&1; cat $OUT/whatif.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_exp.hip -o /tmp/mfma_exp && /tmp/mfma_exp > $OUT/mfma_exp.txt 2>&1
# hipGraph replay against eager launches of the same five-kernel step: per-kernel durations and the idle gaps in front of them
cd /tmp
rocprofv3 --kernel-trace -d /tmp/tl_eager_$TAG -o kt -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras --no-live-pmc --no-graph > /dev/null 2>&1
rocprofv3 --kernel-trace -d /tmp/tl_graph_$TAG -o kt -- python $R/tools/graph_replay.py > /dev/null 2>&1
(echo "== eager launches"; python $R/tools/timeline.py /tmp/tl_eager_$TAG/kt_results.db; echo "== hipGraph replay of the same step"; python $R/tools/timeline.py /tmp/tl_graph_$TAG/kt_results.db) > $OUT/graph_vs_eager_timeline.txt 2>&1
cat $OUT/graph_vs_eager_timeline.txt
