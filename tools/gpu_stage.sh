#!/bin/bash
# One parameterised runner for the GPU box (replaces the per-experiment tools/run_r05*.sh of round 5):
#   gpurun --timeout 1500 -- 'bash tools/gpu_stage.sh <stage> [args]'      results under gpurun_out/<stage>/
# Stages: home_ab (home-tile backward against the default on the BASELINE-shaped workloads, all tile shapes),
#         home_whatif (what-if builds of that kernel: tools/build_mb.sh nolds|noeval|noreduce -DHM_EXP_...),
#         home_dists (the three backward kernels on the size distributions of tools/policy_sweep.sh),
#         tests <pytest args>, bench [bench.py args], collect <tag> (tools/collect_profiles.sh)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-home_ab}; shift
O=gpurun_out/$S; mkdir -p $O
export GSASR_SPLAT_DEV=1
mb() { local label=$1; shift; echo -n "$label: "; timeout 180 tools/bin/mb "$@" 2>&1 | tail -5; }
case $S in
home_ab)
  for v in 0 1 2; do
    export GSASR_SPLAT_HOME_VARIANT=$v
    MB_ALT_FLAGS=32768 mb "C2 v$v" 256 256 4 0.1 0 30 1 6
    MB_ALT_FLAGS=32768 mb "C2x16 v$v" 256 256 4 0.1 0 20 16 6
    MB_ALT_FLAGS=32768 mb "C5 v$v" 192 192 4 0.5 0 20 16 6
    MB_ALT_FLAGS=32768 mb "x2 v$v" 512 512 2 0.1 0 20 1 6
    MB_ALT_FLAGS=32768 mb "x4g4 v$v" 256 256 4 0.1 0 20 4 6
  done > $O/mb.txt 2>&1
  cat $O/mb.txt ;;
home_whatif)
  for b in mb mb_nolds mb_noeval mb_noreduce; do
    for cfg in "C2x16 0 256 256 4 0.1 0 20 16 6" "C5 0 192 192 4 0.5 0 20 16 6" "C2 1 256 256 4 0.1 0 30 1 6" "C2 2 256 256 4 0.1 0 30 1 6"; do
      set -- $cfg; name=$1; v=$2; shift 2
      echo -n "$b $name v$v: "; GSASR_SPLAT_HOME_VARIANT=$v MB_ALT_FLAGS=32768 timeout 120 tools/bin/$b "$@" 2>&1 | grep "alt flags\|g_sigmas" | tr '\n' ' ' | cut -c1-200; echo
    done
  done > $O/whatif.txt 2>&1
  cat $O/whatif.txt ;;
home_dists)
  for dist in 0 1 3 4 5 6; do
    for cfg in "C2x16 0 256 256 4 0.1 0 10 16 6" "C5 0 192 192 4 0.5 0 10 16 6" "x4g4 1 256 256 4 0.1 0 10 4 6"; do
      set -- $cfg; name=$1; v=$2; shift 2
      echo -n "dist$dist $name v$v: "; MB_DIST=$dist GSASR_SPLAT_HOME_VARIANT=$v MB_ALT_FLAGS=32768 timeout 120 tools/bin/mb "$@" 2>&1 | grep "alt flags\|g_coords" | tr '\n' ' ' | cut -c1-220; echo
      echo -n "dist$dist $name tile: "; MB_DIST=$dist GSASR_SPLAT_BWD=tile GSASR_SPLAT_LISTS=1 timeout 120 tools/bin/mb "$@" 2>&1 | grep -o "plan.*bwd [0-9.]* us"
    done
  done > $O/dists.txt 2>&1
  cat $O/dists.txt ;;
tests)
  timeout 1700 python -m pytest "$@" -q -x -m gpu 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt ;;
bench)
  python bench.py "$@" > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json; tail -3 $O/bench.err ;;
collect)
  bash tools/collect_profiles.sh "$@" > $O/collect.log 2>&1; tail -20 $O/collect.log ;;
*) echo "unknown stage $S"; exit 2 ;;
esac
