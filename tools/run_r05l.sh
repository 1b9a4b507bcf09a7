cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_full.txt; cat $O/pytest_full.txt
rm -rf gpurun_out/profiles_r05
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1
tail -30 $O/collect.log
