cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
rm -rf gpurun_out/profiles_r05
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1
tail -40 $O/collect.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_full.txt; cat $O/pytest_full.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
