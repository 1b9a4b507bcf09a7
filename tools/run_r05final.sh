cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/smoke.txt; cat $O/smoke.txt
( time python bench.py > $O/bench_default_line.json 2> $O/bench_default.err ) 2> $O/bench_default_time.txt; tail -3 $O/bench_default_time.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_full.txt; cat $O/pytest_full.txt
rm -rf gpurun_out/profiles_r05
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1
tail -12 $O/collect.log
