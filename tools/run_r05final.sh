cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/smoke.txt; cat $O/smoke.txt
( time python bench.py > $O/bench_default_line.json 2> $O/bench_default.err ) 2> $O/bench_default_time.txt; tail -3 $O/bench_default_time.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_full.txt; cat $O/pytest_full.txt
rm -rf gpurun_out/profiles_r05
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1
tail -12 $O/collect.log
# kernels of the shapes the round-5 rules changed (micro-benchmark under rocprofv3)
bash tools/prof_mb.sh x2_1024 "A=1" "512 512 2 0.1 0 10 1 6" k_ > /dev/null 2>&1
bash tools/prof_mb.sh x4_512 "A=1" "128 128 4 0.1 0 20 1 6" k_ > /dev/null 2>&1
bash tools/prof_mb.sh d16_512 "A=1" "128 128 4 0.1 0 20 16 6" k_ > /dev/null 2>&1
for t in x2_1024 x4_512 d16_512; do echo "=== $t: tools/bin/mb $(grep -o 'N=.*' gpurun_out/prof_$t/mb.txt | cut -c1-140)"; grep -v "^$" gpurun_out/prof_$t/kernel_stats.txt | head -8 | cut -c1-150; done > $O/rule_shapes_kernels.txt
cat $O/rule_shapes_kernels.txt
