cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
timeout 300 python tools/fuzz_cross.py 30 9 2>&1 | tail -1 > $O/sanity.txt
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_adaptive_cutoff.py -m gpu -x -q 2>&1 | tail -2 >> $O/sanity.txt
cat $O/sanity.txt
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1
tail -60 $O/collect.log
