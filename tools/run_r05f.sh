cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
timeout 600 python tools/fuzz_lists.py 40 17 > $O/fuzz_lists_17.txt 2>&1; tail -5 $O/fuzz_lists_17.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 > $O/pytest_full.txt; tail -15 $O/pytest_full.txt
