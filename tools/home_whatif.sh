cd $GRAFT_REPO_ROOT; export GSASR_SPLAT_DEV=1; O=gpurun_out/home_whatif; mkdir -p $O
for b in mb mb_nolds mb_noeval mb_noreduce; do
  for cfg in "C2x16 0 256 256 4 0.1 0 20 16 6" "C5 0 192 192 4 0.5 0 20 16 6" "C2 1 256 256 4 0.1 0 30 1 6" "C2 2 256 256 4 0.1 0 30 1 6"; do
    set -- $cfg; name=$1; v=$2; shift 2
    echo -n "$b $name v$v: "; GSASR_SPLAT_HOME_VARIANT=$v MB_ALT_FLAGS=32768 timeout 120 tools/bin/$b "$@" 2>&1 | grep "alt flags\|g_sigmas" | tr '\n' ' ' | cut -c1-200; echo
  done
done > $O/whatif.txt 2>&1
cat $O/whatif.txt
