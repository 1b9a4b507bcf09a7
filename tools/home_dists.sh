cd $GRAFT_REPO_ROOT; export GSASR_SPLAT_DEV=1; O=gpurun_out/home_dists; mkdir -p $O
for dist in 0 1 3 4 5 6; do
  for cfg in "C2x16 0 256 256 4 0.1 0 10 16 6" "C5 0 192 192 4 0.5 0 10 16 6" "x4g4 1 256 256 4 0.1 0 10 4 6"; do
    set -- $cfg; name=$1; v=$2; shift 2
    echo -n "dist$dist $name v$v: "; MB_DIST=$dist GSASR_SPLAT_HOME_VARIANT=$v MB_ALT_FLAGS=32768 timeout 120 tools/bin/mb "$@" 2>&1 | grep "alt flags\|g_coords" | tr '\n' ' ' | cut -c1-220; echo
    echo -n "dist$dist $name tile: "; MB_DIST=$dist GSASR_SPLAT_BWD=tile GSASR_SPLAT_LISTS=1 timeout 120 tools/bin/mb "$@" 2>&1 | grep -o "plan.*bwd [0-9.]* us"
  done
done > $O/dists.txt 2>&1
cat $O/dists.txt
