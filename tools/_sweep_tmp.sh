cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/tlsep; export GSASR_SPLAT_DEV=1
for sh in 0 1; do for sep in 0 1; do
  for cfg in "C4 1024 1024 8 0.1 0 8 1 6" "x8_2048 256 256 8 0.1 0 12 1 6" "C2x16 256 256 4 0.1 0 12 16 6" "C5 192 192 4 0.5 0 12 16 6" "x2 512 512 2 0.1 0 12 1 6" "x4g4 256 256 4 0.1 0 12 4 6" "C2lists 256 256 4 0.1 0 12 1 6"; do set -- $cfg; n=$1; shift
    if [ $n = C2lists ]; then export MB_LIST_CAP=256; else unset MB_LIST_CAP; fi
    echo -n "shuffle$sh sep$sep $n: "; GSASR_SPLAT_TL_SEPARATE=$sep MB_SHUFFLE=$sh timeout 200 tools/bin/mb "$@" 2>&1 | grep -o "plan.*bwd [0-9.]* us.*sum(img)=[0-9.e+]*" | head -1
  done; done; done > gpurun_out/tlsep/ab.txt 2>&1
cat gpurun_out/tlsep/ab.txt
