#!/bin/bash
# same-box A/B of diagnostic library builds on single conv layers: tools/ab_conv_libs.sh "<tag> <tag> ..." "<B Cin Cout D H W stride>" ...   ('product' = the tree's library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAGS=$1; shift
for SHAPE in "$@"; do
  for i in 1 2; do
    for T in $TAGS; do
      if [ $T = product ]; then L=$R/mvs_gi_amd/libmvsgi_hip.so; else L=$R/mvs_gi_amd/libmvsgi_hip_$T.so; fi
      echo -n "$SHAPE [$T] "; MVSGI_LIB=$L timeout -k 10 120 python3 $R/tools/conv_probe.py --shape $SHAPE --iters 10 2>/dev/null | head -1 || exit 1
    done
  done
done
