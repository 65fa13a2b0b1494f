#!/bin/bash
# A/B of streaming-kernel brick variants on one layer shape (diagnostic library built with -DMVSGI_EXPERIMENTAL:
#   python -c "import __graft_entry__ as g; g.build_variant(['-DMVSGI_EXPERIMENTAL'], 'exp')")
# usage: tools/b3_variants_probe.sh "<B Cin Cout D H W stride>" <variant> <variant> ...      (variants: N64 N64_H5 N96 N96_H5 N128_P N128_PH5 N192_PH5 N64_S)
R=${GRAFT_REPO_ROOT:-$(pwd)}
SHAPE=$1; shift
export MVSGI_LIB=$R/mvs_gi_amd/libmvsgi_hip_exp.so
for V in "$@"; do
  echo "== $SHAPE forced $V"
  MVSGI_B3_FORCE=$V timeout -k 10 120 python3 $R/tools/conv_probe.py --shape $SHAPE --iters 10 || exit 1
done
