#!/bin/bash
# ablation builds of the streaming kernel (-DMVSGI_ABL=n: 1 no weight loads, 2 no LDS fragment reads, 8 no staging; sums combine) on one
# layer shape; results are wrong by construction, the timing bounds what each stream costs.
# usage: tools/abl_probe.sh "<B Cin Cout D H W stride>" <n> <n> ...   (libraries: __graft_entry__.build_variant(['-DMVSGI_ABL=n'], 'abln'))
R=${GRAFT_REPO_ROOT:-$(pwd)}
SHAPE=$1; shift
echo "== $SHAPE product"; timeout -k 10 120 python3 $R/tools/conv_probe.py --shape $SHAPE --iters 10 || exit 1
for N in "$@"; do
  echo "== $SHAPE ABL=$N"
  MVSGI_LIB=$R/mvs_gi_amd/libmvsgi_hip_abl$N.so timeout -k 10 120 python3 $R/tools/conv_probe.py --shape $SHAPE --iters 10 || exit 1
done
