#!/bin/bash
# same-box A/B of diagnostic library builds through bench.py: tools/ab_libs.sh <tag> <tag> ...  (mvs_gi_amd/libmvsgi_hip_<tag>.so), two rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/ab_libs
for i in 1 2; do
  for V in "$@"; do
    MVSGI_LIB=$R/mvs_gi_amd/libmvsgi_hip_$V.so timeout -k 10 200 python3 $R/bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $R/gpurun_out/ab_libs/${V}_$i.json 2> $R/gpurun_out/ab_libs/${V}_$i.err || exit 1
  done
done
