#!/bin/bash
# same-box A/B of the split-padded builder -> regulator hand-over (MVSGI_S2RS): two rounds each
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/ab_s2rs
for i in 1 2; do
  for V in 0 1; do
    MVSGI_S2RS=$V timeout -k 10 200 python3 $R/bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $R/gpurun_out/ab_s2rs/v${V}_$i.json 2> $R/gpurun_out/ab_s2rs/v${V}_$i.err || exit 1
  done
done
