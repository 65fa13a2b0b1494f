#!/bin/bash
# same-box A/B of one environment knob through bench.py: tools/ab_env.sh NAME value value ...  (two rounds)
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; shift
mkdir -p $R/gpurun_out/ab_env
for i in 1 2; do
  for V in "$@"; do
    env $N=$V timeout -k 10 200 python3 $R/bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $R/gpurun_out/ab_env/${N}_${V}_$i.json 2> $R/gpurun_out/ab_env/${N}_${V}_$i.err || exit 1
  done
done
