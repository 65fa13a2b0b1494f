#!/bin/bash
# per-kernel times of the one-frame step (B = 1): rocprofv3 --kernel-trace --stats over bench.py --batch 1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_b1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --batch 1 --steps 50 --warmup 10 --no-extras --no-cpu-baseline --eager > $OUT/stats.log 2>&1 || exit 1
