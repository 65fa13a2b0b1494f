#!/bin/bash
# usage: tools/profile_configs.sh <outdir> [tags...]  -- per-config evidence: the bench line (every launch attributed) and the rocprofv3
# kernel-trace stats of the other BASELINE configurations at their bench batch.  Every pass under its own timeout; stops at the first failure.
OUT=$1; shift
TAGS=${@:-"G16VV E8 4cam-32"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for T in $TAGS; do
  B=${MVSGI_CFG_BATCH:-32}
  timeout -k 10 280 python3 $R/bench.py --config $T --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/$OUT/bench_$T.json 2> $R/$OUT/bench_$T.err || exit 1
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_$T -- python3 $R/bench.py --config $T --batch $B --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $R/$OUT/stats_$T.log 2>&1 || exit 1
  python3 $R/tools/summarize_rocprof.py stats $R/$OUT/stats_$T $R/$OUT/${T}_kernel_stats.txt > /dev/null || exit 1
  rm -rf $R/$OUT/stats_$T
done
