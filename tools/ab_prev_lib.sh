#!/bin/bash
# bench.py on the working tree's library against a previous build kept as mvs_gi_amd/libmvsgi_hip_prev.so, alternating on one box:
#   tools/ab_prev_lib.sh "<config> ..." [rounds] [extra bench flags]
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/mvs_gi_amd/libmvsgi_hip_prev.so
[ -f $P ] || { echo "no $P"; exit 1; }
one() { python3 $R/bench.py --config $1 --steps 10 --warmup 3 --no-extras --no-cpu-baseline $3 2>/dev/null | tail -1 | python3 -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], '$2')
except Exception as e: print('$1 failed', e)"; }
for T in $1; do
  for i in $(seq 1 ${2:-2}); do
    one $T "" "$3" || exit 1
    MVSGI_LIB=$P one $T "(previous library)" "$3" || exit 1
  done
done
