#!/bin/bash
# one-frame latency (hipGraph replay) with out_costs.0 in polyphase form forced on (MVSGI_POLY_MIN_UNITS=0) against the default
R=${GRAFT_REPO_ROOT:-$(pwd)}
for V in 4800 0; do
  for B in 1 2 4; do
    MVSGI_EXPERIMENTAL=1 MVSGI_POLY_MIN_UNITS=$V timeout -k 10 120 python3 $R/bench.py --batch $B --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_units', $V, 'B', $B, d['ms_per_step'], 'ms', d['value'], 'fps')" || exit 1
  done
done
