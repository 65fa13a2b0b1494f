#!/bin/bash
# Same-box A/B of the working tree against a copy of another commit's tree built under .ab_head/ (git archive <commit> | tar -x -C
# .ab_head && (cd .ab_head && python -c "import __graft_entry__ as g; g.build()") in the build container; .ab_head/ is git-ignored
# and travels to the GPU box).  Alternates the two benches N times: tools/ab_head.sh [rounds] [bench args...]
set -e
N=${1:-2}; shift || true
ARGS=${@:---no-extras --no-cpu-baseline --steps 40 --warmup 10}
for i in $(seq 1 $N); do
  (cd .ab_head && python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head ', d['value'], d['ms_per_step'])")
  python bench.py $ARGS 2>gpurun_out/ab_tree.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree ', d['value'], d['ms_per_step'], d.get('saturation_flags'))"
done
