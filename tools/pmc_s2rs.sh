#!/bin/bash
# HBM traffic of the stride-2 kernels: counter-only rocprofv3 passes over tools/s2rs_probe.py (stops at the first failure)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_s2rs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 5 120 rocprofv3 --pmc $SET --output-format csv -d $OUT/$SET -- python3 $R/tools/s2rs_probe.py 64 > $OUT/$SET.log 2>&1 || exit 1
done
