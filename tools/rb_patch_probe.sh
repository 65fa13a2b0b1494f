#!/bin/bash
# brick-order experiment of the split residual-block kernel: time and HBM fetch per patch size (MVSGI_RB_PATCH; 1000 = row-major)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/rb_patch
cd /tmp && export TMPDIR=/tmp
for P in 1000 8 4 6 12; do
  export MVSGI_RB_PATCH=$P
  timeout -k 5 120 python3 $R/tools/extractor_probe.py 32 5 > $R/gpurun_out/rb_patch/time_$P.log 2>&1 || exit 1
  timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/rb_patch/p$P -- python3 $R/tools/extractor_probe.py 32 2 > $R/gpurun_out/rb_patch/pmc_$P.log 2>&1 || exit 1
done
