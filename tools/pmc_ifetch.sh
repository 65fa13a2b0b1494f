#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/pmc_ifetch
cd /tmp && export TMPDIR=/tmp
for SHAPE in "1 64 64 4 20 80 1" "64 64 64 4 20 80 1"; do
  T=$(echo $SHAPE | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_ifetch/$T -- python3 $R/tools/conv_probe.py --shape $SHAPE --iters 5 > $R/gpurun_out/pmc_ifetch/$T.log 2>&1 || exit 1
  python3 $R/tools/summarize_sq.py $R/gpurun_out/pmc_ifetch/summary.txt "$SHAPE" "conv3d_bf16x3" $R/gpurun_out/pmc_ifetch/$T > /dev/null
  rm -rf $R/gpurun_out/pmc_ifetch/$T
done
