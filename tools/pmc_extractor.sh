#!/bin/bash
# usage: tools/pmc_extractor.sh <outdir> [frames]   -- counter-only rocprofv3 passes over tools/extractor_probe.py (stops at the first failure)
OUT=$1; FR=${2:-32}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $SET --output-format csv -d $R/$OUT/p$i -- python3 $R/tools/extractor_probe.py $FR 2 > $R/$OUT/p$i.log 2>&1 || exit 1
done
