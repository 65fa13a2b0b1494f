#!/bin/bash
# usage: tools/pmc_rs.sh <outdir>   -- counter-only rocprofv3 passes over tools/rs_probe.py (register-stationary vs streaming 32->32 conv)
OUT=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $SET --output-format csv -d $R/$OUT/p$i -- python3 $R/tools/rs_probe.py --iters 3 > $R/$OUT/p$i.log 2>&1 || exit 1
done
