#!/bin/bash
# usage: tools/pmc_generic.sh <outdir> -- <python script and args>   (counter-only rocprofv3 passes)
OUT=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TCC_HIT TCC_MISS TCC_REQ" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $SET --output-format csv -d $R/$OUT/p$i -- python3 "$@" > $R/$OUT/p$i.log 2>&1
done
