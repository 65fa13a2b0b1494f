#!/bin/bash
# usage: tools/pmc_mem.sh <outdir> <title> <kernel substring[;...]> -- <python script, relative to the repo root> [args]
# Counter-only rocprofv3 passes around one command for the memory path of a kernel: SQ wave states, vector-memory issue, texture
# addresser (TA), vector L1 (TCP), L2 (TCC).  One counter group per pass, every pass under its own timeout; summary appended to
# <outdir>/summary.txt by tools/summarize_sq.py.
OUT=$1; TITLE=$2; SUBS=$3; shift; shift; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
DIRS=""
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  if [ -n "$PMC_ONLY" ] && ! echo "$SET" | grep -q "$PMC_ONLY"; then continue; fi      # PMC_ONLY=<substring>: only the groups that name it (the texture addresser takes two counters per pass)
  timeout -k 10 150 rocprofv3 --pmc $SET --output-format csv -d $R/$OUT/p$i -- python3 $R/"$@" > $R/$OUT/p$i.log 2>&1 || { echo "pass $i failed ($SET)" >> $R/$OUT/summary.txt; continue; }
  DIRS="$DIRS $R/$OUT/p$i"
done
python3 $R/tools/summarize_sq.py $R/$OUT/summary.txt "$TITLE" "$SUBS" $DIRS > /dev/null
for d in $DIRS; do rm -rf $d; done
