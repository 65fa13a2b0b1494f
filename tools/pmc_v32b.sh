#!/bin/bash
# usage: tools/pmc_v32b.sh <outdir>   -- does halving the weight-fragment instructions move the streaming kernel?  Counters of one
# layer (192 -> 192 [4,20,80] x 32 frames, then 64 -> 64 [32,80,320] x 16) in three schedules: the dispatcher's 16x16x32 units, the
# 32x32x16 schedule with 64-voxel waves, and with 128-voxel waves (B3V_N64B, -DMVSGI_EXPERIMENTAL build: half the weight fragments
# per MFMA).  One counter group per rocprofv3 --pmc pass; summary by tools/summarize_sq.py.
OUT=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
export MVSGI_LIB=$R/mvs_gi_amd/libmvsgi_hip_exp.so MVSGI_EXPERIMENTAL=1
for SHP in "32 192 192 4 20 80 1" "16 64 64 32 80 320 1"; do
  for V in default v32 v32b; do
    ARGS="--shape $SHP --iters 10"
    unset MVSGI_V32B
    [ $V = v32 ] && ARGS="$ARGS --v32"
    [ $V = v32b ] && ARGS="$ARGS --v32" && export MVSGI_V32B=1
    DIRS=""
    i=0
    for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
               "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
               "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
      i=$((i+1))
      timeout -k 10 150 rocprofv3 --pmc $SET --output-format csv -d $R/$OUT/p$i -- python3 $R/tools/conv_probe.py $ARGS > $R/$OUT/p$i.log 2>&1 || { echo "pass $i failed ($SET)" >> $R/$OUT/summary.txt; continue; }
      DIRS="$DIRS $R/$OUT/p$i"
    done
    python3 $R/tools/summarize_sq.py $R/$OUT/summary.txt "$SHP  schedule: $V" "conv3d_bf16x3" $DIRS > /dev/null
    grep "us per launch" $R/$OUT/p1.log >> $R/$OUT/summary.txt
    for d in $DIRS; do rm -rf $d; done
  done
done
