#!/bin/bash
# LDS bank conflicts and issue shares of a stride-2 layer of the streaming kernel with 2 x 4 x 8 bricks (the copy of another commit's
# tree under .ab_head/, see tools/ab_head.sh) and with 2 x 2 x 16 bricks (the working tree): counter-only rocprofv3 passes.
# usage: tools/pmc_s2_bricks.sh <outdir under the repo> "<B Cin Cout D H W stride>"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; SHAPE=${2:-"64 48 96 8 80 320 2"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for TREE in head tree; do
  T=$R; [ $TREE = head ] && T=$R/.ab_head
  i=0
  for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"; do
    i=$((i+1))
    (cd $T && timeout -k 5 120 rocprofv3 --pmc $SET --output-format csv -d $OUT/$TREE/p$i -- python3 $T/tools/conv_probe.py --shape $SHAPE --f16 --iters 3 > $OUT/$TREE.p$i.log 2>&1) || exit 1
  done
  python3 $R/tools/summarize_sq.py $OUT/summary.txt "$TREE: B Cin Cout D H W stride = $SHAPE" "conv3d_f16x3_kernel" $OUT/$TREE/p1 $OUT/$TREE/p2 > /dev/null
done
cat $OUT/summary.txt
