#!/bin/bash
# The stride-2 first layers of the regulators on the working tree (2 x 2 x 16 bricks where the output rows are whole tiles) against a
# copy of another commit's tree under .ab_head/ (2 x 4 x 8 bricks): tools/s2_brick_probe.sh  (see tools/ab_head.sh for .ab_head/)
for shape in "64 48 96 8 80 320 2" "16 64 32 32 80 320 2" "32 16 96 16 80 320 2" "64 32 64 8 40 160 2" "64 96 192 4 40 160 2" "32 96 192 8 40 160 2"; do
  echo "== B Cin Cout D H W stride = $shape (fp16 split)"
  (cd .ab_head && python tools/conv_probe.py --shape $shape --f16 --iters 20 2>/dev/null | sed 's/^/   head: /')
  python tools/conv_probe.py --shape $shape --f16 --iters 20 2>/dev/null | sed 's/^/   tree: /'
done
