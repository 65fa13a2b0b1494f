#!/bin/bash
# cycles per step section of the Winograd-form kernel (workgroup 0, summed over a launch of 64 frames of [8, 40, 160]; 200 plane steps + 25 unit closings per wave)
# needs: python -c "import __graft_entry__ as g; g.build_variant(['-DMVSGI_WINO_STAMPS'], 'winostamps')"
R=$(cd "$(dirname "$0")/.." && pwd)
for res in 1 0; do
  echo "== residual $res"
  MVSGI_LIB=$R/mvs_gi_amd/libmvsgi_hip_winostamps.so MVSGI_WINO_STAMP=1 python3 $R/tools/wino_probe.py --shape 64 8 40 160 --iters 1 --res $res 2>&1 | grep -E "^wave|^direct" | tail -6
done
