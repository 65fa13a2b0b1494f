#!/bin/bash
# 128-voxel waves on the 32x32x16 schedule (B3V_N64B, experimental build) against the dispatcher's 16x16x32 units and the 64-voxel 32x32x16 waves
R=$(cd "$(dirname "$0")/.." && pwd)
export MVSGI_LIB=$R/mvs_gi_amd/libmvsgi_hip_exp.so MVSGI_EXPERIMENTAL=1
for shp in "64 64 64 4 20 80 1" "32 96 96 8 40 160 1" "16 64 64 32 80 320 1" "32 192 192 4 20 80 1"; do
  for r in "" "--res"; do
    echo "== $shp $r"
    python3 $R/tools/conv_probe.py --shape $shp --iters 20 $r 2>&1 | grep -v amdgpu.ids
    python3 $R/tools/conv_probe.py --shape $shp --iters 20 --v32 $r 2>&1 | grep -v amdgpu.ids
    MVSGI_V32B=1 python3 $R/tools/conv_probe.py --shape $shp --iters 20 --v32 $r 2>&1 | grep -v amdgpu.ids
  done
done
