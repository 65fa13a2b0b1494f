#!/bin/bash
# per-layer time of the streaming conv under the compile-time ablations (bit 1 no weight loads, 2 no LDS fragment reads,
# 8 no producer staging writes, 16 no MFMAs): tools/abl_layers.sh  (needs the libmvsgi_hip_stamps<abl>.so builds)
for shape in "64 128 128 2 10 40 1" "64 64 64 4 20 80 1"; do
  echo "shape $shape"; python tools/stamp_probe.py $shape 2>/dev/null | grep "^us"
  for a in 1 2 8 16; do
    echo -n " abl=$a "; MVSGI_LIB=$PWD/mvs_gi_amd/libmvsgi_hip_stamps$a.so python tools/stamp_probe.py $shape 2>/dev/null | grep "^us"
  done
done
