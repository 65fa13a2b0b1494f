R=$(cd "$(dirname "$0")/.." && pwd)
for shp in "64 48 96 8 80 320 2" "16 64 32 32 80 320 2" "32 16 96 16 80 320 2"; do
  for f in "" "--f16"; do
    echo "== $shp $f"
    python3 $R/tools/conv_probe.py --shape $shp --iters 10 $f 2>&1 | grep -v amdgpu.ids
    MVSGI_LIB=$R/mvs_gi_amd/libmvsgi_hip_abl4.so python3 $R/tools/conv_probe.py --shape $shp --iters 10 $f 2>&1 | grep -v amdgpu.ids | head -1
  done
done
