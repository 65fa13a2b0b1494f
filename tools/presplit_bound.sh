R=$PWD
for shp in "128 64 64 4 20 80 1" "128 128 128 2 10 40 1" "8 96 96 16 80 320 1" "32 192 192 8 40 160 1" "32 96 96 8 80 320 1" "64 48 48 8 80 320 1"; do
  for i in 1 2; do
  echo "== $shp (f16 split) product / no-split"
  python3 $R/tools/conv_probe.py --shape $shp --iters 10 --f16 --res 2>&1 | grep "us per"
  MVSGI_LIB=$R/mvs_gi_amd/libmvsgi_hip_abl4.so python3 $R/tools/conv_probe.py --shape $shp --iters 10 --f16 --res 2>&1 | grep "us per"
  done
done
