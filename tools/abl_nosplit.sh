for shape in "64 16 32 16 80 320 2" "64 64 64 4 20 80 1" "64 128 128 2 10 40 1" "64 32 64 8 40 160 2"; do
  echo "shape $shape"
  echo -n "  product : "; python tools/conv_probe.py --shape $shape --iters 10 | head -1
  echo -n "  no split: "; MVSGI_LIB=$PWD/mvs_gi_amd/libmvsgi_hip_abl4.so python tools/conv_probe.py --shape $shape --iters 10 | head -1
done
