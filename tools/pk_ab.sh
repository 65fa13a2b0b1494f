#!/bin/bash
# A/B of scalar fp32 VALU in the producers (default) against the packed forms (-DMVSGI_PK build: __graft_entry__.build_variant(["-DMVSGI_PK"], "pk")) beside the consumers' MFMAs:
# out_costs.0 (fused upsample: blend + split), a 64 -> 64 layer and the stride-2 16 -> 32 layer (split only); us per launch
for arm in default pk; do
  [ $arm = pk ] && export MVSGI_LIB=$PWD/mvs_gi_amd/libmvsgi_hip_pk.so
  for r in 1 2; do
    echo -n "$arm up2 32->16 (out_costs.0, B=32): "; python tools/stamp_probe.py 32 32 16 8 40 160 1 up2 c16 2>/dev/null | grep "^us" | cut -c1-28
    echo -n "$arm 64->64 (B=32): "; python tools/stamp_probe.py 32 64 64 4 20 80 1 2>/dev/null | grep "^us" | cut -c1-28
    echo -n "$arm 16->32 stride 2 (B=32): "; python tools/stamp_probe.py 32 16 32 16 80 320 2 2>/dev/null | grep "^us" | cut -c1-28
  done
done
