#!/usr/bin/env python3
"""Fused-upsample conv unit shapes against each other (-DMVSGI_EXPERIMENTAL build, MVSGI_B3U_FORCE read per call).
python tools/up2_variants_probe.py B Cin Cout Dl Hl Wl "<variant> ..."     variants: N32 N32_M N48 N64 N96 ('' = the dispatcher's choice)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H
B, cin, cout, d, h, w = [int(v) for v in sys.argv[1:7]]
variants = sys.argv[7].split() if len(sys.argv) > 7 else [""]
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev)
wp = H.pack_conv_weights_bf16x3(wt)
sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
ref = None
for rnd in range(2):
    for v in variants + ["default"]:
        if v != "default":
            os.environ["MVSGI_B3U_FORCE"] = v
        else:
            os.environ.pop("MVSGI_B3U_FORCE", None)
        y = H.conv3d_up2(x, wp, sc, sh)
        torch.cuda.synchronize()
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(10):
            H.conv3d_up2(x, wp, sc, sh, out=y)
        e0.record()
        torch.cuda.synchronize()
        us = s0.elapsed_time(e0) / 10 * 1e3
        if ref is None:
            ref = y.clone()
        print(f"{v}: {us:.1f} us  {2 * 27 * cin * cout * B * 8 * d * h * w / us / 1e6:.1f} TFLOP/s  diff {float((y - ref).abs().max() / ref.abs().max()):.1e}", flush=True)
