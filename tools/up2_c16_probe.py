#!/usr/bin/env python3
"""Cout == 16 fused-upsample conv (out_costs.0 of the wide regulators): plane schedule against the general 16-cout schedule.
python tools/up2_c16_probe.py B Cin Dl Hl Wl"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H
B, cin, d, h, w = [int(v) for v in sys.argv[1:6]]
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((16, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev)
sc, sh = torch.ones(16, device=dev), torch.zeros(16, device=dev)
for name, wp, lay in (("plane", H.pack_conv_weights_bf16x3_c16(wt), H.CONV_BF16X3_C16), ("general", H.pack_conv_weights_bf16x3(wt), H.CONV_BF16X3)):
    y = H.conv3d_up2(x, wp, sc, sh, w_layout=lay)
    torch.cuda.synchronize()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(10):
        H.conv3d_up2(x, wp, sc, sh, w_layout=lay, out=y)
    e0.record()
    torch.cuda.synchronize()
    us = s0.elapsed_time(e0) / 10 * 1e3
    print(f"{name}: {us:.1f} us  {2 * 27 * cin * 16 * B * 8 * d * h * w / us / 1e6:.1f} TFLOP/s  {H.conv3d_up2_variant(B, cin, d, h, w, 16, lay)}  mean|y| {float(y.abs().mean()):.5f}")
