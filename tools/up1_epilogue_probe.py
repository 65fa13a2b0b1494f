#!/usr/bin/env python3
"""What the residual read and the split-padded output cost a fused-upsample layer (G16V up1: 64 -> 32 on [4,20,80] -> [8,40,160], 64 frames)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H
B, cin, cout, d, h, w = 64, 64, 32, 4, 20, 80
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev)
wp = H.pack_conv_weights_bf16x3(wt)
sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
res = torch.from_numpy(rng.standard_normal((B, 2 * d, 2 * h, 2 * w, cout), dtype=np.float32)).to(dev)
y = torch.empty_like(res)
ys = H.SplitAct(B, 2 * d, 2 * h, 2 * w, cout, dev)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(n):
        fn()
    e0.record(); torch.cuda.synchronize()
    return s0.elapsed_time(e0) / n * 1e3


for rnd in range(2):
    print("fp32 out, no residual   ", round(t(lambda: H.conv3d_up2(x, wp, sc, sh, out=y)), 1), "us")
    print("fp32 out, residual      ", round(t(lambda: H.conv3d_up2(x, wp, sc, sh, res=res, out=y)), 1), "us")
    print("split out, no residual  ", round(t(lambda: H.conv3d_up2_out_split(x, wp, sc, sh, out=ys)), 1), "us")
    print("split out, residual     ", round(t(lambda: H.conv3d_up2_out_split(x, wp, sc, sh, out=ys, res=res)), 1), "us", flush=True)
