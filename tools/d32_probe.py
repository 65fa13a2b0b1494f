#!/usr/bin/env python3
"""The split kernel on 32-channel slices (impl CONV_BF16X3_D32) against the tap-pair layout (CONV_BF16X3), both splits: results against
an fp64 convolution and per-launch time (hipGraph of 10).   python tools/d32_probe.py ["B cin cout d h w; ..."]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H  # noqa: E402

dev = "cuda:0"
rng = np.random.default_rng(0)
SHAPES = [(2, 64, 64, 3, 9, 21), (128, 64, 64, 4, 20, 80), (128, 128, 128, 2, 10, 40), (8, 96, 96, 16, 80, 320), (32, 192, 192, 8, 40, 160),
          (32, 96, 96, 4, 10, 40), (64, 384, 384, 1, 10, 40), (32, 64, 64, 3, 15, 21)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in t.split()) for t in sys.argv[1].split(";") if t.strip()]


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(5):
        g.replay()
    e0.record()
    torch.cuda.synchronize()
    return s0.elapsed_time(e0) / 50 * 1e3


for shape in SHAPES:
    B, cin, cout, d, h, w = shape
    x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(dev)
    wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev)
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(dev)
    res = torch.from_numpy(rng.standard_normal((B, d, h, w, cout), dtype=np.float32)).to(dev)
    ok = H.conv3d_d32_applies(B, cin, d, h, w, cout)
    line = [f"{shape} d32_applies {int(ok)}"]
    ref = None
    if B * d * h * w * cin * cout <= 3e10:      # fp64 reference on the GPU via torch (probe only)
        y64 = F.conv3d(x[:2].permute(0, 4, 1, 2, 3).double(), wt.double(), padding=1) * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1)
        y64 = y64 + res[:2].permute(0, 4, 1, 2, 3).double()
        ref = torch.where(y64 > 0, y64, y64 * 0.01).permute(0, 2, 3, 4, 1).float()
    for f16 in (False, True):
        outs = {}
        for name, layout in (("pair", H.CONV_BF16X3), ("d32", H.CONV_BF16X3_D32)):
            if layout == H.CONV_BF16X3_D32 and not ok:
                continue
            if f16:
                wp, un = H.pack_conv_weights_f16x3(wt, layout)
                s_, impl = sc * un, layout | H.CONV_F16
            else:
                wp = H.pack_conv_weights_bf16x3(wt) if layout == H.CONV_BF16X3 else H.pack_conv_weights_bf16x3_d32(wt)
                s_, impl = sc, layout
            y = H.conv3d(x, wt, wp, s_, sh, res=res, impl=impl)
            us = timed(lambda: H.conv3d(x, wt, wp, s_, sh, res=res, impl=impl, out=y))
            outs[name] = y
            err = float((y[:2] - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
            line.append(f"{'f16' if f16 else 'bf16'} {name} {us:.1f} us err {err:.1e} [{H.conv3d_variant(B, cin, d, h, w, cout, 1, impl)[7:48]}]")
        if len(outs) == 2:
            line.append(f"d32 vs pair {float((outs['d32'] - outs['pair']).abs().max() / outs['pair'].abs().max()):.1e}")
    print(" | ".join(line), flush=True)
