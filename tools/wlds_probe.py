#!/usr/bin/env python3
"""Small-launch streaming conv variants against each other (diagnostic library built with -DMVSGI_EXPERIMENTAL, MVSGI_B3_FORCE read per call):
results must agree to the summation order, times per launch.   MVSGI_LIB=.../libmvsgi_hip_exp.so python tools/wlds_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H  # noqa: E402

dev = "cuda:0"
rng = np.random.default_rng(0)
variants = sys.argv[1].split() if len(sys.argv) > 1 else ["", "N16_T", "N16_TW", "N32_T", "N64_S"]
SHAPES = [(1, 64, 64, 4, 20, 80), (1, 128, 128, 2, 10, 40), (2, 64, 64, 4, 20, 80), (2, 128, 128, 2, 10, 40), (4, 64, 64, 4, 20, 80),
          (4, 128, 128, 2, 10, 40), (8, 64, 64, 4, 20, 80), (8, 128, 128, 2, 10, 40), (1, 32, 48, 3, 7, 19)]
if len(sys.argv) > 2:       # "B cin cout d h w; B cin cout d h w; ..."
    SHAPES = [tuple(int(v) for v in t.split()) for t in sys.argv[2].split(";") if t.strip()]
for shape in SHAPES:
    B, cin, cout, d, h, w = shape[:6]
    stride = shape[6] if len(shape) > 6 else 1          # (d, h, w: the INPUT sizes)
    x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(dev)
    wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev)
    wp = H.pack_conv_weights_bf16x3(wt)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev)
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(dev)
    res = torch.from_numpy(rng.standard_normal((B, (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1, cout), dtype=np.float32)).to(dev)
    ref, line = None, []
    for v in variants:
        if v:
            os.environ["MVSGI_B3_FORCE"] = v
        else:
            os.environ.pop("MVSGI_B3_FORCE", None)
        y = H.conv3d(x, wt, wp, sc, sh, res=res, stride=stride, impl=H.CONV_BF16X3)
        for _ in range(3):
            H.conv3d(x, wt, wp, sc, sh, res=res, stride=stride, impl=H.CONV_BF16X3, out=y)
        torch.cuda.synchronize()
        # 20 launches as one hipGraph replay: device time per launch (a Python / ctypes launch costs more than these kernels run)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            H.conv3d(x, wt, wp, sc, sh, res=res, stride=stride, impl=H.CONV_BF16X3, out=y)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            for _ in range(20):
                H.conv3d(x, wt, wp, sc, sh, res=res, stride=stride, impl=H.CONV_BF16X3, out=y)
        g.replay()
        torch.cuda.synchronize()
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(10):
            g.replay()
        e0.record()
        torch.cuda.synchronize()
        us = s0.elapsed_time(e0) / 200 * 1e3
        del g
        if ref is None:
            ref = y.clone()
        err = float((y - ref).abs().max() / ref.abs().max())
        line.append(f"{v or 'default'} {us:.1f} us (diff {err:.1e})")
    os.environ.pop("MVSGI_B3_FORCE", None)
    print(shape, " | ".join(line), flush=True)
