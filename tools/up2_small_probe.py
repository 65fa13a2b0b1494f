#!/usr/bin/env python3
"""Small-launch variants of the fused upsample + conv against each other (diagnostic library built with -DMVSGI_EXPERIMENTAL,
MVSGI_B3U_FORCE read per call): results must agree to the summation order, device time per launch from a hipGraph of 20.
   MVSGI_LIB=.../libmvsgi_hip_exp.so python tools/up2_small_probe.py ["<variant> <variant> ..."]     ("" = the dispatcher's choice)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H  # noqa: E402

dev = "cuda:0"
rng = np.random.default_rng(0)
variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["", "N64", "N32_TB", "N32_M", "N32"]
# (frames, Cin, Cout, low-resolution D, H, W): the up blocks of the (16, 32) regulator at 16 and 8 candidates, and a ragged one
for shape in [(1, 128, 64, 2, 10, 40), (2, 128, 64, 2, 10, 40), (4, 128, 64, 2, 10, 40), (8, 128, 64, 2, 10, 40),
              (1, 64, 32, 4, 20, 80), (2, 64, 32, 4, 20, 80), (4, 64, 32, 4, 20, 80),
              (1, 384, 192, 1, 5, 20), (1, 192, 96, 2, 10, 40), (1, 96, 48, 4, 20, 80), (1, 32, 64, 3, 5, 9)]:
    B, cin, cout, d, h, w = shape
    x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(dev)
    wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev)
    wp = H.pack_conv_weights_bf16x3(wt)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev)
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(dev)
    res = torch.from_numpy(rng.standard_normal((B, 2 * d, 2 * h, 2 * w, cout), dtype=np.float32)).to(dev)
    ref, line = None, []
    for v in variants:
        if v:
            os.environ["MVSGI_B3U_FORCE"] = v
        else:
            os.environ.pop("MVSGI_B3U_FORCE", None)
        if v in ("N32_M", "N32") and cout != 32 or v == "N96" and cout % 96:
            continue
        y = H.conv3d_up2(x, wp, sc, sh, res=res)
        for _ in range(3):
            H.conv3d_up2(x, wp, sc, sh, res=res, out=y)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            H.conv3d_up2(x, wp, sc, sh, res=res, out=y)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            for _ in range(20):
                H.conv3d_up2(x, wp, sc, sh, res=res, out=y)
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
        line.append(f"{v or 'default'} {H.conv3d_up2_variant(B, cin, d, h, w, cout)[21:40] if not v else ''} {us:.1f} us (diff {err:.1e})")
    os.environ.pop("MVSGI_B3U_FORCE", None)
    print(shape, " | ".join(line), flush=True)
