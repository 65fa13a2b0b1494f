#!/usr/bin/env python3
"""post_vol (conv3d_rs16_kernel, 16 -> 16 on a split-padded volume) as the G16V step launches it: B frames of [16, 80, 320], fp16 and
bf16 split, split-padded and fp32 output -- time per launch.  Run in two trees (tools/ab_head.sh: the working tree = eight waves, four
of them loaders; .ab_head = four waves) on one box: tools/rs16_loaders_probe.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
d, h, w, dev = 16, 80, 320, "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, 16), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((16, 16, 3, 3, 3)) / np.sqrt(27 * 16)).astype(np.float32)).to(dev)
sc = torch.from_numpy(rng.uniform(0.5, 1.5, 16).astype(np.float32)).to(dev)
sh = torch.from_numpy(rng.standard_normal(16).astype(np.float32) * 0.1).to(dev)


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


gf = 2 * 27 * 16 * 16 * B * d * h * w / 1e9
for fmt in ("f16", "bf16"):
    xs = H.act_to_split(x, fmt=fmt)
    p = H.pack_conv_weights_rs(wt, fmt)
    wp, s_ = (p[0], sc * p[1]) if fmt == "f16" else (p, sc)
    out = H.SplitAct(B, d, h, w, 16, dev)
    y = torch.empty((B, d, h, w, 16), device=dev)
    ts = timeit(lambda: H.conv3d_rs16(xs, wp, s_, sh, out_split=out))
    tf = timeit(lambda: H.conv3d_rs16(xs, wp, s_, sh, out=y))
    print(f"{fmt:5s} B={B}: split-padded out {ts:7.1f} us ({gf / ts * 1e3:5.1f} TF)   fp32 out {tf:7.1f} us ({gf / tf * 1e3:5.1f} TF)   checksum {float(y.double().sum()):.6e}", flush=True)
