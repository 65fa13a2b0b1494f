#!/usr/bin/env python3
"""out_costs.0 of the (16, 32) regulator (ResizeConv3d 32 -> 16 on [8, 40, 160] low-resolution volumes, fp16 split, split-padded
output): the polyphase layer with its main kernel in Winograd form (csrc/conv3d_wino_up2.hip) against the direct register-stationary
main kernel (csrc/conv3d_rs.hip MODE 3) -- time per launch chain (edge + face + main kernels) over batch sizes, and their agreement.
`python tools/poly_wino_probe.py [B ...]`"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvs_gi_amd import hip_ops as H  # noqa: E402

DEV = "cuda:0"
d, h, w = 8, 40, 160
rng = np.random.default_rng(0)
wt = torch.from_numpy((rng.standard_normal((16, 32, 3, 3, 3)) / np.sqrt(27 * 32)).astype(np.float32)).to(DEV)
sc = torch.from_numpy(rng.uniform(0.5, 1.5, 16).astype(np.float32)).to(DEV)
sh = torch.from_numpy((rng.standard_normal(16) * 0.1).astype(np.float32)).to(DEV)
plan, un = H.conv3d_up2_poly_plan(wt, d, h, w, fmt="f16")
scu = sc * un


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'frames':>6s} {'direct us':>10s} {'winograd us':>12s} {'ratio':>6s} {'pays':>5s} {'max |d| / max':>14s}")
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 16, 32, 64]:
    x = torch.from_numpy(rng.standard_normal((B, d, h, w, 32), dtype=np.float32)).to(DEV)
    xs = H.act_to_split(x, fmt="f16")
    o1, o2 = H.SplitAct(B, 2 * d, 2 * h, 2 * w, 16, DEV), H.SplitAct(B, 2 * d, 2 * h, 2 * w, 16, DEV)
    td = timed(lambda: H.conv3d_up2_poly_split(xs, plan, scu, sh, out=o1, neg_slope=0.01, direct=True))
    tw = timed(lambda: H.conv3d_up2_poly_split(xs, plan, scu, sh, out=o2, neg_slope=0.01, wino=True))
    a, b = H.act_from_split(o1), H.act_from_split(o2)
    err = float((a - b).abs().max() / a.abs().max())
    print(f"{B:6d} {td:10.1f} {tw:12.1f} {td / tw:6.2f} {int(H.conv3d_up2_poly_wino_pays(B, d, h, w)):5d} {err:14.2e}", flush=True)
    del x, xs, o1, o2, a, b
    torch.cuda.empty_cache()
assert H.saturation_flags(clear=True) == 0
