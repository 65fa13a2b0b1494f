#!/usr/bin/env python3
"""Randomised cross-check of the split-bf16 conv dispatcher against the exact-fp32 MFMA kernel on the GPU (both through the C ABI):
shapes drawn around the dispatcher's thresholds (one-plane volumes, 5-row planes, 96 / 128 / 192-cout units, stride 2, one frame to
many).  python tools/conv_fuzz.py [n] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H  # noqa: E402

DEV = "cuda:0"


def run(n=60, seed=0, verbose=True):
    """-> (worst max-abs difference relative to the tensor's max, {variant name: count})"""
    rng = np.random.default_rng(seed)
    worst, seen = 0.0, {}
    for it in range(n):
        cin = int(rng.choice([16, 32, 48, 64, 96, 128, 192]))
        cout = int(rng.choice([16, 32, 48, 64, 96, 128, 192, 384]))
        stride = int(rng.choice([1, 1, 2]))
        d = int(rng.choice([1, 1, 2, 3, 4, 5, 8])) * (2 if stride == 2 and rng.random() < 0.5 else 1)
        h = int(rng.choice([4, 5, 7, 8, 10, 12, 15, 20]))
        w = int(rng.choice([8, 16, 19, 24, 40, 48]))
        B = int(rng.choice([1, 1, 2, 3, 4, 8, 16, 33]))
        while B * d * h * w * max(cin, cout) > 6e7:
            B = max(1, B // 2)
        res = bool(rng.random() < 0.5)
        slope = float(rng.choice([0.01, 0.0, 1.0]))
        x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(DEV)
        wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(DEV)
        sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(DEV)
        sh = torch.from_numpy((rng.standard_normal(cout) * 0.2).astype(np.float32)).to(DEV)
        do, ho, wo = (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1
        r = torch.from_numpy(rng.standard_normal((B, do, ho, wo, cout), dtype=np.float32)).to(DEV) if res else None
        ref = H.conv3d(x, wt, H.pack_conv_weights(wt), sc, sh, res=r, stride=stride, neg_slope=slope, impl=H.CONV_MFMA)
        y = H.conv3d(x, wt, H.pack_conv_weights_bf16x3(wt), sc, sh, res=r, stride=stride, neg_slope=slope, impl=H.CONV_BF16X3)
        torch.cuda.synchronize()
        assert torch.isfinite(y).all()
        err = float((y - ref).abs().max() / ref.abs().max())
        v = H.conv3d_variant(B, cin, d, h, w, cout, stride, H.CONV_BF16X3)
        seen[v] = seen.get(v, 0) + 1
        worst = max(worst, err)
        if verbose:
            print(f"{it:3d} B{B} {cin}->{cout} [{d},{h},{w}] s{stride} res={int(res)} slope={slope}: {err:.2e}  {v[20:70]}"
                  f"{'' if err <= 1e-4 else '   <-- FAIL'}", flush=True)
    return worst, seen


if __name__ == "__main__":
    worst_, seen_ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("worst", worst_)
    for k, c in sorted(seen_.items(), key=lambda kv: -kv[1]):
        print(c, k)
    assert worst_ <= 1e-4
