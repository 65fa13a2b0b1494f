#!/usr/bin/env python3
"""Register-stationary 16->16 conv (post_vol) against the plane-schedule streaming kernel: parity and time."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", type=int, nargs=4, default=[32, 16, 80, 320], help="B D H W")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
B, d, h, w = a.shape
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, 16), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((16, 16, 3, 3, 3)) / np.sqrt(27 * 16)).astype(np.float32)).to(dev)
wpc, wpr = H.pack_conv_weights_bf16x3_c16(wt), H.pack_conv_weights_rs(wt)
sc = torch.from_numpy(rng.uniform(0.5, 1.5, 16).astype(np.float32)).to(dev)
sh = torch.from_numpy(rng.standard_normal(16).astype(np.float32) * 0.1).to(dev)
y_ref = H.conv3d(x, wt, wpc, sc, sh, impl=H.CONV_BF16X3_C16)
xs = H.act_to_split(x)
y = H.conv3d_rs16(xs, wpr, sc, sh)
torch.cuda.synchronize()
err = float((y - y_ref).abs().max() / y_ref.abs().max())
print("rs16 vs plane-schedule kernel: max rel", err, "finite", bool(torch.isfinite(y).all()))
bad = ((y - y_ref).abs() > 1e-3 * y_ref.abs().max()).nonzero()
print("bad", bad.shape[0], bad[:6].tolist())


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


yo = torch.empty_like(y_ref)
for rep in range(3):
    t_old = timeit(lambda: H.conv3d(x, wt, wpc, sc, sh, impl=H.CONV_BF16X3_C16, out=yo), a.iters)
    t_new = timeit(lambda: H.conv3d_rs16(xs, wpr, sc, sh, out=yo), a.iters)
    gf = 2 * 27 * 16 * 16 * B * d * h * w / 1e9
    print(f"plane-schedule {t_old:8.1f} us ({gf / t_old * 1e3:6.1f} TF)   register-stationary {t_new:8.1f} us ({gf / t_new * 1e3:6.1f} TF)")
