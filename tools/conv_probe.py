#!/usr/bin/env python3
"""Run one conv3d problem repeatedly (for rocprofv3 --pmc / --kernel-trace on a single kernel)."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", type=int, nargs=7, default=[8, 32, 32, 8, 40, 160, 1], help="B Cin Cout D H W stride")
ap.add_argument("--mode", default="bf16x3")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--res", action="store_true", help="with a residual input")
ap.add_argument("--v32", action="store_true", help="the 32x32x16-MFMA schedule (MVSGI_CONV_BF16X3_V32)")
ap.add_argument("--f16", action="store_true", help="the fp16 split")
a = ap.parse_args()
B, cin, cout, d, h, w, s = a.shape
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev)
sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
if a.mode == "bf16x3":
    impl = H.CONV_BF16X3_V32 if a.v32 else H.CONV_BF16X3
    if a.f16:
        wp, un = H.pack_conv_weights_f16x3(wt, impl)
        sc, impl = sc * un, impl | H.CONV_F16
    else:
        wp = H.pack_conv_weights_bf16x3_v32(wt) if a.v32 else H.pack_conv_weights_bf16x3(wt)
else:
    wp, impl = H.pack_conv_weights(wt), H.CONV_MFMA
res = torch.randn((B, (d - 1) // s + 1, (h - 1) // s + 1, (w - 1) // s + 1, cout), device=dev) if a.res else None
y = torch.empty((B, (d - 1) // s + 1, (h - 1) // s + 1, (w - 1) // s + 1, cout), device=dev)
for _ in range(3):
    H.conv3d(x, wt, wp, sc, sh, res=res, stride=s, impl=impl, out=y)
torch.cuda.synchronize()
s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record()
for _ in range(a.iters):
    H.conv3d(x, wt, wp, sc, sh, res=res, stride=s, impl=impl, out=y)
e0.record()
torch.cuda.synchronize()
us = s0.elapsed_time(e0) / a.iters * 1e3
print(f"us per launch: {us:.1f}  = {2.0 * 27 * cin * cout * y.numel() / cout / us / 1e6:.1f} TFLOP/s")
print(H.conv3d_variant(B, cin, d, h, w, cout, s, impl), float(y.abs().mean()))
