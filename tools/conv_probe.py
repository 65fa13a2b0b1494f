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
a = ap.parse_args()
B, cin, cout, d, h, w, s = a.shape
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev)
if a.mode == "bf16x3":
    wp, impl = H.pack_conv_weights_bf16x3(wt), H.CONV_BF16X3
else:
    wp, impl = H.pack_conv_weights(wt), H.CONV_MFMA
sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
y = torch.empty((B, (d - 1) // s + 1, (h - 1) // s + 1, (w - 1) // s + 1, cout), device=dev)
for _ in range(3):
    H.conv3d(x, wt, wp, sc, sh, stride=s, impl=impl, out=y)
torch.cuda.synchronize()
s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record()
for _ in range(a.iters):
    H.conv3d(x, wt, wp, sc, sh, stride=s, impl=impl, out=y)
e0.record()
torch.cuda.synchronize()
print("us per launch:", s0.elapsed_time(e0) / a.iters * 1e3)
print(H.conv3d_variant(B, cin, d, h, w, cout, s, impl), float(y.abs().mean()))
