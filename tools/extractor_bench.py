#!/usr/bin/env python3
"""Per-layer timing of the HIP feature extractor (development aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H
H.set_conv_mode(sys.argv[2] if len(sys.argv) > 2 else "bf16x3")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
rng = np.random.default_rng(0)

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

M = B * 3
imgs = torch.from_numpy(rng.random((M, 3, 512, 2048), dtype=np.float32)).to(dev)
w5 = torch.from_numpy((rng.standard_normal((16, 3, 5, 5)) / 9).astype(np.float32)).to(dev)
sc, sh = torch.ones(16, device=dev), torch.zeros(16, device=dev)
us = timeit(lambda: H.conv2d(imgs, w5, None, sc, sh, stride=2, in_nchw=True))
print(f"stem 5x5 s2 3->16 ({M} imgs): {us:.1f} us  {2*75*16*M*256*1024/us/1e6:.2f} TFLOP/s")
w3 = torch.from_numpy((rng.standard_normal((16, 16, 3, 3)) / 12).astype(np.float32)).to(dev)
wp = H.pack_conv2d_weights_bf16x3(w3)
for (h, w, s, n) in ((256, 1024, 1, 10), (256, 1024, 2, 1), (128, 512, 1, 21)):
    x = torch.from_numpy(rng.standard_normal((M, h, w, 16), dtype=np.float32)).to(dev)
    r = torch.zeros((M, (h - 1) // s + 1, (w - 1) // s + 1, 16), device=dev)
    for mode, impl, wpk in (("bf16x3", H.CONV_BF16X3, wp), ("direct", H.CONV_DIRECT, None)):
        us = timeit(lambda: H.conv2d(x, w3, wpk, sc, sh, res=r if s == 1 else None, stride=s, impl=impl), iters=5)
        gf = 2 * 9 * 256 * M * ((h - 1) // s + 1) * ((w - 1) // s + 1) / 1e9
        print(f"3x3 16->16 {h}x{w} s{s} [{mode}] x{n}: {us:.1f} us  {gf/us*1e3:.2f} TFLOP/s")
