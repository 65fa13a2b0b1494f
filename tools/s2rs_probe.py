#!/usr/bin/env python3
"""post_vol -> down.0.first timing (G16V level 0, B frames): the fp32 hand-over (rs16 fp32 out + streaming stride-2 kernel) against
the split-padded hand-over (rs16 split out + csrc/conv3d_s2rs.hip); MVSGI_S2RS_TH selects the brick height of the latter.
tools/s2rs_probe.py [B]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
D, Hh, W = 16, 80, 320
dev = "cuda:0"
torch.manual_seed(0)
chunk = 16
xs_in = H.act_to_split(torch.randn((chunk, D, Hh, W, 16), device=dev))
w16 = torch.randn((16, 16, 3, 3, 3), device=dev) / 20
w32 = torch.randn((32, 16, 3, 3, 3), device=dev) / 20
sc16, sh16 = torch.ones(16, device=dev), torch.zeros(16, device=dev)
sc32, sh32 = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
wp16 = H.pack_conv_weights_rs(w16)
wp_b3 = H.pack_conv_weights_bf16x3(w32)
wp_s2 = H.pack_conv_weights_s2rs(w32, sc32)
y32 = torch.empty((B, D, Hh, W, 16), device=dev)
x0s = H.SplitAct(B, D, Hh, W, 16, dev)
out_a = H.SplitAct(B, D // 2, Hh // 2, W // 2, 32, dev)
out_b = H.SplitAct(B, D // 2, Hh // 2, W // 2, 32, dev)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def post_f32():
    for i in range(0, B, chunk):
        H.conv3d_rs16(xs_in, wp16, sc16, sh16, out=y32[i:i + chunk])


def post_split():
    for i in range(0, B, chunk):
        H.conv3d_rs16(xs_in, wp16, sc16, sh16, out_split=H.SplitAct(chunk, D, Hh, W, 16, dev, buf=x0s.buf[i:i + chunk]))


def down_stream():
    H.conv3d_out_split(y32, wp_b3, sc32, sh32, out=out_a, stride=2, neg_slope=0.01)


def down_s2rs():
    H.conv3d_s2rs(x0s, wp_s2, sh32, out_b, neg_slope=0.01)


post_f32(); post_split(); down_stream(); down_s2rs()
torch.cuda.synchronize()
a, b = H.act_from_split(out_a), H.act_from_split(out_b)
print("max |streaming - s2rs| / max:", float((a - b).abs().max() / a.abs().max()))
print(f"B={B}: post_vol fp32 out {timeit(post_f32):.0f} us, split out {timeit(post_split):.0f} us; "
      f"down.0.first streaming {timeit(down_stream):.0f} us, s2rs {timeit(down_s2rs):.0f} us (TH={os.environ.get('MVSGI_S2RS_TH', '4')})")
