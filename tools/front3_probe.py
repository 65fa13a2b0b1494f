#!/usr/bin/env python3
"""sweep -> post_vol (K2f) -> down.0.first (stride-2 16 -> 32, split-padded out) for B frames, whole batch at once or in chunks
of k frames (does the fp32 vol between post_vol and the stride-2 conv come back faster when it is a chunk old?):
tools/front3_probe.py [--batch 64] [--chunks 0,4,8,16]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H, synth  # noqa: E402
from mvs_gi_amd.configs import CONFIGS  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--chunks", default="0,4,8,16")
ap.add_argument("--iters", type=int, default=8)
a = ap.parse_args()
H.set_conv_mode("bf16x3")
cfg = CONFIGS["G16V"]
dev = "cuda:0"
inp = synth.make_inputs(cfg, seed=0, batch=1)
B = a.batch
g = torch.Generator(device=dev).manual_seed(0)
N, C, Hi, Wi = inp["feats"].shape[1:]
feats = torch.randn((B, N, Hi, Wi, C), device=dev, generator=g).permute(0, 1, 4, 2, 3)
grids = torch.from_numpy(inp["grids"]).to(dev)
vm = H.sweep_validity(grids, torch.from_numpy(inp["grid_masks"]).to(dev), torch.from_numpy(inp["masks"]).to(dev))
D, Ho, Wo = grids.shape[2:5]
w1 = torch.randn((16, 16, 3, 3, 3), device=dev, generator=g) * 0.05
w2 = torch.randn((32, 16, 3, 3, 3), device=dev, generator=g) * 0.05
wp1, wp2 = H.pack_conv_weights_rs(w1), H.pack_conv_weights_bf16x3(w2)
s16, z16 = torch.ones(16, device=dev), torch.zeros(16, device=dev)
s32, z32 = torch.ones(32, device=dev), torch.zeros(32, device=dev)
out2 = H.SplitAct(B, D // 2, Ho // 2, Wo // 2, 32, dev)


def run(k):
    k = k if 0 < k < B else B
    raw = H.SplitAct(k, D, Ho, Wo, 16, dev)
    vol = torch.empty((k, D, Ho, Wo, 16), device=dev)

    def step():
        for i in range(0, B, k):
            H.sweep_std_valid_split(feats[i:i + k], grids, vm, raw)
            H.conv3d_rs16(raw, wp1, s16, z16, neg_slope=0.01, out=vol)
            o = H.SplitAct(k, D // 2, Ho // 2, Wo // 2, 32, dev, buf=out2.buf[i:i + k])
            H.conv3d_out_split(vol, wp2, s32, z32, out=o, stride=2, neg_slope=0.01)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        step()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters * 1e3


for k in [int(x) for x in a.chunks.split(",")]:
    print(f"chunk {k:3d}: {run(k):8.1f} us per {B} frames")
