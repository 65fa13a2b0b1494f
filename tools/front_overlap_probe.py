#!/usr/bin/env python3
"""sweep -> post_vol (K2f) in chunks of k frames, serial on one stream vs software-pipelined on two: sweep(chunk i + 1) runs on a
side stream BESIDE post_vol(chunk i).  The two kernels are complementary (sweep: texture path, 96 registers, no LDS; post_vol: matrix
cores, LDS-DMA, one 240-register wave per SIMD), so their workgroups can share a CU.
tools/front_overlap_probe.py [--batch 64] [--chunk 16]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H, synth  # noqa: E402
from mvs_gi_amd.configs import CONFIGS  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--chunk", type=int, default=16)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
H.set_conv_mode("bf16x3")
cfg = CONFIGS["G16V"]
dev = torch.device("cuda:0")
inp = synth.make_inputs(cfg, seed=0, batch=1)
B, k = a.batch, a.chunk
g = torch.Generator(device=dev).manual_seed(0)
N, C, Hi, Wi = inp["feats"].shape[1:]
feats = torch.randn((B, N, Hi, Wi, C), device=dev, generator=g).permute(0, 1, 4, 2, 3)
grids = torch.from_numpy(inp["grids"]).to(dev)
vm = H.sweep_validity(grids, torch.from_numpy(inp["grid_masks"]).to(dev), torch.from_numpy(inp["masks"]).to(dev))
D, Ho, Wo = grids.shape[2:5]
w1 = torch.randn((16, 16, 3, 3, 3), device=dev, generator=g) * 0.05
wp1 = H.pack_conv_weights_rs(w1)
s16, z16 = torch.ones(16, device=dev), torch.zeros(16, device=dev)
raw = [H.SplitAct(k, D, Ho, Wo, 16, dev) for _ in range(2)]
vol = torch.empty((B, D, Ho, Wo, 16), device=dev)
ap2 = os.environ.get("PRIO", "1") == "1"
# post_vol's persistent grid (one 240-register wave per SIMD) must get its CUs FIRST, the sweep's blocks then fill the rest of
# the register file (two waves per SIMD beside it); the other way round five sweep waves per SIMD leave post_vol no room
hi = torch.cuda.Stream(device=dev, priority=-1) if ap2 else torch.cuda.Stream(device=dev)
torch.cuda.set_stream(hi)
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev, priority=0)
n = B // k


def serial():
    for i in range(n):
        H.sweep_std_valid_split(feats[i * k:(i + 1) * k], grids, vm, raw[0])
        H.conv3d_rs16(raw[0], wp1, s16, z16, neg_slope=0.01, out=vol[i * k:(i + 1) * k])


def pipelined():
    swept = [torch.cuda.Event() for _ in range(n)]
    convd = [torch.cuda.Event() for _ in range(n)]
    side.wait_stream(main)
    for i in range(n):
        with torch.cuda.stream(side):                      # sweep(i) on the side stream, into buffer i % 2
            if i >= 2:
                side.wait_event(convd[i - 2])              # post_vol(i - 2) has read that buffer
            H.sweep_std_valid_split(feats[i * k:(i + 1) * k], grids, vm, raw[i & 1])
            swept[i].record(side)
        main.wait_event(swept[i])
        H.conv3d_rs16(raw[i & 1], wp1, s16, z16, neg_slope=0.01, out=vol[i * k:(i + 1) * k])
        convd[i].record(main)
    main.wait_stream(side)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters * 1e3


serial()
ref = vol.clone()
pipelined()
torch.cuda.synchronize()
print("same bits:", bool(torch.equal(ref, vol)))
for name, fn in (("serial", serial), ("pipelined", pipelined), ("serial", serial), ("pipelined", pipelined)):
    print(f"{name:10s} {timeit(fn):8.1f} us per {B} frames (chunks of {k})")
