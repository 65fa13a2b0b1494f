#!/usr/bin/env python3
"""Time the candidate-walking sweep (split-padded output, one shared rig) alone: tools/sweep_split_probe.py [--batch 64]
(MVSGI_LIB selects a diagnostic build)."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H, synth  # noqa: E402
from mvs_gi_amd.configs import CONFIGS  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
cfg = CONFIGS["G16V"]
dev = "cuda:0"
inp = synth.make_inputs(cfg, seed=0, batch=1)
B = a.batch
g = torch.Generator(device=dev).manual_seed(0)
N, C, Hi, Wi = inp["feats"].shape[1:]
feats = torch.randn((B, N, Hi, Wi, C), device=dev, generator=g).permute(0, 1, 4, 2, 3)      # channels-last storage
grids = torch.from_numpy(inp["grids"]).to(dev)
vm = H.sweep_validity(grids, torch.from_numpy(inp["grid_masks"]).to(dev), torch.from_numpy(inp["masks"]).to(dev))
D, Ho, Wo = grids.shape[2:5]
out = H.SplitAct(B, D, Ho, Wo, C, dev)
for _ in range(2):
    H.sweep_std_valid_split(feats, grids, vm, out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.iters):
    H.sweep_std_valid_split(feats, grids, vm, out)
e.record()
torch.cuda.synchronize()
print(f"{os.path.basename(os.environ.get('MVSGI_LIB', 'default')):36s} B={B}: {s.elapsed_time(e) / a.iters * 1e3:8.1f} us per sweep")
