#!/usr/bin/env python3
"""Run the sweep kernel repeatedly (for rocprofv3 --pmc)."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H, synth  # noqa: E402
from mvs_gi_amd.configs import CONFIGS  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="G16V")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--layout", default="auto")
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
cfg = CONFIGS[a.config]
dev = "cuda:0"
inp = synth.make_inputs(cfg, seed=0, batch=1)
B = a.batch
rng = np.random.default_rng(0)
feats = torch.from_numpy(rng.standard_normal((B, *inp["feats"].shape[1:]), dtype=np.float32)).to(dev)
g = torch.from_numpy(inp["grids"]).to(dev).expand(B, *inp["grids"].shape[1:]).contiguous()
gm = torch.from_numpy(inp["grid_masks"]).to(dev).expand(B, *inp["grid_masks"].shape[1:]).contiguous()
m = torch.from_numpy(inp["masks"]).to(dev).expand(B, *inp["masks"].shape[1:]).contiguous()
if a.layout == "cl":
    feats = H._feats_nhwc(feats).permute(0, 1, 4, 2, 3)
for _ in range(a.iters):
    v = H.sweep_std(feats, g, gm, m, layout="auto" if a.layout == "cl" else a.layout) if cfg.builder == "std" else H.sweep_cat(feats, g)
torch.cuda.synchronize()
print(float(v.abs().mean()))
