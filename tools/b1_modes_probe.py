#!/usr/bin/env python3
"""One G16V frame as one hipGraph replay in the two 16-bit splits (the robot's operating point): f16x3 0.484 ms, bf16x3 0.478 ms."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mvs_gi_amd import hip_ops as H, synth
from mvs_gi_amd.configs import CONFIGS
from mvs_gi_amd.pipeline import HotPath
cfg = CONFIGS["G16V"]
inp = synth.make_inputs(cfg, seed=1, batch=1)
w = synth.make_weights(cfg, seed=1)
feats = torch.from_numpy(inp["feats"]).cuda()
for mode in ("f16x3", "bf16x3"):
    H.set_conv_mode(mode)
    hp = HotPath(cfg, w, inp, device="cuda:0")
    for _ in range(3): hp(feats)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hp(feats)
        with torch.cuda.graph(g, stream=s):
            out = hp(feats)
    torch.cuda.synchronize()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): g.replay()
    torch.cuda.synchronize()
    print(mode, "B=1 graph replay ms", (time.perf_counter() - t0) / 500 * 1e3)
