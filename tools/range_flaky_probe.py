#!/usr/bin/env python3
"""Does the headline step ever raise the range report on in-range inputs?  Builds the bench's step (StreamedHotPath, 2 x B frames of
N(0, 1) features) R times in one process and reads the flags behind the first steps of each.  `python tools/range_flaky_probe.py [R] [B]`"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H, synth  # noqa: E402
from mvs_gi_amd.configs import CONFIGS  # noqa: E402
from mvs_gi_amd.pipeline import StreamedHotPath  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = "cuda:0"
H.set_range_check("off")
print("flags right after loading the library:", H.saturation_flags(), flush=True)
cfg = CONFIGS["G16V"]
inp = synth.make_inputs(cfg, seed=0, batch=1)
w = synth.make_weights(cfg, seed=0)
rng = np.random.default_rng(0)
N, C, (Hi, Wi) = cfg.num_cams, cfg.feat_chs, cfg.feat_hw
bad = 0
for r in range(R):
    shp = StreamedHotPath(cfg, w, inp, device=dev, n_streams=2)
    feats = torch.from_numpy(rng.standard_normal((2 * B, N, Hi, Wi, C), dtype=np.float32)).to(dev).permute(0, 1, 4, 2, 3)
    for step in range(3):
        shp(feats)
        torch.cuda.synchronize()
        f = H.saturation_flags(clear=True)
        if f:
            bad += 1
            print(f"round {r} step {step}: flags {f}", flush=True)
    shp.capture(feats)
    for step in range(3):
        shp.replay()
        torch.cuda.synchronize()
        f = H.saturation_flags(clear=True)
        if f:
            bad += 1
            print(f"round {r} replay {step}: flags {f}", flush=True)
    del shp, feats
    torch.cuda.empty_cache()
print("steps with flags:", bad)
