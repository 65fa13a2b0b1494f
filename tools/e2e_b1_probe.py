#!/usr/bin/env python3
"""One frame, images -> inverse distance (G16V, full size): per-launch submission against InferencePipeline.capture() / replay()."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import synth
from mvs_gi_amd.configs import CONFIGS
from mvs_gi_amd.pipeline import InferencePipeline

cfg = CONFIGS["G16V"]
w = synth.make_weights(cfg, seed=0)
w["feature_extractor"] = synth.make_extractor_weights(0)
inp = synth.make_inputs(cfg, seed=0, batch=1)
pipe = InferencePipeline(cfg, w, inp, device="cuda:0")
Hi, Wi = cfg.feat_hw
imgs = torch.randint(0, 256, (cfg.num_cams, 4 * Hi, 4 * Wi, 3), dtype=torch.uint8, device="cuda:0")


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


e = timeit(lambda: pipe.forward_device(imgs))
pipe.capture(imgs)
g = timeit(lambda: pipe.replay())
print(f"images -> inverse distance, one frame: per-launch {e:.3f} ms, hipGraph replay {g:.3f} ms")
