#!/usr/bin/env python3
"""Does running two half-batches on two HIP streams overlap the low-power kernels (sweep, head, soft-argmin) of one with the
power-capped convs of the other?  python tools/two_stream_probe.py [B_total] [n_streams]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H, synth
from mvs_gi_amd.configs import CONFIGS
from mvs_gi_amd.pipeline import HotPath
import bench

Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cfg = CONFIGS["G16V"]
H.set_conv_mode("bf16x3")
inp = synth.make_inputs(cfg, seed=0, batch=1)
w = synth.make_weights(cfg, seed=0)
rng = np.random.default_rng(0)


def run(ns, steps=40):
    b = Bt // ns
    hps = [HotPath(cfg, w, inp, device=dev) for _ in range(ns)]
    feats = [bench.make_feats(b, inp["feats"].shape, rng, dev, torch, np) for _ in range(ns)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    for hp, f in zip(hps, feats):
        hp(f)
    torch.cuda.synchronize()

    def step():
        for hp, f, s in zip(hps, feats, streams):
            with torch.cuda.stream(s):
                hp(f)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{ns} stream(s) x B={b}: {Bt * steps / el:.1f} frames/s, {el / steps * 1e3:.3f} ms per {Bt} frames")
    del hps, feats
    torch.cuda.empty_cache()


for ns in (1, 2, 4, 1, 2):
    run(ns)
