#!/usr/bin/env python3
"""Which launch raises the fp16 split's range report?  Wraps every launcher of mvs_gi_amd.hip_ops, synchronises behind each call and
reads (and clears) mvsgi_saturation_flags: prints the calls that raised a bit, with their tensor shapes and the largest |value| of
their fp32 arguments.  Default workload: the images -> inverse distance chain of bench.py's extras (random-init extractor, random
uint8 images), B frames.  `python tools/range_trace.py [B]`"""
import inspect
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvs_gi_amd import dropin, hip_ops as H, synth  # noqa: E402
from mvs_gi_amd.configs import CONFIGS  # noqa: E402
from mvs_gi_amd.pipeline import HotPath  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = "cuda:0"
H.set_range_check("off")
cfg = CONFIGS["G16V"]
inp = synth.make_inputs(cfg, seed=0, batch=1)
hp = HotPath(cfg, synth.make_weights(cfg, seed=0), inp, device=dev)
N, (Hi, Wi) = cfg.num_cams, cfg.feat_hw
fe = dropin.SimpleFeatExtraction(in_size=(4 * Hi, 4 * Wi), in_chs=3, chs=cfg.feat_chs, k_sz=3, layers=[5, 10])
fe.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_extractor_weights(0).items()}, strict=True)
fe = fe.eval().to(dev)
rng = np.random.default_rng(0)
imgs = torch.from_numpy(rng.integers(0, 256, (B * N, 4 * Hi, 4 * Wi, 3), dtype=np.uint8)).to(dev)


def describe(a):
    if isinstance(a, torch.Tensor):
        return f"{tuple(a.shape)}{' max|x| %.3g' % float(a.abs().max()) if a.dtype == torch.float32 and a.numel() else ''}"
    if isinstance(a, H.SplitAct):
        return f"SplitAct{a.shape}:{a.fmt}"
    return repr(a) if isinstance(a, (int, float, bool, str)) else type(a).__name__


def wrap(name, fn):
    def w(*a, **k):
        r = fn(*a, **k)
        torch.cuda.synchronize()
        f = H.saturation_flags(clear=True)
        if f:
            out = describe(r) if not isinstance(r, tuple) else ", ".join(describe(x) for x in r)
            print(f"flags {f}: {name}({', '.join(describe(x) for x in a)}) -> {out}", flush=True)
        return r
    return w


for name, fn in list(vars(H).items()):
    if inspect.isfunction(fn) and fn.__module__ == H.__name__ and (name.startswith(("conv", "sweep", "act_", "resblock", "f32_to", "softargmin", "deform"))):
        setattr(H, name, wrap(name, fn))
with torch.no_grad():
    f = fe(imgs)
    print("features:", tuple(f.shape), "max |f| %.4g, std %.4g" % (float(f.abs().max()), float(f.std())))
    inv, _ = hp(f.reshape(B, N, *f.shape[1:]))
torch.cuda.synchronize()
print("inv_dist finite:", bool(torch.isfinite(inv).all()), "flags left:", H.saturation_flags(clear=True))
