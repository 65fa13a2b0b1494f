#!/usr/bin/env python3
"""CPU emulation of split-operand conv arithmetics on the full-size goldens: which 16-bit split keeps inv_dist inside the bar
on a sharp softmax?  Every Conv3d of the oracle's forward is replaced by  conv(x_hi, w_hi) + conv(x_hi, w_lo) + conv(x_lo, w_hi)
in float64 (so only the SPLIT's error is seen), with hi / lo in
  bf16      the library's bf16x3 (hi = bf16(x), lo = bf16(x - hi))
  f16       hi = fp16(x), lo = fp16(x - hi); weights pre-scaled per cout by a power of two (undone exactly afterwards)
  f16-ftz   the same with fp16 subnormals flushed to zero (what a matrix core without subnormal support would do)
  f16-nows  fp16 split with NO weight pre-scaling (lo parts of the weights are fp16 subnormals: absolute quantum 2^-24)
  f16-x16   activations pre-scaled by 16 (exactly undone) so that the lo parts of |x| >= 0.008 are normal numbers
Usage: split_arith_emulation.py [case] [gain ...]   (needs tests/golden/<case>_ladder.npz; no GPU, no reference)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_cases import FULL_CASES  # noqa: E402
from mvs_gi_amd import synth  # noqa: E402
from oracle import mvsgi_oracle as O  # noqa: E402

_conv3d = F.conv3d


def _q(t, dtype, ftz):
    q = t.to(dtype).to(torch.float64)
    if ftz:
        q = torch.where(q.abs() < 2.0 ** -14, torch.zeros_like(q), q)
    return q


def make_emu(kind):
    dtype = torch.bfloat16 if kind == "bf16" else torch.float16
    ftz = "ftz" in kind
    xs = float(kind.split("-x")[1].split("-")[0]) if "-x" in kind else 1.0

    def emu(x, w, b=None, stride=1, padding=0):
        x64, w64 = x.double() * xs, w.double()
        if dtype == torch.float16 and "nows" not in kind:
            k = torch.floor(torch.log2(1024.0 / w64.abs().amax(dim=(1, 2, 3, 4), keepdim=True).clamp_min(1e-30)))
            ws = torch.pow(2.0, k)
            w64 = w64 * ws
        else:
            ws = None
        xh = _q(x64.float(), dtype, ftz)
        xl = _q((x64 - xh).float(), dtype, ftz)
        wh = _q(w64.float(), dtype, ftz)
        wl = _q((w64 - wh).float(), dtype, ftz)
        y = _conv3d(xh, wh, None, stride=stride, padding=padding) + _conv3d(xh, wl, None, stride=stride, padding=padding) + \
            _conv3d(xl, wh, None, stride=stride, padding=padding)
        if ws is not None:
            y = y / ws.view(1, -1, 1, 1, 1)
        y = y / xs
        if b is not None:
            y = y + b.double().view(1, -1, 1, 1, 1)
        return y.float()
    return emu


name = sys.argv[1] if len(sys.argv) > 1 else "full_G16V"
gains = [float(g) for g in sys.argv[2:]] or [4.0, 16.0]
case = FULL_CASES[name]
cfg = case["cfg"]
G = os.path.join(ROOT, "tests", "golden")
z, zl = np.load(os.path.join(G, name + ".npz")), np.load(os.path.join(G, name + "_ladder.npz"))
inp = synth.make_inputs(cfg, seed=case["seed"], batch=case["batch"], grid_kind=case["grid_kind"], grid_mask_dtype=case["grid_mask_dtype"])
assert synth.digest(inp) == str(z["inputs_sha256"])
t = O.to_torch(inp)
torch.set_num_threads(8)
for gain in gains:
    key = f"inv_dist_g{gain:g}"
    ref = zl[key] if key in zl.files else z[key]
    w = O.to_torch(synth.make_weights(cfg, seed=case["seed"], gain=gain))
    row = []
    for kind in (os.environ.get("KINDS", "fp32,bf16,f16,f16-ftz,f16-x16").split(",")):
        F.conv3d = _conv3d if kind == "fp32" else make_emu(kind)
        try:
            with torch.no_grad():
                got = O.hot_path(t["feats"], t["grids"], t["grid_masks"], t["masks"], w, cfg.builder, cfg.dist_cands, cfg.bf,
                                 cfg.interp_scale_factor, cfg.pre_interp).numpy()
        finally:
            F.conv3d = _conv3d
        row.append((kind, float(np.abs(got - ref).max() / np.abs(ref).max())))
        print(f"{name} gain {gain:g} {kind:8s}: max-rel {row[-1][1]:.3e}", flush=True)
