#!/usr/bin/env python3
"""Where does each arithmetic of the HIP path cross the 1e-3 bar?  Every full-size case of tests/golden_cases.py at its own gains
and up its gain ladder (tests/golden/<case>_ladder.npz: the REFERENCE's inv_dist per rung, tools/make_goldens.py ladder), in
  bf16x3            the library default
  f16x3             MVSGI_CONV_MODE=f16x3 (the fp16 split)
  bf16x3+head_f32   MVSGI_HEAD_SPLIT=0 (exact-fp32 cost head)
  f32               MVSGI_CONV_MODE=f32
Prints max-rel (max |d| / max |ref|) per row.  (Round 5 also measured out_costs.0 + out_costs.1 in exact fp32 with every other layer in
bf16x3: 0.76-0.85 of the bf16x3 error -- the split's error is spread over the whole network -- so that switch was not kept.)  GPU box only; the reference is not needed (goldens are data)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_cases import FULL_CASES  # noqa: E402
from mvs_gi_amd import hip_ops as H, synth  # noqa: E402
from mvs_gi_amd.dropin import cost_volume_regulator as cr  # noqa: E402
from mvs_gi_amd.pipeline import HotPath  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
MODES = [("bf16x3", "bf16x3", True), ("bf16x3+head_f32", "bf16x3", False), ("f16x3", "f16x3", True), ("f32", "f32", True)]
which = sys.argv[1:] or list(FULL_CASES)
for name in which:
    case = FULL_CASES[name]
    cfg = case["cfg"]
    z, zl = np.load(os.path.join(G, name + ".npz")), np.load(os.path.join(G, name + "_ladder.npz"))
    inp = synth.make_inputs(cfg, seed=case["seed"], batch=case["batch"], grid_kind=case["grid_kind"], grid_mask_dtype=case["grid_mask_dtype"])
    assert synth.digest(inp) == str(z["inputs_sha256"]) == str(zl["inputs_sha256"]), "regenerated inputs differ from the golden run's"
    feats = torch.from_numpy(inp["feats"]).cuda()
    rows = [(float(g), z[f"inv_dist_g{g:g}"], None) for g in sorted(case["gains"])] + \
           [(float(g), zl[f"inv_dist_g{g:g}"], float(p)) for g, p in zip(zl["gains"], zl["mean_maxprob"])]
    print(f"== {name}: " + " | ".join(f"{m[0]:>16s}" for m in MODES), flush=True)
    for gain, ref, mp in rows:
        w = synth.make_weights(cfg, seed=case["seed"], gain=gain)
        errs = []
        for _, mode, head_split in MODES:
            H.set_conv_mode(mode)
            cr._HEAD_SPLIT = head_split
            hp = HotPath(cfg, w, inp, device="cuda:0")
            got = hp(feats)[0].cpu().numpy()
            errs.append(float(np.abs(got - ref).max() / np.abs(ref).max()))
            del hp
        torch.cuda.empty_cache()
        print(f"   gain {gain:7g} maxprob {mp if mp is not None else float('nan'):.4f}: " + " | ".join(f"{e:16.3e}" for e in errs), flush=True)
