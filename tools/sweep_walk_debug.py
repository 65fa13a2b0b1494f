#!/usr/bin/env python3
"""Compare the rig-walk sweep with the per-frame kernel (two child processes, MVSGI_SWEEP_RIG_WALK=1/0)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    from mvs_gi_amd import hip_ops as H, synth
    from mvs_gi_amd.configs import CONFIGS
    cfg = CONFIGS["G16V"]
    B = int(sys.argv[3])
    inp = synth.make_inputs(cfg, seed=0, batch=1)
    g = torch.Generator(device="cuda").manual_seed(0)
    N, C, Hi, Wi = inp["feats"].shape[1:]
    feats = torch.randn((B, N, Hi, Wi, C), device="cuda", generator=g).permute(0, 1, 4, 2, 3)
    grids = torch.from_numpy(inp["grids"]).cuda()
    vm = H.sweep_validity(grids, torch.from_numpy(inp["grid_masks"]).cuda(), torch.from_numpy(inp["masks"]).cuda())
    D, Ho, Wo = grids.shape[2:5]
    out = H.SplitAct(B, D, Ho, Wo, C, "cuda")
    H.sweep_std_valid_split(feats, grids, vm, out)
    v = H.sweep_std_valid(feats, grids, vm)
    torch.cuda.synchronize()
    np.save(sys.argv[2] + "_split.npy", out.buf.cpu().numpy())
    np.save(sys.argv[2] + "_f32.npy", v.cpu().numpy())
    sys.exit(0)
B = sys.argv[1] if len(sys.argv) > 1 else "3"
for w in ("1", "0"):
    env = dict(os.environ, MVSGI_SWEEP_RIG_WALK=w)
    subprocess.run([sys.executable, __file__, "child", f"/tmp/sw{w}", B], check=True, env=env)
for kind in ("split", "f32"):
    a, b = np.load(f"/tmp/sw1_{kind}.npy"), np.load(f"/tmp/sw0_{kind}.npy")
    bad = np.argwhere(a != b)
    print(kind, a.shape, "mismatches", len(bad), "of", a.size)
    if len(bad):
        print(" first", bad[:5].tolist(), " last", bad[-3:].tolist())
        for ax in range(a.ndim):
            u = np.unique(bad[:, ax])
            print("  axis", ax, "distinct", len(u), u[:12].tolist())
