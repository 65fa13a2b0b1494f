#!/usr/bin/env python3
"""Per-layer timing of the hot path on one GPU (development aid; HIP events on the launch stream)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H, synth  # noqa: E402
from mvs_gi_amd.configs import CONFIGS, path_gflop  # noqa: E402
from mvs_gi_amd.pipeline import HotPath  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def conv_layers(B, cfg):
    D = cfg.num_cands
    Hh, W = cfg.cv_hw
    C, cin0, f = cfg.vol_chs, cfg.reg_in_chs, cfg.reg_f_int_chs
    L = [("post_vol", C, C, D, Hh, W, 1, False)]
    d, h, w = D, Hh, W
    cin = cin0
    chs = [f, 2 * f, 4 * f]
    dims = []
    for lvl in range(3):
        L.append((f"down{lvl}.first", cin, chs[lvl], d, h, w, 2, False))
        d, h, w = (d - 1) // 2 + 1, (h - 1) // 2 + 1, (w - 1) // 2 + 1
        dims.append((d, h, w))
        L.append((f"down{lvl}.res x6", chs[lvl], chs[lvl], d, h, w, 1, True))
        cin = chs[lvl]
    L.append(("up0", chs[2], chs[1], *dims[1], 1, True))
    L.append(("up1", chs[1], chs[0], *dims[0], 1, True))
    L.append(("out0", chs[0], cin0, D, Hh, W, 1, False))
    L.append(("out1(head)", cin0, 1, D, Hh, W, 1, False))
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="G16V")
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--direct", action="store_true")
    ap.add_argument("--mode", default="f32")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    dev = "cuda:0"
    H.set_conv_mode(a.mode)
    rng = np.random.default_rng(0)
    for B in a.batch:
        print(f"== {a.config} B={B}")
        tot = 0.0
        for (name, cin, cout, d, h, w, s, res) in conv_layers(B, cfg):
            x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(dev)
            wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev)
            wp = H.pack_conv_weights(wt)
            wpb = H.pack_conv_weights_bf16x3(wt) if a.mode == "bf16x3" else None
            sc = torch.ones(cout, device=dev)
            sh = torch.zeros(cout, device=dev)
            do, ho, wo = (d - 1) // s + 1, (h - 1) // s + 1, (w - 1) // s + 1
            r = torch.zeros((B, do, ho, wo, cout), device=dev) if res else None
            y = torch.empty((B, do, ho, wo, cout), device=dev)
            impl = H.CONV_DIRECT if (a.direct or wp is None) else H.CONV_MFMA
            if wpb is not None:
                impl, wp = H.CONV_BF16X3, wpb
            us = timeit(lambda: H.conv3d(x, wt, wp, sc, sh, res=r, stride=s, impl=impl, out=y))
            gf = 2 * 27 * cin * cout * B * do * ho * wo / 1e9
            mult = 6 if "x6" in name else 1
            tot += us * mult
            print(f"  {name:16s} {cin:4d}->{cout:4d} out {do:3d}x{ho:3d}x{wo:3d}  {us:9.1f} us  {gf / us * 1e3:7.2f} TFLOP/s")
        print(f"  conv total (x6 applied): {tot:.1f} us  -> {path_gflop(cfg, B) / tot * 1e3:.2f} TFLOP/s")
        inp = synth.make_inputs(cfg, seed=0, batch=1)
        hp = HotPath(cfg, synth.make_weights(cfg, seed=0), inp, device=dev)
        feats = torch.from_numpy(rng.standard_normal((B, *inp["feats"].shape[1:]), dtype=np.float32)).to(dev)
        hp(feats)
        us = timeit(lambda: hp(feats), iters=10)
        print(f"  whole path: {us:.1f} us / batch -> {B / us * 1e6:.1f} frames/s")
        g, gm, m = hp.grids, hp.grid_masks, hp.masks
        for layout in ("auto", "nchw"):
            if cfg.builder == "std":
                us = timeit(lambda: H.sweep_std(feats, g, gm, m, layout=layout))
            else:
                us = timeit(lambda: H.sweep_cat(feats, g, layout=layout))
            print(f"  sweep[{layout}]: {us:.1f} us")
        fcl = H._feats_nhwc(feats).permute(0, 1, 4, 2, 3)
        us = timeit(lambda: H.sweep_std(fcl, g, gm, m) if cfg.builder == "std" else H.sweep_cat(fcl, g))
        print(f"  sweep[channels-last input, no transpose]: {us:.1f} us")
        c = torch.zeros((B, cfg.num_cands, *cfg.cv_hw), device=dev)
        us = timeit(lambda: H.softargmin(c, hp.dist_regressor.inv_dist_idx, 2, True))
        print(f"  softargmin(+norm_costs): {us:.1f} us")
        x = torch.zeros((B, cfg.num_cands // 2, cfg.cv_hw[0] // 2, cfg.cv_hw[1] // 2, cfg.reg_f_int_chs), device=dev)
        us = timeit(lambda: H.resize_trilinear(x, (cfg.num_cands, *cfg.cv_hw)))
        print(f"  resize out0 input: {us:.1f} us")


if __name__ == "__main__":
    main()
