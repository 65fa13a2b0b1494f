#!/usr/bin/env python3
"""Per-layer table of one configuration: every launch of the path keyed by (kernel, layer shape), HIP events on the launch stream.

  python tools/layer_table.py <config tag> [batch] [steps]      # e.g. G16VV 32 5
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import torch                # noqa: E402
import bench                # noqa: E402
from mvs_gi_amd import hip_ops as H, synth      # noqa: E402
from mvs_gi_amd.configs import CONFIGS          # noqa: E402
from mvs_gi_amd.pipeline import HotPath         # noqa: E402


def main():
    tag = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    cfg = CONFIGS[tag]
    dev = torch.device("cuda", 0)
    inp = synth.make_inputs(cfg, seed=0, batch=1)
    hp = HotPath(cfg, synth.make_weights(cfg, seed=0), inp, device=dev)
    rng = np.random.default_rng(0)
    feats = bench.make_feats(B, inp["feats"].shape, rng, dev, torch, np)
    with bench.ConvProbe(H) as probe:
        probe.by_shape = True
        for _ in range(3):
            hp(feats)
        torch.cuda.synchronize()
        probe.enabled = True
        for _ in range(steps):
            hp(feats)
        torch.cuda.synchronize()
        probe.enabled = False
        agg, hbm = probe.summary(), probe.hbm_summary()
    rows = [(v[2] / steps, k, v[0] // steps, v[1] / (v[2] * 1e-3) / 1e12, None) for k, v in agg.items()]
    rows += [(v[2] / steps, k, v[0] // steps, None, v[1] / (v[2] * 1e-3) / 1e9) for k, v in hbm.items()]
    tot = sum(r[0] for r in rows)
    print(f"# {tag}, {B} frames per step: {tot:.3f} ms attributed per step = {B / tot * 1e3:.1f} frames/s (per-launch submission with events)")
    print(f"{'ms/step':>9s} {'share':>6s} {'n':>3s} {'us/launch':>10s} {'TFLOP/s':>8s} {'GB/s':>8s}  kernel @ layer")
    for ms, k, n, tf, gb in sorted(rows, reverse=True):
        print(f"{ms:9.3f} {ms / tot:6.3f} {n:3d} {ms / max(n, 1) * 1e3:10.1f} {tf if tf is not None else float('nan'):8.1f} "
              f"{gb if gb is not None else float('nan'):8.1f}  {k}")


if __name__ == "__main__":
    main()
