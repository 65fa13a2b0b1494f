#!/usr/bin/env python3
"""Why the sweep does not stage feature tiles in LDS (review item: "LDS staging of D-candidate tiles").

For the benchmark's rig (synth.smooth_grids, G16V) and the sweep's work unit -- 64 consecutive wo of one (b, ho) row, one
candidate, one camera -- this counts, in 64-byte feature texels,
  gathered : the 4 x 64 bilinear taps the kernel requests (what the texture path moves, duplicates included),
  unique   : the distinct texels among them (what a perfect software cache would fetch once),
  bbox     : the bounding box [y0, y1] x [x0, x1] of those taps (what a coalesced rectangular LDS stage must copy),
  rows     : distinct feature rows touched, and the same for all D candidates of the unit together (the "D-candidate tile").
CPU only; writes the table to stdout.  python tools/sweep_footprint.py > profiles/r03_sweep_footprint.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import synth                      # noqa: E402
from mvs_gi_amd.configs import CONFIGS            # noqa: E402


def main():
    cfg = CONFIGS["G16V"]
    g, gm, m = synth.smooth_grids(cfg)
    N, D, Ho, Wo, _ = g.shape
    Hi, Wi = cfg.feat_hw
    x = ((g[..., 0].astype(np.float64) + 1) * Wi - 1) / 2
    y = ((g[..., 1].astype(np.float64) + 1) * Hi - 1) / 2
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    rows = {k: [] for k in ("gathered", "unique", "bbox", "nrows")}
    allD = {k: [] for k in ("unique", "bbox", "nrows")}
    for cam in range(N):
        for ho in range(Ho):
            for wt in range((Wo + 63) // 64):
                sl = slice(wt * 64, min(Wo, wt * 64 + 64))
                ud = set()
                for d in range(D):
                    taps = set()
                    n_in = 0
                    for dy in (0, 1):
                        for dx in (0, 1):
                            xx, yy = x0[cam, d, ho, sl] + dx, y0[cam, d, ho, sl] + dy
                            ok = (xx >= 0) & (xx < Wi) & (yy >= 0) & (yy < Hi)
                            n_in += int(ok.sum())
                            taps.update(zip(yy[ok].tolist(), xx[ok].tolist()))
                    if not taps:
                        continue
                    ys = [t[0] for t in taps]
                    xs = [t[1] for t in taps]
                    rows["gathered"].append(n_in)
                    rows["unique"].append(len(taps))
                    rows["bbox"].append((max(ys) - min(ys) + 1) * (max(xs) - min(xs) + 1))
                    rows["nrows"].append(len(set(ys)))
                    ud |= taps
                if ud:
                    ys = [t[0] for t in ud]
                    xs = [t[1] for t in ud]
                    allD["unique"].append(len(ud))
                    allD["bbox"].append((max(ys) - min(ys) + 1) * (max(xs) - min(xs) + 1))
                    allD["nrows"].append(len(set(ys)))

    def stat(v):
        v = np.asarray(v, np.float64)
        return f"mean {v.mean():9.1f}  p50 {np.percentile(v, 50):8.0f}  p90 {np.percentile(v, 90):8.0f}  max {v.max():8.0f}"
    print(__doc__.split("CPU only")[0].strip())
    print()
    print(f"rig: synth.smooth_grids(G16V): {N} cameras, D = {D}, cv {Ho} x {Wo}, features {Hi} x {Wi} x 16 ch (64 B per texel)")
    print(f"units: {len(rows['gathered'])} (camera, candidate, ho, 64-wo tile) with at least one tap inside the image\n")
    print("per (unit, candidate, camera), in texels:")
    for k in ("gathered", "unique", "bbox", "nrows"):
        print(f"  {k:9s} {stat(rows[k])}")
    print("per (unit, camera), all D candidates together:")
    for k in ("unique", "bbox", "nrows"):
        print(f"  {k:9s} {stat(allD[k])}")
    gsum, usum, bsum = sum(rows["gathered"]), sum(rows["unique"]), sum(rows["bbox"])
    print(f"\ntotals per frame: gathered {gsum * 64 / 1e6:.1f} MB, unique {usum * 64 / 1e6:.1f} MB, bbox {bsum * 64 / 1e6:.1f} MB "
          f"(bbox / gathered = {bsum / gsum:.2f}, unique / gathered = {usum / gsum:.2f})")
    print(f"all-D tiles: unique {sum(allD['unique']) * 64 / 1e6:.1f} MB, bbox {sum(allD['bbox']) * 64 / 1e6:.1f} MB per frame; "
          f"largest all-D bbox {max(allD['bbox']) * 64 / 1024:.0f} KiB per camera (LDS: 160 KiB per CU)")


if __name__ == "__main__":
    main()
