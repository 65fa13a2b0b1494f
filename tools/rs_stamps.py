#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the register-stationary conv (diagnostic build only, never the product library):
builds libmvsgi_hip_stamps.so with -DMVSGI_RS_STAMPS, runs one 32->32 layer twice and prints, per wave of workgroup 8,
the median ticks of every segment of a phase (see the STAMP() comments in csrc/conv3d_rs.hip)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "MVSGI_LIB" not in os.environ:
    import __graft_entry__ as g
    abl = int(os.environ.get("RS_ABL", "0"))
    lib = g.build_stamps(abl)
    env = dict(os.environ, MVSGI_LIB=lib, MVSGI_STAMP="2")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
    rows = [l for l in r.stderr.splitlines() if l.startswith("wave ")]
    if r.returncode or not rows:
        print(r.stdout[-2000:], r.stderr[-4000:])
        sys.exit(1)
    if os.environ.get("RS_RAW"):      # short launches (a frame): [entry, weights resident, image 0 landed, then per phase: start, body end, waits done ..., exit]
        for l in rows[-4:]:
            t = [int(v) for v in l.split(":")[1].split()]
            print(l.split(":")[0], [v for i, v in enumerate(t) if v > 0 or i == 0])
        sys.exit(0)
    import statistics
    names = ["barrier -> phase start", "phase body (336 MFMAs + fillers)", "wait vmcnt / lgkmcnt"]
    for l in rows[-4:]:
        t = [int(v) for v in l.split(":")[1].split()]
        t = [v for i, v in enumerate(t) if v > 0 or i == 0]
        t = t[3:]                                   # (entry, weights resident, image 0 landed)
        ph = [t[i:i + 3] for i in range(0, len(t) - 3, 3)]
        segs = [[] for _ in range(3)]
        for k in range(3, len(ph) - 3):          # steady-state phases only
            segs[0].append(ph[k][0] - ph[k - 1][2])
            segs[1].append(ph[k][1] - ph[k][0])
            segs[2].append(ph[k][2] - ph[k][1])
        tot = sum(statistics.median(sg) for sg in segs if sg)
        print(l.split(":")[0], " | ".join(f"{nm}: {statistics.median(sg):.0f}" for nm, sg in zip(names, segs) if sg),
              f"| phase {tot:.0f} ticks = {tot / 336:.2f} per MFMA ({len(ph)} phases)")
    sys.exit(0)
import numpy as np
import torch
from mvs_gi_amd import hip_ops as H
B, d, h, w = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 8, 40, 160)))
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, 32), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((32, 32, 3, 3, 3)) / np.sqrt(27 * 32)).astype(np.float32)).to(dev)
wp = H.pack_conv_weights_rs(wt)
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
xs = H.act_to_split(x)
ys = H.SplitAct(B, d, h, w, 32, dev)
for _ in range(3):
    H.conv3d_rs(xs, wp, sc, sh, res=xs, out=ys)       # the third call prints the stamps of the second
torch.cuda.synchronize()
