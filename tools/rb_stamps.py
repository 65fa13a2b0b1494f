#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the split residual-block kernel (diagnostic build only, never the product library): builds
libmvsgi_hip_stamps.so with -DMVSGI_RS_STAMPS, runs one block on [N, 256, 1024, 16] three times and prints, per wave of
workgroup 8, the median ticks (100 MHz) of every segment of a brick: wait for the window | barrier | conv1 | epilogue A |
barrier + DMA issue | conv2 | epilogue B."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "MVSGI_LIB" not in os.environ:
    import __graft_entry__ as g
    lib = g.build_stamps(0)
    env = dict(os.environ, MVSGI_LIB=lib, MVSGI_STAMP="2")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
    rows = [l for l in r.stderr.splitlines() if l.startswith("rbwave ")]
    if r.returncode or not rows:
        print(r.stdout[-2000:], r.stderr[-4000:])
        sys.exit(1)
    import statistics
    names = ["wait vmcnt", "barrier", "->conv1", "conv1", "epi A", "barrier+dma", "conv2", "epi B + loop"]
    NS = 7
    for l in rows[-4:]:
        t = [int(v) for v in l.split(":")[1].split()]
        t = [v for i, v in enumerate(t) if v > 0 or i == 0]
        ph = [t[i:i + NS] for i in range(0, len(t) - NS, NS)]
        segs = [[] for _ in range(NS + 1)]
        for k in range(3, len(ph) - 2):
            segs[0].append(ph[k][1] - ph[k][0])
            segs[1].append(ph[k][2] - ph[k][1])
            segs[3].append(ph[k][3] - ph[k][2])
            segs[4].append(ph[k][4] - ph[k][3])
            segs[5].append(ph[k][5] - ph[k][4])
            segs[6].append(ph[k][6] - ph[k][5])
            segs[7].append(ph[k + 1][0] - ph[k][6])
        tot = sum(statistics.median(sg) for sg in segs if sg)
        print(l.split(":")[0], " | ".join(f"{nm}: {statistics.median(sg):.0f}" for nm, sg in zip(names, segs) if sg),
              f"| brick {tot:.0f} ticks ({len(ph)} bricks)")
    print(r.stdout[-300:])
    sys.exit(0)
import time
import numpy as np
import torch
from mvs_gi_amd import hip_ops as H
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
Hh, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 1024)
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.randn((N, Hh, W, 16), device=dev)
sc, sh = torch.ones(16, device=dev), torch.zeros(16, device=dev)
w1 = H.pack_resblock2d_split_weights(torch.randn((16, 16, 3, 3), device=dev) / 12, sc)
w2 = H.pack_resblock2d_split_weights(torch.randn((16, 16, 3, 3), device=dev) / 12, sc)
xs = H.f32_to_split2d(x)
ys = H.split2d_buffer(N, Hh, W, dev)
for _ in range(3):
    H.resblock2d_split(xs, w1, sh, w2, sh, 0.01, out_split=ys)       # the third call prints the stamps of the second
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    H.resblock2d_split(xs, w1, sh, w2, sh, 0.01, out_split=ys)
torch.cuda.synchronize()
print(f"{N}x{Hh}x{W}: {(time.perf_counter() - t0) / 5 * 1e6:.0f} us per block (stamped build)")
