import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H
B, d, h, w = 1, 4, 4, 16
dev = "cuda:0"
idx = np.arange(B * d * h * w * 16, dtype=np.float32).reshape(B, d, h, w, 16)
x = torch.from_numpy(idx).to(dev)
xs = H.act_to_split(x)
sc, sh = torch.ones(16, device=dev), torch.zeros(16, device=dev)
xp = np.pad(idx, ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)))
for tap in range(27):
    wt = torch.zeros((16, 16, 27), device=dev)
    for c in range(16):
        wt[c, c, tap] = 1.0
    wpr = H.pack_conv_weights_rs(wt.reshape(16, 16, 3, 3, 3).contiguous())
    y = H.conv3d_rs16(xs, wpr, sc, sh, neg_slope=1.0).cpu().numpy()
    kd, kh, kw = tap // 9, (tap // 3) % 3, tap % 3
    ref = xp[:, kd:kd + d, kh:kh + h, kw:kw + w]
    bad = np.argwhere(y != ref)
    msg = ""
    if len(bad):
        outs = sorted({(int(b[1]), int(b[2])) for b in bad})[:8]
        b_ = tuple(bad[0])
        g = y[b_]
        src = np.unravel_index(int(g), idx.shape)[1:] if 0 <= g < idx.size and g == int(g) else None
        msg = f" (d,h) {outs} first out {b_[1:]} expected {ref[b_]} got {g} = x{src}"
    print(f"tap {tap} ({kd},{kh},{kw}): bad {len(bad)}{msg}")
