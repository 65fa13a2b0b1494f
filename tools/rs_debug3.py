import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H
B, d, h, w = 1, 4, 8, 32
dev = "cuda:0"
idx = np.arange(B * d * h * w * 32, dtype=np.float32).reshape(B, d, h, w, 32)
x = torch.from_numpy(idx).to(dev)
xs = H.act_to_split(x)
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
xp = np.pad(idx, ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)))
for tap in range(27):
    wt = torch.zeros((32, 32, 27), device=dev)
    for c in range(32):
        wt[c, c, tap] = 1.0
    wpr = H.pack_conv_weights_rs(wt.reshape(32, 32, 3, 3, 3).contiguous())
    ys = H.SplitAct(B, d, h, w, 32, dev)
    H.conv3d_rs(xs, wpr, sc, sh, out=ys, neg_slope=1.0)
    torch.cuda.synchronize()
    y = H.act_from_split(ys).cpu().numpy()
    kd, kh, kw = tap // 9, (tap // 3) % 3, tap % 3
    ref = xp[:, kd:kd + d, kh:kh + h, kw:kw + w]
    bad = np.argwhere(y != ref)
    msg = ""
    if len(bad):
        b_ = tuple(bad[0])
        g = y[b_]
        src = np.unravel_index(int(g), idx.shape) if 0 <= g < idx.size else None
        msg = f" first bad out {b_[1:]} expected x{np.unravel_index(int(ref[b_]), idx.shape)[1:] if ref[b_] > 0 else 0} got {g} = x{src[1:] if src else None}"
    print(f"tap {tap} ({kd},{kh},{kw}): bad {len(bad)}{msg}")
