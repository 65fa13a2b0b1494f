import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H
B, d, h, w = 1, 4, 8, 32
dev = "cuda:0"
idx = np.arange(B * d * h * w * 32, dtype=np.float32).reshape(B, d, h, w, 32)
x = torch.from_numpy(idx).to(dev)
tap = int(sys.argv[1]) if len(sys.argv) > 1 else 13
wt = torch.zeros((32, 32, 27), device=dev)
for c in range(32):
    wt[c, c, tap] = 1.0
wt = wt.reshape(32, 32, 3, 3, 3).contiguous()
wpr = H.pack_conv_weights_rs(wt)
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
xs = H.act_to_split(x)
ys = H.SplitAct(B, d, h, w, 32, dev)
H.conv3d_rs(xs, wpr, sc, sh, out=ys, neg_slope=1.0)
torch.cuda.synchronize()
y = H.act_from_split(ys).cpu().numpy()
kd, kh, kw = tap // 9, (tap // 3) % 3, tap % 3
ref = np.zeros_like(idx)
xp = np.pad(idx, ((0, 0), (1, 1), (1, 1), (1, 1), (0, 0)))
ref = xp[:, kd:kd + d, kh:kh + h, kw:kw + w]
bad = np.argwhere(y != ref)
print("tap", tap, "bad", len(bad), "of", y.size)
for b_ in bad[:24]:
    got = y[tuple(b_)]
    exp = ref[tuple(b_)]
    g = int(got) if np.isfinite(got) and abs(got) < 2**31 else None
    src = None
    if g is not None and 0 <= g < idx.size:
        src = np.unravel_index(g, idx.shape)
    print("out", tuple(b_), "expected", exp, "got", got, "= x", src)
