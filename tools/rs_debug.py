import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H
B, d, h, w = (int(v) for v in sys.argv[1:5])
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, 32), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((32, 32, 3, 3, 3)) / np.sqrt(27 * 32)).astype(np.float32)).to(dev)
wp, wpr = H.pack_conv_weights_bf16x3(wt), H.pack_conv_weights_rs(wt)
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
y_ref = H.conv3d(x, wt, wp, sc, sh, impl=H.CONV_BF16X3, neg_slope=1.0)
xs = H.act_to_split(x)
ys = H.SplitAct(B, d, h, w, 32, dev)
H.conv3d_rs(xs, wpr, sc, sh, out=ys, neg_slope=1.0)
torch.cuda.synchronize()
y = H.act_from_split(ys)
e = (y - y_ref).abs()
e = torch.where(torch.isfinite(e), e, torch.full_like(e, 1e9))
print("overall max", float(e.max()), "nan count", int((~torch.isfinite(y)).sum()))
for dd in range(min(d, 2)):
    for hh in range(min(h, 4)):
        sub = e[:, dd::2, hh::4]
        print(f"d%2={dd} h%4={hh}: max {float(sub.max()):.3e}  per-cout-quarter", [f"{float(sub[..., c:c+8].max()):.2e}" for c in range(0, 32, 8)])
print("by w%16:", [f"{float(e[:, :, :, ww::16].max()):.1e}" for ww in range(16)])
print("raw words of y split at voxel (0,0,2,0):", ys.buf[0, 1, 3, 1, :8].tolist())
