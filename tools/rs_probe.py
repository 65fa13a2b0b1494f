#!/usr/bin/env python3
"""Register-stationary 32->32 conv (mvsgi_conv3d_rs_split) against the streaming split-bf16 kernel: parity and time."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", type=int, nargs=4, default=[32, 8, 40, 160], help="B D H W")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--res", type=int, default=1)
a = ap.parse_args()
B, d, h, w = a.shape
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, 32), dtype=np.float32)).to(dev)
r = torch.from_numpy(rng.standard_normal((B, d, h, w, 32), dtype=np.float32)).to(dev) if a.res else None
wt = torch.from_numpy((rng.standard_normal((32, 32, 3, 3, 3)) / np.sqrt(27 * 32)).astype(np.float32)).to(dev)
wp = H.pack_conv_weights_bf16x3(wt)
wpr = H.pack_conv_weights_rs(wt)
sc = torch.from_numpy(rng.uniform(0.5, 1.5, 32).astype(np.float32)).to(dev)
sh = torch.from_numpy(rng.standard_normal(32).astype(np.float32) * 0.1).to(dev)
y_ref = H.conv3d(x, wt, wp, sc, sh, res=r, impl=H.CONV_BF16X3)
xs = H.act_to_split(x)
rs = H.act_to_split(r) if a.res else None
back = H.act_from_split(xs)
print("split round trip max rel", float((back - x).abs().max() / x.abs().max()))
ys = H.SplitAct(B, d, h, w, 32, dev)
H.conv3d_rs(xs, wpr, sc, sh, res=rs, out=ys)
torch.cuda.synchronize()
y = H.act_from_split(ys)
err = float((y - y_ref).abs().max() / y_ref.abs().max())
print("rs vs streaming kernel: max rel", err, "finite", bool(torch.isfinite(y).all()))
bad = ((y - y_ref).abs() > 1e-3 * y_ref.abs().max()).nonzero()
print("bad voxels", bad.shape[0], bad[:8].tolist())
assert ys.buf[:, 0].abs().max() == 0 and ys.buf[:, :, 0].abs().max() == 0 and ys.buf[:, :, :, 0].abs().max() == 0 \
    and ys.buf[:, -1].abs().max() == 0 and ys.buf[:, :, -1].abs().max() == 0 and ys.buf[:, :, :, -1].abs().max() == 0, "border written"


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


yo = torch.empty_like(y_ref)
for rep in range(3):
    t_old = timeit(lambda: H.conv3d(x, wt, wp, sc, sh, res=r, impl=H.CONV_BF16X3, out=yo), a.iters)
    t_new = timeit(lambda: H.conv3d_rs(xs, wpr, sc, sh, res=rs, out=ys), a.iters)
    gf = 2 * 27 * 32 * 32 * B * d * h * w / 1e9
    print(f"streaming {t_old:8.1f} us ({gf / t_old * 1e3:6.1f} TF)   register-stationary {t_new:8.1f} us ({gf / t_new * 1e3:6.1f} TF)")
t_cv = timeit(lambda: H.act_to_split(x, out=xs), a.iters)
print(f"f32 -> split conversion {t_cv:.1f} us")
