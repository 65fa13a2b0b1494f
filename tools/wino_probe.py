#!/usr/bin/env python3
"""Winograd-form 32->32 conv (mvsgi_conv3d_wino32_f16) against the register-stationary direct kernel in the fp16 split and the
exact-fp32 kernel: parity and time.  With MVSGI_LIB=<a -DMVSGI_WINO_STAMPS build> and MVSGI_WINO_STAMP=1 every launch prints the
cycles workgroup 0's waves spent per step section (summed over the launch: before the span | - | span (MFMAs + transform + staging
requests) | exchange write + wait | barrier | epilogue + next raw reads)."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", type=int, nargs=4, default=[64, 8, 40, 160], help="B D H W")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--res", type=int, default=1)
a = ap.parse_args()
B, d, h, w = a.shape
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, 32), dtype=np.float32)).to(dev)
r = torch.from_numpy(rng.standard_normal((B, d, h, w, 32), dtype=np.float32)).to(dev) if a.res else None
wt = torch.from_numpy((rng.standard_normal((32, 32, 3, 3, 3)) / np.sqrt(27 * 32)).astype(np.float32)).to(dev)
sc = torch.from_numpy(rng.uniform(0.5, 1.5, 32).astype(np.float32)).to(dev)
sh = torch.from_numpy(rng.standard_normal(32).astype(np.float32) * 0.1).to(dev)
y_ref = H.conv3d(x, wt, H.pack_conv_weights(wt), sc, sh, res=r, impl=H.CONV_MFMA)          # exact fp32 MFMA
wpr, unr = H.pack_conv_weights_rs(wt, "f16")
wpw, unw = H.pack_conv_weights_wino(wt)
xs = H.act_to_split(x, fmt="f16")
rs = H.act_to_split(r, fmt="f16") if a.res else None
yd = H.SplitAct(B, d, h, w, 32, dev)
yw = H.SplitAct(B, d, h, w, 32, dev)
H.conv3d_rs(xs, wpr, sc * unr, sh, res=rs, out=yd)
H.conv3d_wino(xs, wpw, sc * unw, sh, res=rs, out=yw)
y32 = H.conv3d_wino(xs, wpw, sc * unw, sh, res=rs, out_f32=True)
torch.cuda.synchronize()
print("fp32-output variant vs split output: max abs diff", float((y32 - H.act_from_split(yw)).abs().max()))
ref_max = float(y_ref.abs().max())
xp, rp = H.act_to_f32p(x), (H.act_to_f32p(r) if a.res else None)
yp = H.SplitAct(B, d, h, w, 32, dev)
H.conv3d_wino(xp, wpw, sc * unw, sh, res=rp, out=yp)
torch.cuda.synchronize()
for name, ys in (("direct f16x3", yd), ("winograd f16x3", yw), ("winograd f16x3, fp32-padded activations", yp)):
    y = H.act_from_f32p(ys) if ys.fmt == "f32p" else H.act_from_split(ys)
    err = float((y - y_ref).abs().max()) / ref_max
    print(f"{name}: max rel error vs exact fp32 {err:.3e}  finite {bool(torch.isfinite(y).all())}")
    if err > 1e-4:
        print("  per plane:", [f"{float((y[:, k] - y_ref[:, k]).abs().max()) / ref_max:.1e}" for k in range(d)])
        print("  per frame:", [f"{float((y[k] - y_ref[k]).abs().max()) / ref_max:.1e}" for k in range(min(B, 8))])
    bad = ((y - y_ref).abs() > 1e-3 * ref_max).nonzero()
    if bad.shape[0]:
        print("  bad voxels", bad.shape[0], bad[:8].tolist())
    b_ = ys.buf
    assert all(float(t.abs().max()) == 0 for t in (b_[:, 0], b_[:, -1], b_[:, :, 0], b_[:, :, -1], b_[:, :, :, 0], b_[:, :, :, -1])), "border written"


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


gf = 2 * 27 * 32 * 32 * B * d * h * w / 1e9
for rep in range(3):
    t_d = timeit(lambda: H.conv3d_rs(xs, wpr, sc * unr, sh, res=rs, out=yd), a.iters)
    t_w = timeit(lambda: H.conv3d_wino(xs, wpw, sc * unw, sh, res=rs, out=yw), a.iters)
    t_p = timeit(lambda: H.conv3d_wino(xp, wpw, sc * unw, sh, res=rp, out=yp), a.iters)
    print(f"direct {t_d:8.1f} us ({gf / t_d * 1e3:6.1f} TF)   winograd {t_w:8.1f} us ({gf / t_w * 1e3:6.1f} TF direct-equivalent)   x{t_d / t_w:.2f}   fp32-padded {t_p:8.1f} us x{t_d / t_p:.2f}")
