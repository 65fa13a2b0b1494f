#!/usr/bin/env python3
"""What would Winograd F(2x2x2, 3x3x3) cost in accuracy with split operands?  (the one lever left that cuts the
matrix work itself -- 64 multiplications per 8 outputs instead of 216, 3.4x -- now that the convolutions sit at the part's
power-capped matrix rate.)  CPU emulation in float64 of one layer, random N(0,1) activations and He-uniform weights:
  direct   y = sum_taps w x with both operands split (hi*hi + hi*lo + lo*hi)          -- what the kernels do today
  winograd Y = A^T [ sum_cin (G w G^T) . (B^T x B) ] A per 2x2x2 output tile, the TRANSFORMED operands split the same way
each in the bf16 and the fp16 split, against the exact float64 convolution: max |err| / max |y|.
Usage: winograd_split_emulation.py [Cin Cout D H W]"""
import sys

import numpy as np
import torch

torch.set_num_threads(8)
Cin, Cout, D, H, W = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (32, 32, 8, 16, 32)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((1, Cin, D, H, W)))
b = np.sqrt(6.0 / (27 * Cin))
w = torch.from_numpy(rng.uniform(-b, b, (Cout, Cin, 3, 3, 3)))
y_ref = torch.nn.functional.conv3d(x, w, padding=1)


def split(t, kind, per_out_scale_dim=None):
    """-> (hi, lo) float64 tensors of the 16-bit split `kind`; fp16: pre-scale by a power of two per leading index of `t` when asked"""
    dt = torch.bfloat16 if kind == "bf16" else torch.float16
    s = 1.0
    if kind == "f16" and per_out_scale_dim is not None:
        amax = t.abs().amax(dim=tuple(i for i in range(t.dim()) if i != per_out_scale_dim), keepdim=True).clamp_min(1e-30)
        s = torch.pow(2.0, torch.floor(torch.log2(1024.0 / amax)))
    ts = (t * s).float()
    hi = ts.to(dt).double()
    lo = (ts.double() - hi).float().to(dt).double()
    return hi / s, lo / s


def direct(kind):
    xh, xl = split(x, kind)
    wh, wl = split(w, kind, 0)
    c = lambda a, b_: torch.nn.functional.conv3d(a, b_, padding=1)
    return c(xh, wh) + c(xh, wl) + c(xl, wh)


Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd(kind):
    U = torch.einsum("ai,bj,ck,oiijk->oiabc".replace("oiijk", "onijk").replace("oiabc", "onabc"), G, G, G, w)       # [Cout, Cin, 4, 4, 4]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1, 1, 1))
    # tiles: stride 2, size 4 along each axis
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2).unfold(4, 4, 2)                       # [1, Cin, Td, Th, Tw, 4, 4, 4]
    V = torch.einsum("ai,bj,ck,zndhwijk->zndhwabc", Bt, Bt, Bt, t)
    Uh, Ul = split(U, kind, 0)
    Vh, Vl = split(V, kind)
    m = lambda u, v: torch.einsum("onabc,zndhwabc->zodhwabc", u, v)
    M = m(Uh, Vh) + m(Ul, Vh) + m(Uh, Vl)
    Y = torch.einsum("ia,jb,kc,zodhwabc->zodhwijk", At, At, At, M)               # [1, Cout, Td, Th, Tw, 2, 2, 2]
    return Y.permute(0, 1, 2, 5, 3, 6, 4, 7).reshape(1, Cout, D, H, W)


den = float(y_ref.abs().max())
print(f"layer {Cin} -> {Cout} on [{D}, {H}, {W}]: max |err| / max |y| against the float64 convolution")
for kind in ("bf16", "f16"):
    ed = float((direct(kind) - y_ref).abs().max()) / den
    ew = float((winograd(kind) - y_ref).abs().max()) / den
    print(f"  {kind:5s} split: direct {ed:.2e}   winograd F(2x2x2, 3x3x3) {ew:.2e}   ({ew / ed:.1f}x)")
