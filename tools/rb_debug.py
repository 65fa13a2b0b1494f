import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from mvs_gi_amd import hip_ops as H
dev="cuda:0"
torch.manual_seed(0)
N,Hh,W = [int(v) for v in sys.argv[1:4]] if len(sys.argv)>3 else (3,256,1024)
x = torch.randn((N,Hh,W,16), device=dev)
w1 = torch.randn((16,16,3,3), device=dev)/12; w2 = torch.randn((16,16,3,3), device=dev)/12
sc1 = torch.rand(16, device=dev)+0.5; sh1 = torch.randn(16, device=dev)*0.3
sc2 = torch.rand(16, device=dev)+0.5; sh2 = torch.randn(16, device=dev)*0.3
p1 = H.pack_conv2d_weights_bf16x3(w1); p2 = H.pack_conv2d_weights_bf16x3(w2)
ref = H.resblock2d(x, p1, sc1, sh1, p2, sc2, sh2, 0.01)
q1 = H.pack_resblock2d_split_weights(w1, sc1); q2 = H.pack_resblock2d_split_weights(w2, sc2)
xs = H.f32_to_split2d(x)
for mode in ("f32","split"):
    if mode=="f32":
        y = H.resblock2d_split(xs, q1, sh1, q2, sh2, 0.01)
    else:
        out = H.split2d_buffer(N,Hh,W,dev)
        y = H.split2d_to_f32(H.resblock2d_split(xs, q1, sh1, q2, sh2, 0.01, out_split=out))
    torch.cuda.synchronize()
    bad = ~torch.isfinite(y)
    d = (y-ref).abs()
    d[bad] = 1e9
    print(mode, "nan count", int(bad.sum()), "max diff", float(d.max()), "ref max", float(ref.abs().max()))
    big = (d.amax(dim=3) > 1e-2).nonzero()
    print(" n bad pixels", big.shape[0])
    if big.shape[0]:
        print(" first", big[:12].tolist())
        hs = torch.unique(big[:,1]); ws = torch.unique(big[:,2])
        print(" rows", hs[:40].tolist(), "\n cols", ws[:60].tolist())
