"""One conv launch of the stamps build (MVSGI_LIB=...stamps.so MVSGI_STAMP=2): prints the s_memtime stamps of block 8.
usage: stamp_probe.py B cin cout d h w stride [up2] [c16] [v32] [d32]   (up2: d h w are the LOW-resolution sizes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mvs_gi_amd import hip_ops as H
B, cin, cout, d, h, w, s = [int(x) for x in sys.argv[1:8]]
up2, c16, v32, d32 = "up2" in sys.argv[8:], "c16" in sys.argv[8:], "v32" in sys.argv[8:], "d32" in sys.argv[8:]
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev)
wp = H.pack_conv_weights_bf16x3_c16(wt) if c16 else (H.pack_conv_weights_bf16x3_v32(wt) if v32 else (H.pack_conv_weights_bf16x3_d32(wt) if d32 else H.pack_conv_weights_bf16x3(wt)))
impl = H.CONV_BF16X3_C16 if c16 else (H.CONV_BF16X3_V32 if v32 else (H.CONV_BF16X3_D32 if d32 else H.CONV_BF16X3))
sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)


def run():
    if up2:
        return H.conv3d_up2(x, wp, sc, sh, w_layout=impl)
    return H.conv3d(x, wt, wp, sc, sh, stride=s, impl=impl)


for _ in range(3):
    y = run()
torch.cuda.synchronize()
s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record()
y = run()
e0.record()
torch.cuda.synchronize()
print("us:", s0.elapsed_time(e0) * 1e3, H.conv3d_up2_variant(B, cin, d, h, w, cout, impl) if up2 else H.conv3d_variant(B, cin, d, h, w, cout, s, impl))
