import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mvs_gi_amd import hip_ops as H
B, cin, cout, d, h, w, s = [int(x) for x in sys.argv[1:8]]
dev = "cuda:0"
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, d, h, w, cin), dtype=np.float32)).to(dev)
wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev)
wp = H.pack_conv_weights_bf16x3(wt)
sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
pass
for _ in range(3):
    y = H.conv3d(x, wt, wp, sc, sh, stride=s, impl=H.CONV_BF16X3)
torch.cuda.synchronize()
pass
y = H.conv3d(x, wt, wp, sc, sh, stride=s, impl=H.CONV_BF16X3)
torch.cuda.synchronize()
