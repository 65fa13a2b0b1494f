"""Diagnostic: the polyphase ResizeConv3d in parts (main kernel alone with the face / edge weights zeroed; the corrections alone)
against the float64 statement in dropin/polyphase.py."""
import sys, os
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvs_gi_amd import hip_ops as H
from mvs_gi_amd.dropin import polyphase as P  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import polyphase_ref as R  # noqa: E402

DEV = "cuda:0"
shape = tuple(int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (1, 1, 1, 1)))
B, d, h, w = shape
rng = np.random.default_rng(sum(shape))
x = torch.from_numpy(rng.standard_normal((B, d, h, w, 32), dtype=np.float32)).to(DEV)
wt = (rng.standard_normal((16, 32, 3, 3, 3)) / np.sqrt(27 * 32)).astype(np.float32)
sc = torch.ones(16, device=DEV)
sh = torch.zeros(16, device=DEV)
xs = H.act_to_split(x)
plan = H.conv3d_up2_poly_plan(torch.from_numpy(wt).to(DEV), d, h, w)
hdr = plan[:128].cpu().numpy()
ints = hdr[:32].view(np.int32)
offs = hdr[32:80].view(np.int64)
print("header", ints[:8], offs)
off_main, off_facew, off_roles, off_edgew, off_cells, total = (int(v) for v in offs)
xq = H.act_from_split(xs).cpu().permute(0, 4, 1, 2, 3).double().numpy()

def run(pl):
    y = torch.zeros((B, 2 * d, 2 * h, 2 * w, 16), device=DEV)
    H.conv3d_up2_poly(xs, pl, sc, sh, neg_slope=1.0, out=y)
    torch.cuda.synchronize()
    return y.cpu().numpy()

full = run(plan)
ref = R.reference_up2_conv(xq, wt).transpose(0, 2, 3, 4, 1)
print("full vs reference: max abs err", np.abs(full - ref).max(), "max ref", np.abs(ref).max())
p2 = plan.clone()
p2[off_facew:off_roles] = 0
p2[off_edgew:off_cells] = 0
main_only = run(p2)
# expectation: every cell with (class_d, INT, INT)
xp = np.pad(xq, ((0, 0), (0, 0), (1, 1), (1, 1), (1, 1)))
exp = np.zeros_like(ref)
for i_d in range(d):
    cd = P.cell_class(i_d, d)
    for pd in range(2):
        for ph in range(2):
            for pw in range(2):
                We = P.effective_weights(wt, P.class_matrix(pd, cd), P.class_matrix(ph, P.INT), P.class_matrix(pw, P.INT))
                for i_h in range(h):
                    for i_w in range(w):
                        patch = xp[:, :, i_d:i_d + 3, i_h:i_h + 3, i_w:i_w + 3]
                        exp[:, 2 * i_d + pd, 2 * i_h + ph, 2 * i_w + pw] = np.einsum("oiabc,niabc->no", We, patch)
print("main only vs expectation: max abs err", np.abs(main_only - exp).max())
corr = full - main_only
print("corrections (full - main) vs (ref - exp): max abs err", np.abs(corr - (ref - exp)).max(), "max corr", np.abs(ref - exp).max())
if max(shape) <= 2:
    np.set_printoptions(precision=4, suppress=True, linewidth=200)
    print("got  ", full.reshape(-1, 16)[:8])
    print("ref  ", ref.reshape(-1, 16)[:8])
    print("main ", main_only.reshape(-1, 16)[:8])
    print("exp  ", exp.reshape(-1, 16)[:8])
