"""CPU: the algebra behind csrc/conv3d_wino.hip, in float64 numpy -- no GPU, no oracle import needed.
(1) F(2x2, 3x3) over (H, W) x direct over D with the kernel's matrices and its plane-march bookkeeping equals the convolution;
(2) the per-cout power of two of the weight transform; (3) the claim DESIGN section 11 makes about the polyphase layer's first /
last output planes: their folded weights are linear combinations of the interior planes' (what would let K3 run in this form on
resident weights)."""
import numpy as np
import torch
import torch.nn.functional as F

from mvs_gi_amd.dropin.polyphase import FIRST, INT, LAST, class_matrix

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def test_winograd_plane_march_equals_the_convolution():
    """y = A^T [ sum_kd U_kd (.) V_(d + kd - 1) ] A with U = G g G^T, V = B^T x B, the depth taps accumulated plane by plane as the
    kernel does (input plane p adds U_kd V_p to output plane p + 1 - kd; taps on the zero border skipped)."""
    rng = np.random.default_rng(5)
    Ci, Co, D, H, W = 5, 4, 6, 8, 12
    x = rng.standard_normal((1, Ci, D, H, W))
    w = rng.standard_normal((Co, Ci, 3, 3, 3))
    ref = F.conv3d(torch.from_numpy(x), torch.from_numpy(w), padding=1).numpy()[0]
    U = np.einsum("ah,bw,oidhw->abdoi", G, G, w)                        # [a, b, kd, co, ci]
    xp = np.pad(x[0], ((0, 0), (0, 0), (1, 1), (1, 1)))                # in-plane zero border (the padded tensor's)
    out = np.zeros((Co, D, H, W))
    Y = np.zeros((D, H // 2, W // 2, 4, 4, Co))                        # per output plane and tile, the 16 transform points
    for p in range(D):                                                 # the march over real input planes
        V = np.zeros((H // 2, W // 2, 4, 4, Ci))
        for r in range(H // 2):
            for c in range(W // 2):
                patch = xp[:, p, 2 * r:2 * r + 4, 2 * c:2 * c + 4]     # [ci, 4, 4]
                V[r, c] = np.einsum("ai,bj,kij->abk", BT, BT, patch)
        for kd in range(3):
            o = p + 1 - kd
            if 0 <= o < D:
                Y[o] += np.einsum("aboi,rcabi->rcabo", U[:, :, kd], V)
    for o in range(D):
        t = np.einsum("pa,qb,rcabo->orpcq", AT, AT, Y[o])              # [co, r, pa, c, q]
        out[:, o] = t.reshape(Co, H, W)
    assert np.abs(out - ref).max() <= 1e-12 * np.abs(ref).max() + 1e-12


def test_transformed_weights_power_of_two():
    """hip_ops._pow2_unscale on max |U| per cout: U * 2^k in (512, 1024], k an integer, a zero channel keeps k = 0 (what the HIP
    pack kernel computes with frexp; checked against it on the GPU by test_conv3d_winograd_weights_range_and_misuse)."""
    from mvs_gi_amd.hip_ops import _pow2_unscale
    rng = np.random.default_rng(6)
    w = rng.standard_normal((8, 4, 3, 3, 3)) * np.exp(rng.uniform(-8, 8, (8, 1, 1, 1, 1)))
    w[3] = 0.0
    U = np.einsum("ah,bw,oidhw->abdoi", G, G, w)
    amax = torch.from_numpy(np.abs(U).max(axis=(0, 1, 2, 4)).astype(np.float32))
    up, un = _pow2_unscale(amax)
    k = np.log2(up.numpy().astype(np.float64))
    assert (k == np.round(k)).all() and k[3] == 0 and np.allclose(up.numpy() * un.numpy(), 1.0)
    live = amax.numpy() > 0
    scaled = amax.numpy()[live].astype(np.float64) * up.numpy()[live]
    assert ((scaled > 512) & (scaled <= 1024)).all()


def test_polyphase_boundary_planes_are_combinations_of_the_interior_weights():
    """along D, the folded weights of the first (last) low-resolution plane are linear combinations of the
    interior plane's three depth taps -- phase 1 at the top and phase 0 at the bottom by replication (K1 + K0, K1 + K2), phase 0 at
    the top and phase 1 at the bottom with (-0.5, 1.5, -1.5) and (-1.5, 1.5, -0.5) -- on the in-range samples (the taps that
    multiply the zero border do not matter)."""
    for pd, cls, coef in ((1, FIRST, (1.0, 1.0, 0.0)), (0, FIRST, (-0.5, 1.5, -1.5)), (0, LAST, (0.0, 1.0, 1.0)), (1, LAST, (-1.5, 1.5, -0.5))):
        Mi, Mb = class_matrix(pd, INT), class_matrix(pd, cls)          # M[t][k]: coefficient of x[i + t - 1] under conv tap k
        centre = sum(c * Mi[t] for t, c in enumerate(coef))            # the combination of the interior rows t = 0, 1, 2
        assert np.allclose(Mb[1], centre), (pd, cls)                   # = the boundary cell's weight on its own sample x[i]
        keep = 2 if cls == FIRST else 0                                # the other in-range neighbour keeps its interior weight
        assert np.allclose(Mb[keep], Mi[keep]), (pd, cls)
