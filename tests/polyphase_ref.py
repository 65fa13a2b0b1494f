"""Test helpers for the polyphase form of ResizeConv3d (mvs_gi_amd/dropin/polyphase.py states the algebra the product lowers
weights with): the layer by its definition in float64 PyTorch, and the same result assembled cell by cell from the 8 phase
convolutions.  Test infrastructure only -- nothing in the package imports this."""
import numpy as np

from mvs_gi_amd.dropin.polyphase import cell_class, class_matrix, effective_weights


def reference_up2_conv(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """conv3d(interpolate(x, x2, trilinear), w, padding=1) in float64 by its definition (tests)."""
    import torch
    import torch.nn.functional as F
    xt = torch.from_numpy(np.asarray(x, np.float64))
    up = F.interpolate(xt, scale_factor=2, mode="trilinear", align_corners=False)
    return F.conv3d(up, torch.from_numpy(np.asarray(w, np.float64)), padding=1).numpy()


def polyphase_up2_conv(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """The same result assembled from the 8 phase convolutions over the low-resolution tensor with per-cell class matrices
    (float64; the CPU statement of what main kernel + face corrections compute together)."""
    B, Ci, D, H, W = x.shape
    Co = w.shape[0]
    xp = np.pad(np.asarray(x, np.float64), ((0, 0), (0, 0), (1, 1), (1, 1), (1, 1)))
    out = np.zeros((B, Co, 2 * D, 2 * H, 2 * W))
    cache = {}
    for i_d in range(D):
        for i_h in range(H):
            for i_w in range(W):
                cd, ch, cw = cell_class(i_d, D), cell_class(i_h, H), cell_class(i_w, W)
                patch = xp[:, :, i_d:i_d + 3, i_h:i_h + 3, i_w:i_w + 3]
                for pd in range(2):
                    for ph in range(2):
                        for pw in range(2):
                            key = (pd, cd, ph, ch, pw, cw)
                            if key not in cache:
                                cache[key] = effective_weights(w, class_matrix(pd, cd), class_matrix(ph, ch), class_matrix(pw, cw))
                            out[:, :, 2 * i_d + pd, 2 * i_h + ph, 2 * i_w + pw] = np.einsum("oiabc,niabc->no", cache[key], patch)
    return out
