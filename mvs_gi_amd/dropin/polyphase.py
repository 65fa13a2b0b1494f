"""Polyphase form of ResizeConv3d (dsta_mvs/model/common/common_modules.py:332-355): trilinear x2 upsample
(F.interpolate, align_corners=False, :335-341) followed by a 3x3x3 convolution with zero padding (:97-101).

Along one axis the upsampled sample at u = 2 i + p (phase p of low-resolution cell i) is
    p = 0:  0.25 x[i-1] + 0.75 x[i]        p = 1:  0.75 x[i] + 0.25 x[i+1]        (source index clamped to [0, n-1]: ATen's rule)
and the convolution's tap k reads up[u + k - 1], ZERO outside [0, 2n) (the conv pads the UPSAMPLED grid).  Composing the
two, every output phase is a 3-tap filter over the low-resolution neighbours x[i-1], x[i], x[i+1]:
    out[2 i + p] = sum_t ( sum_k M[p, class(i)][t][k] w[k] ) x[i + t - 1]
with a 3x3 matrix M that depends on the phase and on whether the cell is the first, an interior, the last (or the only) cell
of the axis: the clamp and the zero padding act on in-range samples there.  Out-of-range x (the zero border of the
split-padded activations, csrc/conv3d_rs.hip) multiplies whatever coefficient it gets, so only the centre row t = 1 of M
differs between the classes.  In 3-D the 8 output phases of a cell are 8 ordinary 3x3x3 convolutions over the
LOW-resolution tensor, W_eff = (M_d x M_h x M_w) w: the x2 upsample never materialises and no blend is evaluated at run time.

Host logic only (weight lowering, float64 -> fp32); the convolutions run in csrc/conv3d_rs.hip / csrc/conv3d_up2.hip.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

FIRST, INT, LAST, ONLY = 0, 1, 2, 3          # position class of a low-resolution cell along one axis
CLASS_NAMES = ("first", "int", "last", "only")


def cell_class(i: int, n: int) -> int:
    if n == 1:
        return ONLY
    return FIRST if i == 0 else (LAST if i == n - 1 else INT)


def class_cell(cls: int, n: int) -> int:
    """A representative cell index of a class on an axis of n cells (n >= 3 for INT)."""
    return {FIRST: 0, INT: 1, LAST: n - 1, ONLY: 0}[cls]


def axis_matrix(p: int, i: int, n: int) -> np.ndarray:
    """M[t][k]: coefficient of x[i + t - 1] in up[2 i + p + k - 1], from first principles (clamp + zero padding)."""
    M = np.zeros((3, 3))
    for k in range(3):
        v = 2 * i + p + k - 1                  # upsampled index read by tap k
        if v < 0 or v >= 2 * n:
            continue                           # the conv's zero padding
        m, q = divmod(v, 2)
        taps = ((m - 1, 0.25), (m, 0.75)) if q == 0 else ((m, 0.75), (m + 1, 0.25))
        for src, wgt in taps:
            src = min(max(src, 0), n - 1)      # ATen clamps the source index
            M[src - i + 1][k] += wgt
    return M


def class_matrix(p: int, cls: int) -> np.ndarray:
    """axis_matrix of a representative cell of the class (the matrix depends on (p, class) only)."""
    n = {FIRST: 4, INT: 4, LAST: 4, ONLY: 1}[cls]
    return axis_matrix(p, class_cell(cls, n), n)


def effective_weights(w: np.ndarray, Md: np.ndarray, Mh: np.ndarray, Mw: np.ndarray) -> np.ndarray:
    """w [Co, Ci, 3, 3, 3] (kd, kh, kw) -> W_eff [Co, Ci, 3, 3, 3] (td, th, tw) = (Md x Mh x Mw) w, float64."""
    return np.einsum("ak,bl,cm,oiklm->oiabc", Md, Mh, Mw, np.asarray(w, np.float64), optimize=True)


def main_weight_sets(w: np.ndarray) -> Dict[Tuple[int, int, int], np.ndarray]:
    """The register-stationary main kernel's weight sets: key (pd, d-class, ph) -> [32 = pw * 16 + co, Ci, 3, 3, 3] fp32 with
    INTERIOR matrices along H and W (the H / W faces are corrected by the face kernels)."""
    co = w.shape[0]
    out = {}
    for pd in range(2):
        for cd in (FIRST, INT, LAST, ONLY):
            for ph in range(2):
                parts = [effective_weights(w, class_matrix(pd, cd), class_matrix(ph, INT), class_matrix(pw, INT)) for pw in range(2)]
                out[(pd, cd, ph)] = np.concatenate(parts, axis=0).astype(np.float32).reshape(2 * co, w.shape[1], 3, 3, 3)
    return out


def face_delta(p: int, cls: int) -> np.ndarray:
    """M[p, class] - M[p, interior]: non-zero in the centre row only (the row that multiplies the in-range sample x[i])."""
    d = class_matrix(p, cls) - class_matrix(p, INT)
    if cls != INT:
        # rows t = 0 / t = 2 may differ too, but only where they multiply out-of-range samples (zeros): drop them
        if cls in (FIRST, ONLY):
            d[0] = 0.0
        if cls in (LAST, ONLY):
            d[2] = 0.0
    return d
