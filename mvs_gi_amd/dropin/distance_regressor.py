"""Drop-in dist_regressor: DistanceRegressorWithFixedCandidates
(dsta_mvs/model/distance_regressor/distance_regressor.py:7-79) with the same constructor,
buffer name (`inv_dist_idx` [1, D, 1, 1], persistent), attributes (bf, inv_dist_idx_min/max,
interp_scale_factor, pre_interp) and `update_dist_cands`; forward is the fused HIP
upsample + softmax + expectation kernel (mvsgi_softargmin_f32).
"""
from __future__ import annotations

from typing import Sequence

import torch
from torch import nn, Tensor

from .. import hip_ops as H


def regressor_forward(self, costs: Tensor):
    """costs [B, 1(+), D, H, W] -> (inv_dist [B,1,sH,sW], norm_costs [B,D,sH,sW])."""
    c = costs[:, 0]
    scale = 1
    if self.pre_interp and self.interp_scale_factor > 0:
        if float(self.interp_scale_factor) not in (1.0, 2.0):
            raise NotImplementedError(
                f"interp_scale_factor={self.interp_scale_factor}: the HIP soft-argmin fuses x1 and x2 only")
        scale = int(self.interp_scale_factor)
    want = getattr(self, "return_norm_costs", True)
    return H.softargmin(c, self.inv_dist_idx, scale, want, getattr(self, "post_div", 1.0))


class DistanceRegressorWithFixedCandidates(nn.Module):
    def __init__(self, bf: float = 96, dist_cands: Sequence[float] = [0.5, 1, 1.5, 2, 5, 10, 20, 30, 50, 100],
                 interp_scale_factor: float = 0, pre_interp: bool = False):
        super().__init__()
        inv = bf / torch.tensor(list(dist_cands), dtype=torch.float32)
        self.register_buffer("inv_dist_idx", inv.view(1, -1, 1, 1), persistent=True)
        self.inv_dist_idx_min = float(torch.min(self.inv_dist_idx))
        self.inv_dist_idx_max = float(torch.max(self.inv_dist_idx))
        self.bf = bf
        self.interp_scale_factor = interp_scale_factor if interp_scale_factor > 0 else 0
        self.pre_interp = pre_interp
        # inference callers discard norm_costs (spherical_sweep_stereo.py:266); set False to skip its store
        self.return_norm_costs = True

    def update_dist_cands(self, dist_cands: Sequence[float]):
        inv = self.bf / torch.tensor(list(dist_cands), dtype=torch.float32)
        self.inv_dist_idx[:] = inv.view(1, -1, 1, 1).to(dtype=self.inv_dist_idx.dtype,
                                                        device=self.inv_dist_idx.device)
        self.inv_dist_idx_min = float(torch.min(self.inv_dist_idx))
        self.inv_dist_idx_max = float(torch.max(self.inv_dist_idx))

    forward = regressor_forward
