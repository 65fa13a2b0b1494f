"""Sampling-grid generator: the reference's closed forms
(dsta_mvs/support/dataset/torch_cuda_sweep.py) behind the same class / function names, running as
HIP kernels (mvsgi_rays_panorama_f32, mvsgi_transform_points_f32, mvsgi_grid_double_sphere_f32,
mvsgi_grid_equirect_f32).  Differences from the reference objects: plain torch tensors instead of
mvs_utils.FTensor (frame names are not tracked), fp32 candidate distances, tensors live on the GPU.

`make_sweep_grids` composes them the way MultiViewCameraModelDataset.create_grids_from_rays does
(support/dataset/multi_view_camera_model_dataset.py:474-557) and returns tensors in the layout the
cv_builder takes: grids [1, N, D, Ho, Wo, 2], grid_masks [1, N, D, Ho, Wo, 1] bool.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch

from .. import _lib
from ..hip_ops import _dev, _stream_ptr


def _as_f32_cuda(t, device) -> torch.Tensor:
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t))
    return t.to(device=device, dtype=torch.float32).contiguous()


class RayMaker_UEPanorama:
    """torch_cuda_sweep.py:13-132.  distance: 1-D array of candidate distances; ranges in radians."""

    def __init__(self, distance, long_range, lat_range, frame_name: str = "rbf", device="cuda"):
        self.device = torch.device(device)
        self.dist = _as_f32_cuda(distance, self.device)
        self.long_range = (float(long_range[0]), float(long_range[1]))
        self.lat_range = (float(lat_range[0]), float(lat_range[1]))
        self.frame_name = frame_name

    def make_rays_for_candidates(self, grid_shape: Sequence[int]) -> torch.Tensor:
        """-> rays [3, N, H, W] in the panorama frame (z backward, x left, y down); :76-132."""
        lib = _lib.load()
        H, W = int(grid_shape[0]), int(grid_shape[1])
        N = self.dist.numel()
        rays = torch.empty((3, N, H, W), device=self.device, dtype=torch.float32)
        _lib.check(lib.mvsgi_rays_panorama_f32(self.dist.data_ptr(), rays.data_ptr(), N, H, W, self.lat_range[0],
                                               self.lat_range[1], self.long_range[0], self.long_range[1],
                                               _stream_ptr(rays)), "mvsgi_rays_panorama_f32")
        return rays


def transform_3D_points_torch(T: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """T [B, 4, 4], points [B, 3, N, H, W] -> R p + t, same shape (torch_cuda_sweep.py:385-408)."""
    lib = _lib.load()
    points = _dev(points, "points")
    T = _dev(T.to(torch.float32), "T")
    B = points.shape[0]
    if points.dim() != 5 or points.shape[1] != 3 or tuple(T.shape) != (B, 4, 4):
        raise AssertionError(f"expected T [B,4,4] and points [B,3,N,H,W], got {tuple(T.shape)}, {tuple(points.shape)}")
    out = torch.empty_like(points)
    M = points[0, 0].numel()
    _lib.check(lib.mvsgi_transform_points_f32(T.data_ptr(), points.data_ptr(), out.data_ptr(), B, M, _stream_ptr(points)),
               "mvsgi_transform_points_f32")
    return out


class DoubleSphereSampleGridMaker:
    """torch_cuda_sweep.py:235-298: double-sphere projection of 3-D points to grid_sample coordinates."""

    def __init__(self, params=(-0.203, 0.589, 232.0, 232.0, 611.5, 513.5), calib_shape=(1028, 1224)):
        self.xi, self.alpha, self.fx, self.fy, self.cx, self.cy = (float(p) for p in params)
        self.calib_shape = [int(calib_shape[0]), int(calib_shape[1])]
        self.w1 = self.alpha / (1 - self.alpha) if self.alpha <= 0.5 else (1 - self.alpha) / self.alpha   # :251-254
        self.w2 = (self.w1 + self.xi) / math.sqrt(2 * self.w1 * self.xi + self.xi ** 2 + 1)              # :256-257

    def make_grid(self, points: torch.Tensor):
        """points [B, 3, N, H, W] -> (grid [B, N, H, W, 2] in [-1, 1], mask [B, N, H, W] bool); :262-298."""
        lib = _lib.load()
        points = _dev(points, "points")
        B, _, N, H, W = points.shape
        grid = torch.empty((B, N, H, W, 2), device=points.device, dtype=torch.float32)
        mask = torch.empty((B, N, H, W), device=points.device, dtype=torch.uint8)
        _lib.check(lib.mvsgi_grid_double_sphere_f32(points.data_ptr(), grid.data_ptr(), mask.data_ptr(), B, N * H * W,
                                                    self.xi, self.alpha, self.fx, self.fy, self.cx, self.cy,
                                                    self.calib_shape[0], self.calib_shape[1], self.w2,
                                                    _stream_ptr(points)), "mvsgi_grid_double_sphere_f32")
        return grid, mask.bool()


class EquirectangularSampleGridMaker:
    """torch_cuda_sweep.py:300-335: longitude / latitude of 3-D points as grid_sample coordinates."""

    def make_grid(self, points: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        points = _dev(points, "points")
        B, _, N, H, W = points.shape
        grid = torch.empty((B, N, H, W, 2), device=points.device, dtype=torch.float32)
        _lib.check(lib.mvsgi_grid_equirect_f32(points.data_ptr(), grid.data_ptr(), B, N * H * W, _stream_ptr(points)),
                   "mvsgi_grid_equirect_f32")
        return grid


CAMERA_MODEL_GRID_MAKER_MAP = {                      # torch_cuda_sweep.py:379-383 (CameraModel needs mvs_utils)
    "DoubleSphere": DoubleSphereSampleGridMaker,
    "Equirectangular": EquirectangularSampleGridMaker,
}


def make_sweep_grid(grid_maker, rays: torch.Tensor, pose: torch.Tensor):
    """make_sweep_grid_cuda (multi_view_camera_model_dataset.py:474-521): rays [3, N, H, W] of the cv camera,
    pose [4, 4] of the camera in the cv frame -> (grid [N, H, W, 2], valid_mask [N, H, W] bool or None)."""
    inv_pose = torch.linalg.inv(pose.to(torch.float64)).to(torch.float32)           # :505 (tiny 4x4, host-side algebra)
    pts = transform_3D_points_torch(inv_pose.unsqueeze(0).to(rays.device), rays.unsqueeze(0))
    out = grid_maker.make_grid(pts)
    if isinstance(out, tuple):
        return out[0].squeeze(0), out[1].squeeze(0)
    return out.squeeze(0), None


def make_sweep_grids(ray_maker: RayMaker_UEPanorama, grid_makers, poses, cv_shape):
    """create_grids_from_rays (:531-557) for a rig -> grids [1, N, D, Ho, Wo, 2], grid_masks [1, N, D, Ho, Wo, 1]."""
    rays = ray_maker.make_rays_for_candidates(cv_shape)
    grids, masks = [], []
    for gm, pose in zip(grid_makers, poses):
        g, m = make_sweep_grid(gm, rays, torch.as_tensor(pose))
        grids.append(g)
        masks.append(m if m is not None else torch.ones(g.shape[:-1], dtype=torch.bool, device=g.device))
    return torch.stack(grids).unsqueeze(0), torch.stack(masks).unsqueeze(0).unsqueeze(-1)
