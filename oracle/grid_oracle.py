"""CPU restatement (torch fp32) of the reference's sampling-grid closed forms
(dsta_mvs/support/dataset/torch_cuda_sweep.py).  TEST INFRASTRUCTURE ONLY: imported by tests/ and
tools/, never by mvs_gi_amd.  Pinned against outputs of the reference file itself
(tools/make_grid_goldens.py -> tests/golden/sweep_grids.npz; mvs_utils helpers stubbed there:
debug printers = no-ops, torch_meshgrid = torch.meshgrid, FTensor = the plain tensor)."""
import math

import numpy as np
import torch


def rays_panorama(dist, long_range, lat_range, grid_shape):
    """RayMaker_UEPanorama.make_rays_for_candidates, torch_cuda_sweep.py:76-132 -> [3, N, H, W]."""
    H, W = grid_shape
    dist = torch.as_tensor(dist, dtype=torch.float32)
    phi = ((torch.arange(0, H) + 0.5) / H * (lat_range[1] - lat_range[0])) + lat_range[0]           # :91-92
    theta = ((torch.arange(0, W) + 0.5) / W * (long_range[1] - long_range[0])) + long_range[0]      # :99-100
    gd, gp, gt = torch.meshgrid(dist, phi, theta, indexing="ij")                                    # :113-114
    ds = gd * torch.sin(gp)                                                                         # :125
    x = ds * torch.cos(gt)
    y = -gd * torch.cos(gp)
    z = -ds * torch.sin(gt)
    return torch.stack((x, y, z), dim=0)                                                            # :131


def transform_points(T, points):
    """transform_3D_points_torch, :385-408."""
    B, _, N, H, W = points.shape
    p = points.reshape(B, 3, N * H * W)
    p = torch.matmul(T[:, :3, :3], p) + T[:, :3, 3].unsqueeze(2)
    return p.reshape(B, 3, N, H, W)


def double_sphere_w2(xi, alpha):
    w1 = alpha / (1 - alpha) if alpha <= 0.5 else (1 - alpha) / alpha                               # :251-254
    return (w1 + xi) / np.sqrt(2 * w1 * xi + xi ** 2 + 1)                                           # :256-257


def grid_double_sphere(points, params, calib_shape):
    """DoubleSphereSampleGridMaker.make_grid, :262-298 -> (grid [B,N,H,W,2], mask [B,N,H,W])."""
    xi, alpha, fx, fy, cx, cy = params
    x, y, z = torch.split(points, 1, dim=1)
    x2, y2, z2 = x ** 2, y ** 2, z ** 2
    d1 = torch.sqrt(x2 + y2 + z2)
    d2 = torch.sqrt(x2 + y2 + (xi * d1 + z) ** 2)
    t = alpha * d2 + (1 - alpha) * (xi * d1 + z)
    ux = (fx / t * x + cx) / (calib_shape[1] - 1) * 2 - 1
    uy = (fy / t * y + cy) / (calib_shape[0] - 1) * 2 - 1
    B, _, N, H, W = points.shape
    mask = (z > -double_sphere_w2(xi, alpha) * d1).view(B, N, H, W)
    return torch.cat((ux.view(B, N, H, W, 1), uy.view(B, N, H, W, 1)), dim=4), mask


def grid_equirect(points):
    """EquirectangularSampleGridMaker.make_grid, :305-335."""
    x, y, z = torch.split(points, 1, dim=1)
    xz = torch.sqrt(x ** 2 + z ** 2)
    lon = -1 * torch.atan2(z, x)
    lat = torch.atan2(y, xz)
    B, _, N, H, W = points.shape
    return torch.cat(((lon / np.pi).view(B, N, H, W, 1), (2 * lat / np.pi).view(B, N, H, W, 1)), dim=4)


def sweep_grid(maker, rays, pose):
    """make_sweep_grid_cuda, multi_view_camera_model_dataset.py:474-521 (pose error = none)."""
    inv = torch.linalg.inv(pose.to(torch.float64)).to(torch.float32)
    pts = transform_points(inv.unsqueeze(0), rays.unsqueeze(0))
    return maker(pts)


def ring_poses(n_cams: int, radius: float = 0.1):
    """Synthetic rig for tests / benches: cameras on a ring in the cv frame, each yawed towards its spoke."""
    poses = []
    for i in range(n_cams):
        a = 2 * math.pi * i / n_cams
        c, s = math.cos(0.3 * a), math.sin(0.3 * a)
        T = np.eye(4)
        T[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
        T[:3, 3] = [radius * math.cos(a), 0.02 * i, radius * math.sin(a)]
        poses.append(torch.from_numpy(T))
    return poses
