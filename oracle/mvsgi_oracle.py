"""CPU oracle for the mvs_gi plane-sweep hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this module; the product package `mvs_gi_amd` never does and has no
CPU fallback (its ops raise when the HIP library is missing).

This is a from-scratch fp32 restatement, on torch-CPU / ATen primitives (the same
arithmetic library the reference itself runs on), of

  * the bilinear zero-padded sampler   dsta_mvs/model/backports/backports.py:34-86
  * the masked-variance sweep          cost_volume_builder/spherical_sweep_avg.py:38-136
  * the concat sweep                   cost_volume_builder/spherical_sweep.py:38-68
  * conv block = Conv3d -> eval BN -> (+res) -> LeakyReLU
                                       common/common_modules.py:82-115
  * residual block                     common/common_modules.py:231-244
  * resize-conv (trilinear x2, optional re-interp to the skip's size)
                                       common/common_modules.py:332-355
  * the UNet regulator forward         cost_volume_regulator/unet_regulator.py:120-140
  * the fixed-candidate soft-argmin    distance_regressor/distance_regressor.py:51-79

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference's own
modules, imported in the build container by `tools/make_goldens.py`, and stored
as fixtures in `tests/golden/` (checked by tests/test_oracle_golden.py).
Parameters are plain dicts of tensors keyed by the reference's state-dict names.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5          # nn.BatchNorm3d default
LRELU_SLOPE = 0.01     # nn.LeakyReLU default (common/__init__.py:7-11)


# ----------------------------------------------------------------------------
# sampler
# ----------------------------------------------------------------------------
def bilinear_sample_zeros(im: Tensor, grid: Tensor) -> Tensor:
    """im [n,c,h,w], grid [n,gh,gw,2] in normalised coords, align_corners=False,
    zero padding; weights come from the UNCLAMPED coordinates (backports.py:41-55)
    and any tap outside the image contributes 0 (backports.py:58-72)."""
    n, c, h, w = im.shape
    gn, gh, gw, _ = grid.shape
    if n != gn:
        raise AssertionError("batch mismatch between image and grid")
    x = ((grid[..., 0] + 1) * w - 1) / 2
    y = ((grid[..., 1] + 1) * h - 1) / 2
    x = x.reshape(n, -1)
    y = y.reshape(n, -1)
    xf = torch.floor(x)
    yf = torch.floor(y)
    x0 = xf.long()
    y0 = yf.long()
    x1 = x0 + 1
    y1 = y0 + 1
    # weight of tap (x0,y0), (x0,y1), (x1,y0), (x1,y1)
    w00 = (x1 - x) * (y1 - y)
    w01 = (x1 - x) * (y - y0)
    w10 = (x - x0) * (y1 - y)
    w11 = (x - x0) * (y - y0)
    flat = im.reshape(n, c, h * w)

    def tap(xi, yi):
        inside = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
        idx = (xi.clamp(0, w - 1) + yi.clamp(0, h - 1) * w)
        v = torch.gather(flat, 2, idx.unsqueeze(1).expand(-1, c, -1))
        return v * inside.unsqueeze(1).to(v.dtype)

    out = tap(x0, y0) * w00.unsqueeze(1) + tap(x0, y1) * w01.unsqueeze(1) \
        + tap(x1, y0) * w10.unsqueeze(1) + tap(x1, y1) * w11.unsqueeze(1)
    return out.reshape(n, c, gh, gw)


# ----------------------------------------------------------------------------
# sweeps
# ----------------------------------------------------------------------------
def sweep_std_masked(feats: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor) -> Tensor:
    """feats [B,N,C,Hi,Wi], grids [B,N,D,Ho,Wo,2], grid_masks [B,N,D,Ho,Wo,1] (bool or
    float), masks [B,N,1,Hm,Wm] -> vol_raw [B,C,D,Ho,Wo]
    (spherical_sweep_avg.py:38-136)."""
    B, N, C = feats.shape[:3]
    D, Ho, Wo = grids.shape[2:5]
    f = feats.flatten(0, 1)
    m = masks.flatten(0, 1)
    g = grids.flatten(0, 1)
    planes = []
    for d in range(D):
        gd = g[:, d]
        sf = bilinear_sample_zeros(f, gd).reshape(B, N, C, Ho, Wo)
        sm = (bilinear_sample_zeros(m, gd) > 0.0).reshape(B, N, 1, Ho, Wo)
        gm = grid_masks[:, :, d, :, :, 0].reshape(B, N, 1, Ho, Wo)
        valid = torch.logical_and(sm, gm)                         # :102
        vf = valid.to(torch.float32)
        n = vf.sum(dim=1, keepdim=True)                           # :106
        ok = n > 1.0                                              # :108
        cnt = torch.where(ok, n, torch.ones_like(n))              # :111
        mean = (sf * vf).sum(dim=1, keepdim=True) / cnt           # :114
        sf2 = torch.where(valid, sf, mean)                        # :119
        var = ((sf2 - mean) ** 2).sum(dim=1, keepdim=True) / cnt  # :122
        var = torch.where(ok.expand_as(var), var, torch.zeros_like(var))
        planes.append(var[:, 0].unsqueeze(2))
    return torch.cat(planes, dim=2)


def sweep_concat(feats: Tensor, grids: Tensor) -> Tensor:
    """vol_raw[b, cam*C + c, d] = sample(feats[b, cam, c]) (spherical_sweep.py:52-67)."""
    B, N, C = feats.shape[:3]
    D, Ho, Wo = grids.shape[2:5]
    f = feats.flatten(0, 1)
    g = grids.flatten(0, 1)
    planes = []
    for d in range(D):
        sf = bilinear_sample_zeros(f, g[:, d]).reshape(B, N * C, Ho, Wo)
        planes.append(sf.unsqueeze(2))
    return torch.cat(planes, dim=2)


# ----------------------------------------------------------------------------
# conv blocks
# ----------------------------------------------------------------------------
def conv_block(x: Tensor, p: Dict[str, Tensor], prefix: str, stride: int = 1,
               res: Optional[Tensor] = None, act: bool = True) -> Tensor:
    """BaseConvBlk3d.forward (common_modules.py:107-115): conv (pad k//2) -> eval BN ->
    (+res) -> LeakyReLU.  Blocks without norm/activation (out_costs.1) pass act=False."""
    w = p[f"{prefix}.conv_layer.weight"]
    b = p.get(f"{prefix}.conv_layer.bias")
    y = F.conv3d(x, w, b, stride=stride, padding=w.shape[-1] // 2)
    if f"{prefix}.norm_layer.weight" in p:
        y = F.batch_norm(y, p[f"{prefix}.norm_layer.running_mean"], p[f"{prefix}.norm_layer.running_var"],
                         p[f"{prefix}.norm_layer.weight"], p[f"{prefix}.norm_layer.bias"],
                         training=False, eps=BN_EPS)
    if res is not None:
        y = y + res
    if act:
        y = F.leaky_relu(y, LRELU_SLOPE)
    return y


def res_block(x: Tensor, p: Dict[str, Tensor], prefix: str) -> Tensor:
    """ResConvBlk3d.forward with in_chs == out_chs (common_modules.py:231-244)."""
    r = conv_block(x, p, f"{prefix}.blk1")
    return conv_block(r, p, f"{prefix}.blk2", res=x)


def resize_conv(x: Tensor, p: Dict[str, Tensor], prefix: str, res: Optional[Tensor] = None) -> Tensor:
    """ResizeConv3d.forward (common_modules.py:332-355)."""
    up = [int(2 * s) for s in x.shape[2:]]
    x = F.interpolate(x, size=up, mode="trilinear", align_corners=False)
    if res is not None and x.shape != res.shape:
        x = F.interpolate(x, size=res.shape[2:], mode="trilinear")
    return conv_block(x, p, f"{prefix}.conv", res=res)


def post_vol(vol_raw: Tensor, p: Dict[str, Tensor]) -> Tensor:
    return conv_block(vol_raw, p, "post_vol")


def regulator_forward(vol: Tensor, p: Dict[str, Tensor], u_depth: int = 3, blk_width: int = 4,
                      return_intermediates: bool = False):
    """UNetCostVolumeRegulatorBase.forward (unet_regulator.py:120-140)."""
    x = vol
    skips = []
    inter = {}
    for i in range(u_depth):
        x = conv_block(x, p, f"down_blks.{i}.first", stride=2)
        for j in range(blk_width - 1):
            x = res_block(x, p, f"down_blks.{i}.blks.{j}")
        inter[f"down{i}"] = x
        if i != u_depth - 1:
            skips.append(x)
    skips.reverse()
    for i in range(u_depth - 1):
        x = resize_conv(x, p, f"upBlks.{i}", res=skips[i])
        inter[f"up{i}"] = x
    x = resize_conv(x, p, "out_costs.0")
    inter["out0"] = x
    x = conv_block(x, p, "out_costs.1", act=False)
    return (x, inter) if return_intermediates else x


# ----------------------------------------------------------------------------
# feature extractor
# ----------------------------------------------------------------------------
def conv_block2d(x: Tensor, p: Dict[str, Tensor], prefix: str, stride: int = 1,
                 res: Optional[Tensor] = None) -> Tensor:
    """BaseConvBlk2d.forward (common_modules.py:56-70): conv (pad k//2) -> eval BN -> (+res) -> LeakyReLU."""
    w = p[f"{prefix}.conv_layer.weight"]
    y = F.conv2d(x, w, p.get(f"{prefix}.conv_layer.bias"), stride=stride, padding=w.shape[-1] // 2)
    y = F.batch_norm(y, p[f"{prefix}.norm_layer.running_mean"], p[f"{prefix}.norm_layer.running_var"],
                     p[f"{prefix}.norm_layer.weight"], p[f"{prefix}.norm_layer.bias"], training=False, eps=BN_EPS)
    if res is not None:
        y = y + res
    return F.leaky_relu(y, LRELU_SLOPE)


def feature_extractor(imgs: Tensor, p: Dict[str, Tensor], layers: Sequence[int] = (5, 10)) -> Tensor:
    """SimpleFeatExtraction.forward (feature_extractor/simple_feature_extractor.py:81-84):
    5x5 stride-2 stem, `layers[i]` residual blocks per stage (ResConvBlk2d.forward,
    common_modules.py:165-176), a 3x3 stride-2 conv between stages, final 3x3 conv.
    imgs [M, 3, H, W] -> [M, chs, H/4, W/4]."""
    x = conv_block2d(imgs, p, "first", stride=2)
    i = 0
    for step, n in enumerate(layers):
        for _ in range(n):
            r = conv_block2d(x, p, f"blks.{i}.blk1")
            x = conv_block2d(r, p, f"blks.{i}.blk2", res=x)
            i += 1
        if step != len(layers) - 1:
            x = conv_block2d(x, p, f"blks.{i}", stride=2)
            i += 1
    return conv_block2d(x, p, "final_layer")


# ----------------------------------------------------------------------------
# sphere-convolution final layer.  PARITY UNPINNED for deform_conv2d itself:
# torchvision (the reference's dependency for this operator) is neither vendored in /root/reference nor
# installed in the build container, so the operator is restated from its published definition (Dai et al.,
# "Deformable Convolutional Networks", 2017; torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp) and anchored
# by the identities the tests check (zero offsets == F.conv2d, integer offsets == shifted taps).  The offset
# field IS pinned: the reference's own gen_offset output is a golden (tests/golden/sphere_offsets.npz).
# ----------------------------------------------------------------------------
def deform_conv2d(x: Tensor, offset: Tensor, w: Tensor, bias: Optional[Tensor] = None, stride=(1, 1),
                  padding=(0, 0), dilation=(1, 1)) -> Tensor:
    """torchvision.ops.deform_conv2d(input, offset, weight, bias, stride, padding, dilation, mask=None), as
    SphereConvEquirect2d.forward calls it (common_modules.py:411-425).  x [N, Cin, H, W], offset [N, 2*Kh*Kw,
    Ho, Wo] (channel 2k = dy, 2k+1 = dx of tap k = i*Kw + j), w [Cout, Cin, Kh, Kw]."""
    N, Cin, H, W = x.shape
    Cout, _, Kh, Kw = w.shape
    Ho = (H + 2 * padding[0] - (dilation[0] * (Kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (Kw - 1) + 1)) // stride[1] + 1
    offset = offset.expand(N, -1, -1, -1)
    ho = torch.arange(Ho, dtype=torch.float32).view(1, Ho, 1)
    wo = torch.arange(Wo, dtype=torch.float32).view(1, 1, Wo)
    xf = x.reshape(N, Cin, H * W)
    cols = []
    for i in range(Kh):
        for j in range(Kw):
            k = i * Kw + j
            y = ho * stride[0] - padding[0] + i * dilation[0] + offset[:, 2 * k]            # [N, Ho, Wo]
            xx = wo * stride[1] - padding[1] + j * dilation[1] + offset[:, 2 * k + 1]
            inside = ~((y <= -1) | (y >= H) | (xx <= -1) | (xx >= W))
            yl, xl = torch.floor(y), torch.floor(xx)
            ly, lx = y - yl, xx - xl
            hy, hx = 1 - ly, 1 - lx
            yl, xl = yl.long(), xl.long()
            yh, xh = yl + 1, xl + 1

            def corner(yy, xc, ok):
                ok = ok & inside
                idx = (yy.clamp(0, H - 1) * W + xc.clamp(0, W - 1)).view(N, 1, Ho * Wo).expand(-1, Cin, -1)
                v = torch.gather(xf, 2, idx).view(N, Cin, Ho, Wo)
                return torch.where(ok.unsqueeze(1), v, torch.zeros((), dtype=x.dtype))
            v1 = corner(yl, xl, (yl >= 0) & (xl >= 0))
            v2 = corner(yl, xh, (yl >= 0) & (xh <= W - 1))
            v3 = corner(yh, xl, (yh <= H - 1) & (xl >= 0))
            v4 = corner(yh, xh, (yh <= H - 1) & (xh <= W - 1))
            cols.append((hy * hx).unsqueeze(1) * v1 + (hy * lx).unsqueeze(1) * v2 + (ly * hx).unsqueeze(1) * v3 +
                        (ly * lx).unsqueeze(1) * v4)
    col = torch.stack(cols, dim=2)                                     # [N, Cin, K, Ho, Wo]
    out = torch.einsum("ock,nckhw->nohw", w.reshape(Cout, Cin, Kh * Kw), col)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def sphere_block2d(x: Tensor, p: Dict[str, Tensor], prefix: str, res: Optional[Tensor] = None) -> Tensor:
    """SphereConvBlk.forward (common_modules.py:538-547) with the extractor's settings (k 3, stride 1, padding 1,
    eval BatchNorm2d, LeakyReLU): state-dict keys {prefix}.blk.0.{weight,offset[,bias]}, {prefix}.blk.1.*."""
    w = p[f"{prefix}.blk.0.weight"]
    k = w.shape[-1]
    y = deform_conv2d(x, p[f"{prefix}.blk.0.offset"], w, p.get(f"{prefix}.blk.0.bias"), padding=(k // 2, k // 2))
    y = F.batch_norm(y, p[f"{prefix}.blk.1.running_mean"], p[f"{prefix}.blk.1.running_var"],
                     p[f"{prefix}.blk.1.weight"], p[f"{prefix}.blk.1.bias"], training=False, eps=BN_EPS)
    if res is not None:
        y = y + res
    return F.leaky_relu(y, LRELU_SLOPE)


def sphere_feature_extractor(imgs: Tensor, p: Dict[str, Tensor], layers: Sequence[int] = (5, 10)) -> Tensor:
    """SphereEquirectFeatExtraction.forward (feature_extractor/sphere_feature_extractor.py:80-83)."""
    x = conv_block2d(imgs, p, "first", stride=2)
    i = 0
    for step, n in enumerate(layers):
        for _ in range(n):
            r = conv_block2d(x, p, f"blks.{i}.blk1")
            x = conv_block2d(r, p, f"blks.{i}.blk2", res=x)
            i += 1
        if step != len(layers) - 1:
            x = conv_block2d(x, p, f"blks.{i}", stride=2)
            i += 1
    return sphere_block2d(x, p, "final_layer")


def full_model(imgs: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor, weights, builder: str,
               dist_cands: Sequence[float], **kw) -> Tensor:
    """SphericalSweepStereoBase.forward (mvs_model/torch_only.py:20-36): imgs [B, N, 3, H, W]."""
    with torch.no_grad():
        B, N = imgs.shape[:2]
        f = feature_extractor(imgs.reshape(B * N, *imgs.shape[2:]), weights["feature_extractor"])
        feats = f.reshape(B, N, *f.shape[1:])
    return hot_path(feats, grids, grid_masks, masks, weights, builder, dist_cands, **kw)


# ----------------------------------------------------------------------------
# regression
# ----------------------------------------------------------------------------
def soft_argmin(costs: Tensor, dist_cands: Sequence[float], bf: float = 96.0,
                interp_scale_factor: float = 2, pre_interp: bool = True) -> Tuple[Tensor, Tensor]:
    """DistanceRegressorWithFixedCandidates.forward (distance_regressor.py:51-79)."""
    inv_idx = (bf / torch.tensor(list(dist_cands), dtype=torch.float32)).view(1, -1, 1, 1)
    c = costs[:, 0]
    if pre_interp and interp_scale_factor > 0:
        c = F.interpolate(c, scale_factor=interp_scale_factor, mode="bilinear")
    pr = F.softmax(c, 1)
    inv = (pr * inv_idx.expand_as(pr)).sum(dim=1, keepdim=True)
    return inv, pr


# ----------------------------------------------------------------------------
# whole path
# ----------------------------------------------------------------------------
def hot_path(feats: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor,
             weights: Dict[str, Dict[str, Tensor]], builder: str, dist_cands: Sequence[float],
             bf: float = 96.0, interp_scale_factor: float = 2, pre_interp: bool = True,
             return_stages: bool = False):
    """feats, grids, grid_masks, masks -> inv_dist, the composition in
    mvs_model/torch_only.py:32-34."""
    with torch.no_grad():
        if builder == "std":
            vol_raw = sweep_std_masked(feats, grids, grid_masks, masks)
        elif builder == "cat":
            vol_raw = sweep_concat(feats, grids)
        else:
            raise ValueError(builder)
        vol = post_vol(vol_raw, weights["cv_builder"])
        costs = regulator_forward(vol, weights["cv_regulator"])
        inv, pr = soft_argmin(costs, dist_cands, bf, interp_scale_factor, pre_interp)
    if return_stages:
        return dict(vol_raw=vol_raw, vol=vol, costs=costs, inv_dist=inv, norm_costs=pr)
    return inv


def to_torch(d):
    """numpy dict (possibly nested) -> torch CPU tensors."""
    import numpy as np
    if isinstance(d, dict):
        return {k: to_torch(v) for k, v in d.items()}
    return torch.from_numpy(np.ascontiguousarray(d))
