"""Deterministic synthetic inputs and weights for the plane-sweep path.

Everything here is seeded `numpy.random.default_rng`, so the golden-vector script
(run where the reference is importable), the tests and `bench.py` (run on the
GPU box, where it is not) all see bit-identical arrays.

The "smooth" grids are produced by a build-owned generator that follows the
closed forms of the reference's grid construction (unit rays of an equirect
cv camera times the candidate distances: dsta_mvs/support/dataset/
torch_cuda_sweep.py:80-127; rigid transform into each camera: :385-408;
equirect projection u = -atan2(z, x)/pi, v from atan2(y, |xz|): :305-335).
Sampling grids are an INPUT of the hot path, so this generator
only has to be realistic, not identical to the un-vendored camera models.
"""
from __future__ import annotations

import hashlib
from typing import Dict

import numpy as np

from .configs import PathConfig, regulator_conv_specs


# ----------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------
def smooth_grids(cfg: PathConfig, ring_radius: float = 0.1, fov_deg: float = 220.0):
    """Returns grids [N, D, Ho, Wo, 2] f32, grid_masks [N, D, Ho, Wo, 1] bool and
    image-resolution camera masks [N, 1, Hm, Wm] f32 for a ring of `num_cams`
    upper-half-sphere equirect cameras."""
    N, D = cfg.num_cams, cfg.num_cands
    Ho, Wo = cfg.cv_hw
    Hm, Wm = cfg.mask_hw
    lat = -np.pi / 2 + (np.arange(Ho, dtype=np.float64) + 0.5) / Ho * (np.pi / 2)
    lon = -np.pi + (np.arange(Wo, dtype=np.float64) + 0.5) / Wo * (2 * np.pi)
    lat, lon = np.meshgrid(lat, lon, indexing="ij")
    ray = np.stack([np.cos(lat) * np.cos(lon), np.sin(lat), -np.cos(lat) * np.sin(lon)], 0)
    dist = np.asarray(cfg.dist_cands, dtype=np.float64)
    pts = ray[:, None] * dist[None, :, None, None]            # [3, D, Ho, Wo]

    ang = 2 * np.pi * np.arange(N) / N + 0.3
    centers = ring_radius * np.stack([np.cos(ang), np.zeros(N), -np.sin(ang)], 1)  # [N, 3]

    grids = np.empty((N, D, Ho, Wo, 2), np.float32)
    gmask = np.empty((N, D, Ho, Wo, 1), bool)
    for k in range(N):
        p = pts - centers[k][:, None, None, None]
        x, y, z = p
        lon_k = -np.arctan2(z, x)
        lat_k = np.arctan2(y, np.sqrt(x * x + z * z))
        u = lon_k / np.pi
        v = lat_k / (np.pi / 2) * 2 + 1
        grids[k, ..., 0] = u
        grids[k, ..., 1] = v
        gmask[k, ..., 0] = (v <= 1.0) & (v >= -1.0)

    # per-camera field-of-view mask on the image-resolution surrogate
    latm = -np.pi / 2 + (np.arange(Hm, dtype=np.float64) + 0.5) / Hm * (np.pi / 2)
    lonm = -np.pi + (np.arange(Wm, dtype=np.float64) + 0.5) / Wm * (2 * np.pi)
    latm, lonm = np.meshgrid(latm, lonm, indexing="ij")
    dirm = np.stack([np.cos(latm) * np.cos(lonm), np.sin(latm), -np.cos(latm) * np.sin(lonm)], 0)
    masks = np.empty((N, 1, Hm, Wm), np.float32)
    cos_half = np.cos(np.deg2rad(fov_deg) / 2)
    for k in range(N):
        axis = centers[k] / np.linalg.norm(centers[k])
        masks[k, 0] = (np.tensordot(axis, dirm, axes=1) > cos_half).astype(np.float32)
    return grids, gmask, masks


# ----------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------
def make_inputs(cfg: PathConfig, seed: int = 0, batch: int = 1, grid_kind: str = "smooth",
                grid_mask_dtype: str = "bool") -> Dict[str, np.ndarray]:
    """feats [B,N,C,Hi,Wi] f32, grids [B,N,D,Ho,Wo,2] f32, grid_masks [B,N,D,Ho,Wo,1]
    bool|f32, masks [B,N,1,Hm,Wm] f32 (SURVEY.md §8(b) signature)."""
    rng = np.random.default_rng(seed)
    N, C, D = cfg.num_cams, cfg.feat_chs, cfg.num_cands
    Hi, Wi = cfg.feat_hw
    Ho, Wo = cfg.cv_hw
    Hm, Wm = cfg.mask_hw
    feats = rng.standard_normal((batch, N, C, Hi, Wi), dtype=np.float32)
    if grid_kind == "smooth":
        g, gm, m = smooth_grids(cfg)
        grids = np.broadcast_to(g[None], (batch,) + g.shape).copy()
        grid_masks = np.broadcast_to(gm[None], (batch,) + gm.shape).copy()
        masks = np.broadcast_to(m[None], (batch,) + m.shape).copy()
    elif grid_kind == "random":
        grids = rng.uniform(-1.1, 1.1, (batch, N, D, Ho, Wo, 2)).astype(np.float32)
        grid_masks = rng.random((batch, N, D, Ho, Wo, 1)) < 0.9
        masks = (rng.random((batch, N, 1, Hm, Wm)) < 0.95).astype(np.float32)
    else:
        raise ValueError(grid_kind)
    if grid_mask_dtype == "f32":
        grid_masks = grid_masks.astype(np.float32)
    return dict(feats=feats, grids=grids, grid_masks=grid_masks, masks=masks)


# ----------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------
def _conv_block(rng, prefix, cin, cout, has_norm, has_bias, out):
    fan_in = cin * 27
    bound = float(np.sqrt(6.0 / fan_in))
    out[f"{prefix}.conv_layer.weight"] = rng.uniform(-bound, bound, (cout, cin, 3, 3, 3)).astype(np.float32)
    if has_bias:
        out[f"{prefix}.conv_layer.bias"] = rng.normal(0, 0.1, (cout,)).astype(np.float32)
    if has_norm:
        out[f"{prefix}.norm_layer.weight"] = rng.uniform(0.5, 1.5, (cout,)).astype(np.float32)
        out[f"{prefix}.norm_layer.bias"] = rng.normal(0, 0.1, (cout,)).astype(np.float32)
        out[f"{prefix}.norm_layer.running_mean"] = rng.normal(0, 0.1, (cout,)).astype(np.float32)
        out[f"{prefix}.norm_layer.running_var"] = rng.uniform(0.5, 1.5, (cout,)).astype(np.float32)
        out[f"{prefix}.norm_layer.num_batches_tracked"] = np.asarray(1, np.int64)


def make_weights(cfg: PathConfig, seed: int = 0, gain: float = 1.0) -> Dict[str, Dict[str, np.ndarray]]:
    """State dicts (reference key names) for cv_builder and
    cv_regulator.  `gain` scales the last conv so that the softmax over D is
    peaky like a trained network's (SURVEY.md §7 'Precision vs the 1e-3 bar')."""
    rng = np.random.default_rng(10_000 + seed)
    builder: Dict[str, np.ndarray] = {}
    _conv_block(rng, "post_vol", cfg.vol_chs, cfg.vol_chs, True, False, builder)
    reg: Dict[str, np.ndarray] = {}
    for prefix, cin, cout, has_norm, has_bias in regulator_conv_specs(cfg.reg_in_chs, cfg.reg_f_int_chs):
        _conv_block(rng, prefix, cin, cout, has_norm, has_bias, reg)
    reg["out_costs.1.conv_layer.weight"] = (reg["out_costs.1.conv_layer.weight"] * np.float32(gain)).astype(np.float32)
    return dict(cv_builder=builder, cv_regulator=reg)


def _conv_block2d(rng, prefix, cin, cout, k, out):
    bound = float(np.sqrt(6.0 / (cin * k * k)))
    out[f"{prefix}.conv_layer.weight"] = rng.uniform(-bound, bound, (cout, cin, k, k)).astype(np.float32)
    out[f"{prefix}.norm_layer.weight"] = rng.uniform(0.5, 1.5, (cout,)).astype(np.float32)
    out[f"{prefix}.norm_layer.bias"] = rng.normal(0, 0.1, (cout,)).astype(np.float32)
    out[f"{prefix}.norm_layer.running_mean"] = rng.normal(0, 0.1, (cout,)).astype(np.float32)
    out[f"{prefix}.norm_layer.running_var"] = rng.uniform(0.5, 1.5, (cout,)).astype(np.float32)
    out[f"{prefix}.norm_layer.num_batches_tracked"] = np.asarray(1, np.int64)


def make_extractor_weights(seed: int = 0, in_chs: int = 3, chs: int = 16, layers=(5, 10),
                           out_gain: float = 0.06) -> Dict[str, np.ndarray]:
    """State dict of SimpleFeatExtraction (reference key names: first, blks.{i}[.blk1|.blk2], final_layer).
    `out_gain` scales the last BatchNorm so that the features come out O(1) like the N(0,1) features
    the hot-path cases use (31 He-initialised layers with residual adds otherwise grow them ~18x, which
    turns the downstream softmax into an arg-max and the end-to-end test into a tie-breaking test)."""
    rng = np.random.default_rng(20_000 + seed)
    sd: Dict[str, np.ndarray] = {}
    _conv_block2d(rng, "first", in_chs, chs, 5, sd)
    i = 0
    for step, n in enumerate(layers):
        for _ in range(n):
            _conv_block2d(rng, f"blks.{i}.blk1", chs, chs, 3, sd)
            _conv_block2d(rng, f"blks.{i}.blk2", chs, chs, 3, sd)
            i += 1
        if step != len(layers) - 1:
            _conv_block2d(rng, f"blks.{i}", chs, chs, 3, sd)
            i += 1
    _conv_block2d(rng, "final_layer", chs, chs, 3, sd)
    sd["final_layer.norm_layer.weight"] = (sd["final_layer.norm_layer.weight"] * np.float32(out_gain)).astype(np.float32)
    sd["final_layer.norm_layer.bias"] = (sd["final_layer.norm_layer.bias"] * np.float32(out_gain)).astype(np.float32)
    return sd


def make_images(cfg: PathConfig, seed: int = 0, batch: int = 1) -> np.ndarray:
    """imgs [B, N, 3, 4*Hi, 4*Wi] in [0, 1) (the extractor's output is the 1/4-resolution feature map)."""
    rng = np.random.default_rng(30_000 + seed)
    Hi, Wi = cfg.feat_hw
    return rng.random((batch, cfg.num_cams, 3, 4 * Hi, 4 * Wi), dtype=np.float32)


def digest(arrays: Dict[str, np.ndarray]) -> str:
    """Order-independent sha256 over named arrays: lets a test on the GPU box prove it
    regenerated exactly the inputs the golden outputs were computed from."""
    h = hashlib.sha256()
    for k in sorted(arrays):
        a = np.ascontiguousarray(arrays[k])
        h.update(k.encode())
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()
