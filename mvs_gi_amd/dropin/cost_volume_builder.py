"""Drop-in cv_builder modules: same names, constructor arguments, child names and forward
signatures as dsta_mvs/model/cost_volume_builder/spherical_sweep_avg.py:9-167
(SphericalSweepStdMasked) and spherical_sweep.py:9-102 (SphericalSweep); the sweep and
post_vol run as HIP kernels (mvsgi_sweep_*_f32, mvsgi_conv3d_f32).

forward(feats [B,N,C,Hi,Wi], grids [B,N,D,Ho,Wo,2], grid_masks [B,N,D,Ho,Wo,1] bool|f32,
        masks [B,N,1,Hm,Wm]) -> vol [B,Cv,D,Ho,Wo]
The returned tensor has the reference's shape and channels_last_3d strides (its storage is
[B,D,Ho,Wo,Cv]); `.contiguous()` gives the reference's memory order if a caller needs it.
"""
from __future__ import annotations

from torch import nn, Tensor

import os
import weakref

import torch

from .. import hip_ops as H
from . import common_modules as cm
from .common_modules import NORM3D_TYPE, RELU_TYPE


_RIG_VALIDITY = weakref.WeakKeyDictionary()      # module -> ((grids, grid_masks, masks), versions, vmask, shared grids | None)
_RIG_CACHE_ENV = os.environ.get("MVSGI_RIG_CACHE", "1") != "0"


class _SweepBase(nn.Module):
    cache_rig_constants = True       # class attribute: unpickled reference modules get it too
    __getstate__ = cm.module_getstate

    def __init__(self, num_cams: int, feat_chs: int, post_k_sz: int, norm_type: str = "batch",
                 relu_type: str = "leaky"):
        super().__init__()
        self.num_cams = num_cams
        self.feat_chs = feat_chs
        self.norm_type = NORM3D_TYPE[norm_type]
        self.relu_type = RELU_TYPE[relu_type]
        self.grid_sample_mode = "bilinear"
        self.post_vol = cm.BaseConvBlk3d(in_chs=feat_chs, out_chs=feat_chs, kernel_size=post_k_sz,
                                         activation=self.relu_type(), norm_layer=self.norm_type(feat_chs))


def _rig_hit(entry, tensors) -> bool:
    """A cache entry holds STRONG references to the three rig tensors it was computed from and their in-place
    version counters: identity (`is`) + version decide a hit.  Holding the references also keeps the caching
    allocator from handing the same addresses to another rig's tensors (a key built from data_ptr() alone would
    then match a freed-and-reallocated rig and silently reuse the wrong validity byte)."""
    return entry is not None and all(a is b for a, b in zip(entry[0], tensors)) \
        and entry[1] == tuple(t._version for t in tensors)


def _rig_cache_usable(owner, feats, grids) -> bool:
    """Whether the validity-byte kernels serve this call (decided ONCE per forward, before a code path is chosen)."""
    return owner is not None and getattr(owner, "cache_rig_constants", True) and _RIG_CACHE_ENV \
        and H.nhwc_sweep_ok(feats) and grids.is_cuda


def _owned_split_buffer(owner, attr: str, key: tuple, make):
    """Module-owned split-padded buffers, one per shape key, NEVER replaced or freed while the module lives: a captured
    hipGraph holds their addresses, and their zero borders are written exactly once (at allocation)."""
    bufs = owner.__dict__.setdefault(attr, {})
    if key not in bufs:
        bufs[key] = make()
    return bufs[key]


def std_sweep_ndhwc(feats, grids, grid_masks, masks, owner=None, split_out: bool = False, buf_frames: int = 0):
    """Masked-variance sweep.  grids / grid_masks / masks are constants of the camera rig (the
    reference builds them once, api/inference_class.py:40-45), so the mask half of the sweep
    (spherical_sweep_avg.py:92-102) is evaluated once per rig and cached on `owner`, keyed on the
    three tensors' storage pointer, in-place version counter, shape and dtype: a new or modified
    tensor recomputes it.  `owner.cache_rig_constants = False` (or MVSGI_RIG_CACHE=0) re-samples
    the masks every call."""
    use_cache = _rig_cache_usable(owner, feats, grids)
    if not use_cache:
        return None if split_out else H.sweep_std(feats, grids, grid_masks, masks)
    tensors = (grids, grid_masks, masks)               # identity of what the caller passed
    cached = _RIG_VALIDITY.get(owner)
    if not _rig_hit(cached, tensors):
        # a batch-broadcast view (stride 0 along the batch, e.g. `grids[:1].expand(B, ...)`) is ONE rig for every frame:
        # the kernel then reads frame 0's constants for all frames (they stay in L2) instead of B copies from HBM
        if all(t.dim() > 0 and t.shape[0] > 1 and t.stride(0) == 0 for t in tensors):
            g1, gm1, m1 = (t[:1].contiguous() for t in tensors)
        else:
            g1, gm1, m1 = grids, grid_masks, masks
        cached = (tensors, tuple(t._version for t in tensors), H.sweep_validity(g1, gm1, m1), g1 if g1 is not grids else None)
        _RIG_VALIDITY[owner] = cached            # weak on the module: dies with it, never pickled with it
    g_use = cached[3] if cached[3] is not None else grids
    if split_out:
        # vol_raw straight into a module-owned split-padded buffer (zero border, allocated once per shape) for the
        # register-stationary post_vol
        # (`buf_frames` > B: the buffer is sized for a whole chunk and a shorter tail chunk uses a leading slice of it)
        B, D, Ho, Wo = feats.shape[0], grids.shape[2], grids.shape[3], grids.shape[4]
        nb = max(B, int(buf_frames))
        full = _owned_split_buffer(owner, "_mvsgi_rs_vol", (nb, D, Ho, Wo, feats.device),
                                   lambda: H.SplitAct(nb, D, Ho, Wo, 16, feats.device))
        out = full if nb == B else H.SplitAct(B, D, Ho, Wo, 16, feats.device, buf=full.buf[:B])
        return H.sweep_std_valid_split(feats, g_use, cached[2], out=out, fmt=H.mode_fmt())
    return H.sweep_std_valid(feats, g_use, cached[2])


def cat_sweep_ndhwc(feats, grids) -> Tensor:
    return H.sweep_cat(feats, grids)


_USE_RS = H.exp_env("MVSGI_RS", "1") != "0"
# frames per sweep -> post_vol chunk (0: the whole batch at once): a chunk's split-padded vol_raw is written by the sweep and read
# straight back by post_vol, and the buffer is 16 frames instead of B.  4763 -> 4835 frames/s at B=64 (chunks of 4: 4845;
# tools/ab_bench.py, one box, 3 rounds); the same chunking of out_costs.0 -> head measured slower (4769 -> 4667)
_FRONT_CHUNK = int(os.environ.get("MVSGI_FRONT_CHUNK", "16"))
_RS_MIN_UNITS = int(H.exp_env("MVSGI_RS_MIN_UNITS", "0"))      # measured faster down to one frame (B=1: 16.9 vs 24.2 us, 23.5 vs 32.2 us)


def _std_front(self, feats: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor, hand_over_split: bool):
    """sweep -> split-padded vol_raw -> register-stationary post_vol (csrc/conv3d_rs.hip).  Returns the fp32 volume
    [B, D, H, W, 16], or (hand_over_split) post_vol's output as a module-owned split-padded H.SplitAct -- the form the
    regulator's stride-2 first layer stages by LDS-DMA (csrc/conv3d_s2rs.hip) -- or None when the layer shapes / mode do not
    put post_vol on that kernel."""
    L = cm.lower_conv_block(self.post_vol)
    if not (_USE_RS and H.split_mode() and L.cin == 16 and L.cout == 16 and L.stride == 1
            and 0.0 <= L.neg_slope <= 1.0 and feats.dim() == 5 and feats.shape[2] == 16 and grids.dim() == 6):
        return None
    B, D, Ho, Wo = feats.shape[0], grids.shape[2], grids.shape[3], grids.shape[4]
    if not (B * ((D + 3) // 4) * ((Ho + 3) // 4) * ((Wo + 15) // 16) >= _RS_MIN_UNITS and _rig_cache_usable(self, feats, grids)):
        return None
    wp_rs, sc_rs = L._rs(H.mode_fmt())      # post_vol's weights / scale in the split the sweep writes
    xs = None
    if hand_over_split:      # one buffer per shape, never replaced while the module lives (a captured hipGraph holds the address)
        xs = _owned_split_buffer(self, "_mvsgi_rs_x0", (B, D, Ho, Wo, feats.device), lambda: H.SplitAct(B, D, Ho, Wo, 16, feats.device))
    k = _FRONT_CHUNK
    shared = all(t.dim() > 0 and t.shape[0] > 1 and t.stride(0) == 0 for t in (grids, grid_masks, masks))
    if k > 0 and B > k and shared:
        # frames in chunks of k: a chunk's split-padded vol_raw (k x 30 MB) is written by the sweep and read straight back
        # by post_vol while much of it is still in the 256 MB memory-side cache; the buffer is k frames, not B
        y = None if hand_over_split else torch.empty((B, D, Ho, Wo, 16), device=feats.device, dtype=torch.float32)
        for i in range(0, B, k):
            j = min(i + k, B)
            vs = std_sweep_ndhwc(feats[i:j], grids, grid_masks, masks, owner=self, split_out=True, buf_frames=k)    # the rig tensors whole: cache identity
            if hand_over_split:
                H.conv3d_rs16(vs, wp_rs, sc_rs, L.shift, neg_slope=L.neg_slope,
                              out_split=H.SplitAct(j - i, D, Ho, Wo, 16, feats.device, buf=xs.buf[i:j]))
            else:
                H.conv3d_rs16(vs, wp_rs, sc_rs, L.shift, neg_slope=L.neg_slope, out=y[i:j])
        if hand_over_split:
            xs.fmt = H.mode_fmt()      # written chunk by chunk through views
        return xs if hand_over_split else y
    vs = std_sweep_ndhwc(feats, grids, grid_masks, masks, owner=self, split_out=True)
    if hand_over_split:
        return H.conv3d_rs16(vs, wp_rs, sc_rs, L.shift, neg_slope=L.neg_slope, out_split=xs)
    return H.conv3d_rs16(vs, wp_rs, sc_rs, L.shift, neg_slope=L.neg_slope)


def std_forward(self, feats: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor) -> Tensor:
    y = _std_front(self, feats, grids, grid_masks, masks, False)
    if y is not None:
        return cm._to_ncdhw_view(y)
    L = cm.lower_conv_block(self.post_vol)
    vol_raw = std_sweep_ndhwc(feats, grids, grid_masks, masks, owner=self)
    return cm._to_ncdhw_view(L.run(vol_raw))


def std_forward_split(self, feats: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor):
    """forward() with the cost volume handed over as a split-padded H.SplitAct (module-owned, valid until the next call) instead
    of an fp32 tensor, for a regulator that takes it (cost_volume_regulator.regulator_takes_split); None when this builder
    configuration does not produce it -- the caller then uses forward()."""
    return _std_front(self, feats, grids, grid_masks, masks, True)


def cat_forward(self, feats: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor) -> Tensor:
    # grid_masks / masks are accepted and ignored, as in the reference (spherical_sweep.py:80)
    vol_raw = cat_sweep_ndhwc(feats, grids)
    return cm._to_ncdhw_view(cm.lower_conv_block(self.post_vol).run(vol_raw))


class SphericalSweepStdMasked(_SweepBase):
    def sweep(self, feats: Tensor, grids: Tensor, grid_masks: Tensor, masks: Tensor) -> Tensor:
        """vol_raw [B, C, D, Ho, Wo] (spherical_sweep_avg.py:38-136)."""
        return cm._to_ncdhw_view(std_sweep_ndhwc(feats, grids, grid_masks, masks, owner=self))

    forward = std_forward
    forward_split = std_forward_split


class SphericalSweep(_SweepBase):
    def sweep(self, feats: Tensor, grids: Tensor, masks: Tensor = None) -> Tensor:
        """vol_raw [B, N*C, D, Ho, Wo], channel = cam*C + c (spherical_sweep.py:38-68)."""
        return cm._to_ncdhw_view(cat_sweep_ndhwc(feats, grids))

    forward = cat_forward
