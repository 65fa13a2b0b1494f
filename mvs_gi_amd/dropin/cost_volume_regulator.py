"""Drop-in cv_regulator: UNetCostVolumeRegulatorBase (dsta_mvs/model/cost_volume_regulator/
unet_regulator.py:16-140), the older UNetCostVolumeRegulator (:142-270; widths in*2/4/8,
identical state dict and output to Base(in, 2*in), SURVEY.md §8 note) and UNetDownBlk
(:273-310).  forward(vol [B,C,D,H,W]) -> costs [B,final_chs,D,H,W]; every layer is one
mvsgi_conv3d_f32 / mvsgi_resize_trilinear_f32 launch on the current stream.
"""
from __future__ import annotations

import copy
import os
from typing import Sequence

from torch import nn, Tensor

from .. import hip_ops as H
from . import common_modules as cm
from .common_modules import NORM3D_TYPE, RELU_TYPE


class UNetDownBlk(nn.Module):
    def __init__(self, in_chs: int, out_chs: int, kernel_size: int, width: int,
                 activation: nn.Module = cm.NoOp(), norm_layer: nn.Module = cm.NoOp()):
        super().__init__()
        self.k_sz = kernel_size
        self.first = cm.BaseConvBlk3d(in_chs, out_chs, kernel_size, stride=2,
                                      activation=copy.deepcopy(activation), norm_layer=copy.deepcopy(norm_layer))
        self.blks = nn.Sequential(*[
            cm.ResConvBlk3d(out_chs, out_chs, kernel_size, activation=copy.deepcopy(activation),
                            norm_layer=copy.deepcopy(norm_layer)) for _ in range(width - 1)])

    __getstate__ = cm.module_getstate

    def forward(self, x: Tensor) -> Tensor:
        return cm._to_ncdhw_view(down_block_ndhwc(self, H.as_ndhwc(x)))


# MVSGI_RS=0 keeps every layer on the streaming kernels.  MVSGI_RS_MIN_UNITS: minimum
# 128-voxel bricks per launch (one workgroup per CU, each with a prologue and two drain phases: small launches lose)
_USE_RS = H.exp_env("MVSGI_RS", "1") != "0"
_USE_WINO = os.environ.get("MVSGI_WINO", "1") != "0"        # MVSGI_WINO=0: the level-0 residual convs stay on the direct kernel in the fp16 split too
_WINO_A32 = H.exp_env("MVSGI_WINO_A32", "1") != "0"          # (experiment switch: 0 = split-padded fp16 pairs between the Winograd layers)
_RS_MIN_UNITS = int(H.exp_env("MVSGI_RS_MIN_UNITS", "0"))      # measured faster down to one frame (B=1: 16.9 vs 24.2 us, 23.5 vs 32.2 us)


def _rs_chain(blk, x: Tensor):
    """The residual convs of a UNetDownBlk as launch records if ALL of them can run register-stationary on split-padded
    activations (32 -> 32 channels: UNet level 0 of the (16, 32) regulator), else None."""
    if not (_USE_RS and H.split_mode() and len(blk.blks) > 0):
        return None
    L0 = cm.lower_conv_block(blk.first)
    if L0.cin % 16 or L0.cout != 32:
        return None
    B, D, Hh, W, _ = x.shape
    s = L0.stride
    Do, Ho, Wo = (D - 1) // s + 1, (Hh - 1) // s + 1, (W - 1) // s + 1
    if B * ((Do + 1) // 2) * ((Ho + 3) // 4) * ((Wo + 15) // 16) < _RS_MIN_UNITS:
        return None
    chain = []
    for rb in blk.blks:
        if not cm._is_identity(rb.one_by_one) or getattr(rb, "out_pad", 0) != 0:
            return None
        L1, L2 = cm.lower_conv_block(rb.blk1), cm.lower_conv_block(rb.blk2)
        if not (L1.rs_ok() and L2.rs_ok()):
            return None
        chain.append((L1, L2))
    return L0, chain, (B, Do, Ho, Wo)


_CUS = {}


def _cu_count(device) -> int:
    import torch
    key = str(device)
    if key not in _CUS:
        _CUS[key] = torch.cuda.get_device_properties(device).multi_processor_count
    return _CUS[key]


def _wino_pays(B: int, Ho: int, Wo: int, device) -> bool:
    """A Winograd unit (2 rows x 32 columns, all planes) occupies a CU for ~16 us: the form pays when the launch's units fill
    their rounds of one unit per CU (measured on MI355X, [8, 40, 160]: 1 frame 0.95-0.99 x the direct kernel, 2 frames 1.26,
    3 frames -- 300 units, the second round 17 % full -- 0.92-0.98, 4 frames 1.15, 16 and more 1.21-1.27)."""
    units = B * (Ho // 2) * (Wo // 32)
    cus = _cu_count(device)
    rounds = -(-units // cus)
    return units >= 0.7 * rounds * cus


def _down_block_rs(blk, x: Tensor, L0, chain, dims) -> Tensor:
    """first conv (streaming kernel, output written split-padded) -> residual blocks on the register-stationary kernel, three
    rotating split-padded buffers owned by the module (zero borders, allocated once) -> last conv writes plain fp32."""
    B, Do, Ho, Wo = dims
    # one buffer set per shape, never replaced while the module lives (a captured hipGraph holds the addresses; the zero
    # borders are written once, at allocation)
    sets = blk.__dict__.setdefault("_mvsgi_rs_bufs", {})
    key = (B, Do, Ho, Wo, x.device)
    if key not in sets:
        sets[key] = [H.SplitAct(B, Do, Ho, Wo, 32, x.device) for _ in range(3)]
    b = sets[key]
    fmt = H.mode_fmt()              # the split of this chain's activations and weights (the library's mode)
    # fp16 split on a [8 | 16, even, 32 k] volume: the Winograd form (2.25 x fewer matrix instructions, csrc/conv3d_wino.hip) when its
    # units fill the chip; behind the split-padded hand-over its layers pass fp32-padded activations to each other
    wino = fmt == "f16" and _USE_WINO and _wino_pays(B, Ho, Wo, x.device) and \
        all(L1.wino_ok(Do, Ho, Wo) and L2.wino_ok(Do, Ho, Wo) for L1, L2 in chain)
    if isinstance(x, H.SplitAct):      # the builder handed post_vol's output over split-padded: staged by LDS-DMA (csrc/conv3d_s2rs.hip)
        if x.fmt != fmt:
            raise RuntimeError(f"split-padded cost volume holds {x.fmt} pieces, the library's mode writes {fmt}")
        wp0, sh0, un0 = L0._s2(fmt)
        H.conv3d_s2rs(x, wp0, sh0, out=b[0], neg_slope=L0.neg_slope, unscale=un0, out_f32p=wino and _WINO_A32)
    else:
        wp0, sc0 = L0._b3(fmt)
        H.conv3d_out_split(x, wp0, sc0, L0.shift, out=b[0], stride=L0.stride, neg_slope=L0.neg_slope, fmt=fmt)
    conv = H.conv3d_wino if wino else H.conv3d_rs
    cur, out = 0, None
    for i, (L1, L2) in enumerate(chain):
        r, y = (cur + 1) % 3, (cur + 2) % 3
        (wp1, sc1) = L1._wino() if wino else L1._rs(fmt)
        (wp2, sc2) = L2._wino() if wino else L2._rs(fmt)
        conv(b[cur], wp1, sc1, L1.shift, neg_slope=L1.neg_slope, out=b[r])
        if i == len(chain) - 1:
            out = conv(b[r], wp2, sc2, L2.shift, res=b[cur], neg_slope=L2.neg_slope, out_f32=True)
        else:
            conv(b[r], wp2, sc2, L2.shift, res=b[cur], neg_slope=L2.neg_slope, out=b[y])
            cur = y
    return out


# MVSGI_S2RS=0: the builder -> regulator hand-over stays an fp32 tensor (the regulator's first layer on the streaming kernel).
# MVSGI_S2RS_MIN_FRAMES: smallest batch for the split-padded hand-over
_USE_S2RS = os.environ.get("MVSGI_S2RS", "1") != "0"
_S2RS_MIN_FRAMES = int(H.exp_env("MVSGI_S2RS_MIN_FRAMES", "1"))


class _Shape:
    __slots__ = ("shape",)

    def __init__(self, shape):
        self.shape = tuple(shape)


def regulator_takes_split(self, shape) -> bool:
    """True when forward_split_in() can take a split-padded cost volume of geometry `shape` = (B, D, H, W, C): the (16, 32)
    regulator in split-bf16 mode, whose first layer (16 -> 32, stride 2) then stages pre-split voxels by LDS-DMA and whose
    level-0 residual blocks run register-stationary."""
    if not (_USE_S2RS and H.split_mode() and len(self.down_blks) > 0 and len(shape) == 5 and shape[4] == 16
            and shape[0] >= _S2RS_MIN_FRAMES):
        return False
    blk = self.down_blks[0]
    return cm.lower_conv_block(blk.first).s2rs_ok() and _rs_chain(blk, _Shape(shape)) is not None


def regulator_forward_split_in(self, xs) -> Tensor:
    """forward() on the builder's split-padded cost volume (H.SplitAct, 16 channels); see regulator_takes_split."""
    if not regulator_takes_split(self, xs.shape):
        raise RuntimeError(f"this regulator does not take a split-padded volume of geometry {xs.shape}: call forward() with the fp32 tensor")
    return cm._to_ncdhw_view(regulator_forward_ndhwc(self, xs))


def down_block_ndhwc(blk, x) -> Tensor:
    rs = _rs_chain(blk, x)
    if rs is not None:
        return _down_block_rs(blk, x, *rs)
    if isinstance(x, H.SplitAct):
        raise RuntimeError("split-padded input needs the register-stationary level-0 chain (regulator_takes_split)")
    x = cm.lower_conv_block(blk.first).run(x)
    for rb in blk.blks:
        x = cm.res_block_ndhwc(rb, x)
    return x


# MVSGI_POLY=0 keeps out_costs.0 on the streaming kernel with the upsample evaluated in its producers.  MVSGI_POLY_MIN_UNITS: minimum
# 128-cell bricks (low resolution) per launch for the polyphase form
_USE_POLY = os.environ.get("MVSGI_POLY", "1") != "0"
_HEAD_SPLIT = os.environ.get("MVSGI_HEAD_SPLIT", "1") != "0"      # 0: polyphase out_costs.0 writes fp32 and the exact-fp32 head reads it
# (measured on MI355X, G16V, 400 bricks per frame and role, one hipGraph replay per step: 1 frame 0.506 vs 0.497 ms with / without,
# 2 frames 0.779 vs 0.766, 4 frames 1.150 vs 1.167, 64 frames 12.57 vs 13.28 -- three launches and a prologue + two drain phases
# per workgroup need ~4 frames to pay)
_POLY_MIN_UNITS = int(H.exp_env("MVSGI_POLY_MIN_UNITS", "1600"))


def _poly_tail(self, x: Tensor, skip: Tensor):
    """The last up block + out_costs as (up block writing its result split-padded) -> (polyphase ResizeConv3d on the
    register-stationary kernel) -> (cost head), when the layer shapes allow it (32 -> 16 channels: the (16, 32) regulator);
    returns costs [B, D, H, W, final_chs] or None."""
    if not (_USE_POLY and len(self.upBlks) > 0):
        return None
    up, oc = self.upBlks[len(self.upBlks) - 1], self.out_costs[0]
    Lu, Lo = cm.lower_conv_block(up.conv), cm.lower_conv_block(oc.conv)
    if not (Lo.poly_ok() and Lu.can_fuse_up2() and Lu.cout == 32 and oc.scale == 2 and oc.out_pad == 0 and up.scale == 2 and up.out_pad == 0):
        return None
    B, Dl, Hl, Wl, _ = x.shape
    if tuple(skip.shape[1:4]) != (2 * Dl, 2 * Hl, 2 * Wl):
        return None                                      # odd pyramid: the second trilinear resize of common_modules.py:343-350
    if B * ((2 * Dl + 1) // 2) * ((2 * Hl + 3) // 4) * ((2 * Wl + 15) // 16) < _POLY_MIN_UNITS:
        return None
    bufs = self.__dict__.setdefault("_mvsgi_poly_bufs", {})          # one split-padded buffer per shape, never replaced
    key = (B, 2 * Dl, 2 * Hl, 2 * Wl, x.device)
    if key not in bufs:
        bufs[key] = H.SplitAct(B, 2 * Dl, 2 * Hl, 2 * Wl, 32, x.device)
    xs = Lu.run_up2_split(x, skip, bufs[key])
    # the cost head reads split-padded fragments straight into the matrix cores: out_costs.0 then writes that format
    Lh = cm.lower_conv_block(self.out_costs[1])
    if _HEAD_SPLIT and Lh.head_split_ok() and Lh.cin == 16:
        hkey = (B, 4 * Dl, 4 * Hl, 4 * Wl, "hi", x.device)
        if hkey not in bufs:
            bufs[hkey] = H.SplitAct(B, 4 * Dl, 4 * Hl, 4 * Wl, 16, x.device)
        return Lh.run_head_split(Lo.run_up2_poly_split(xs, bufs[hkey]))
    return Lh.run(Lo.run_up2_poly(xs))


def regulator_forward_ndhwc(self, x: Tensor) -> Tensor:
    """unet_regulator.py:120-140 on channels-last tensors."""
    skips = []
    n_down = len(self.down_blks)
    for i, blk in enumerate(self.down_blks):
        x = down_block_ndhwc(blk, x)
        if i != n_down - 1:
            skips.append(x)
    skips.reverse()
    n_up = len(self.upBlks)
    for i, up in enumerate(self.upBlks):
        if i == n_up - 1:
            y = _poly_tail(self, x, skips[i])          # ... -> out_costs.0 (polyphase) -> out_costs.1
            if y is not None:
                return y
        x = cm.resize_conv_ndhwc(up, x, res=skips[i])
    y = _split_head_tail(self, x)          # out_costs.0 writes the split-padded format the split-bf16 cost head reads
    if y is not None:
        return y
    x = cm.resize_conv_ndhwc(self.out_costs[0], x)
    return cm.lower_conv_block(self.out_costs[1]).run(x)


def _split_head_tail(self, x: Tensor):
    """out_costs of a regulator the polyphase tail does not serve (every shape but (16, 32)): the ResizeConv3d on the streaming
    kernel (upsample in its producers) writes its result split-padded into a module-owned buffer, and the cost head
    (csrc/conv3d_headsplit.hip) takes whole B operands from it -- 2 x 16 matrix-core cycles per 16 voxels and 16 taps instead of the
    exact-fp32 head's 8 x 64 per 32 voxels, which is bound by the fp32 matrix rate.  Returns costs [B, D, H, W, 1] or None."""
    oc = self.out_costs[0]
    Lo, Lh = cm.lower_conv_block(oc.conv), cm.lower_conv_block(self.out_costs[1])
    if not (_HEAD_SPLIT and Lh.head_split_ok() and Lo.can_fuse_up2() and oc.scale == 2 and oc.out_pad == 0 and Lh.cin == Lo.cout):
        return None
    B, Dl, Hl, Wl, _ = x.shape
    if (2 * Dl + 2) * (2 * Hl + 4) * (2 * Wl + 4) * Lo.cout * 4 >= 2 ** 31:        # the launcher's bound for a split-padded output frame (32-bit offsets)
        return None
    return Lh.run_head_split(Lo.run_up2_split(x, None, _owned_head_buffer(self, B, 2 * Dl, 2 * Hl, 2 * Wl, Lo.cout, x.device)))


def _owned_head_buffer(self, B, D, Hh, W, C, device):
    """The module-owned split-padded buffer between out_costs.0 and the split head: ONE allocation per frame geometry, sized for the
    largest batch seen, handed out as a view of its first B frames (the format is per frame: a prefix of the batch is a valid
    buffer).  A caller that varies its batch (batch sweeps, dataset tails) therefore holds one buffer, not one per batch size; a
    larger batch replaces it -- unless a captured hipGraph may hold its address (`_mvsgi_pinned`, set while capturing), in which
    case the old one stays alive beside the new."""
    import torch
    bufs = self.__dict__.setdefault("_mvsgi_head_bufs", {})
    key = (D, Hh, W, C, device)
    ent = bufs.get(key)
    if ent is None or ent[0].B < B:
        big = H.SplitAct(B, D, Hh, W, C, device)
        keep = ent[1] + [ent[0]] if ent is not None and (ent[2] or ent[1]) else []      # graphs may hold the old buffers
        ent = bufs[key] = [big, keep, False]
    if torch.cuda.is_current_stream_capturing():
        ent[2] = True
    big = ent[0]
    if big.B == B:
        return big
    return H.SplitAct(B, D, Hh, W, C, device, buf=big.buf[:B])


def regulator_forward(self, x: Tensor) -> Tensor:
    H.check_range("cv_regulator: an earlier call")      # the fp16 split's range report (hip_ops.check_range; frames already completed)
    return cm._to_ncdhw_view(regulator_forward_ndhwc(self, H.as_ndhwc(x)))


class UNetCostVolumeRegulatorBase(nn.Module):
    def __init__(self, in_chs: int = 16, f_int_chs: int = 32, final_chs: int = 1, u_depth: int = 3,
                 blk_width: int = 4, stage_factor: int = 2, cost_k_sz: int = 3, keep_last_chs: Sequence[int] = [],
                 deconv_k_sz: int = 3, norm_type: str = "batch", relu_type: str = "leaky"):
        super().__init__()
        self.in_chs, self.f_int_chs, self.final_chs = in_chs, f_int_chs, final_chs
        self.u_depth, self.blkWidth, self.stage_factor = u_depth, blk_width, stage_factor
        self.k_sz, self.keep_last_chs, self.deconv_k_sz = cost_k_sz, keep_last_chs, deconv_k_sz
        self.norm_type = NORM3D_TYPE[norm_type]
        self.relu_type = RELU_TYPE[relu_type]

        # channel plan: f_int at level 0, multiplied by stage_factor per level unless kept
        self._int_chs = []
        downs = []
        cin, cout = in_chs, f_int_chs
        for lvl in range(u_depth):
            if lvl != 0 and lvl not in keep_last_chs:
                cout *= stage_factor
            downs.append(UNetDownBlk(cin, cout, cost_k_sz, blk_width, activation=self.relu_type(),
                                     norm_layer=self.norm_type(cout)))
            self._int_chs.append(cout)
            cin = cout
        self.down_blks = nn.ModuleList(downs)

        rev = self._int_chs[::-1]
        self.upBlks = nn.ModuleList([
            cm.ResizeConv3d(rev[i], rev[i + 1], deconv_k_sz, stride=2, activation=self.relu_type(),
                            norm_layer=self.norm_type(rev[i + 1])) for i in range(u_depth - 1)])
        self.out_costs = nn.Sequential(
            cm.ResizeConv3d(self._int_chs[0], in_chs, deconv_k_sz, stride=2, activation=self.relu_type(),
                            norm_layer=self.norm_type(in_chs)),
            cm.BaseConvBlk3d(in_chs, final_chs, deconv_k_sz, bias_on=True, activation=cm.NoOp(),
                             norm_layer=cm.NoOp()))

    forward = regulator_forward
    takes_split = regulator_takes_split
    forward_split_in = regulator_forward_split_in
    __getstate__ = cm.module_getstate


class UNetCostVolumeRegulator(nn.Module):
    """Older class named by un-suffixed recipes (configs/base_model.yaml:11-26 passes its kwargs)."""
    def __init__(self, in_chs: int, final_chs: int, u_depth: int, blk_width: int, stage_factor: int,
                 cost_k_sz: int, keep_last_chs: Sequence[int], deconv_k_sz: int, sweep_fuse_ch_reduce: int = 2,
                 num_cams: int = 3, only_one_cam: bool = False, norm_type: str = "batch", relu_type: str = "leaky"):
        super().__init__()
        self.in_chs = in_chs if only_one_cam else (in_chs * num_cams) // sweep_fuse_ch_reduce
        self.final_chs, self.u_depth, self.blkWidth = final_chs, u_depth, blk_width
        self.stage_factor, self.k_sz = stage_factor, cost_k_sz
        self.keep_last_chs, self.deconv_k_sz = keep_last_chs, deconv_k_sz
        self.norm_type = NORM3D_TYPE[norm_type]
        self.relu_type = RELU_TYPE[relu_type]

        downs = []
        cin = cout = self.in_chs
        for lvl in range(u_depth):
            if lvl not in keep_last_chs:
                cout *= stage_factor
            downs.append(UNetDownBlk(cin, cout, cost_k_sz, blk_width, activation=self.relu_type(),
                                     norm_layer=self.norm_type(cout)))
            if lvl not in keep_last_chs:
                cin *= stage_factor
        self.down_blks = nn.ModuleList(downs)

        top = self.in_chs * stage_factor ** (u_depth - len(keep_last_chs))
        ups = []
        cin = cout = top
        for i in range(u_depth - 1):
            if i + 1 not in keep_last_chs:
                cout //= stage_factor
            ups.append(cm.ResizeConv3d(cin, cout, deconv_k_sz, stride=2, activation=self.relu_type(),
                                       norm_layer=self.norm_type(cout)))
            if i + 1 not in keep_last_chs:
                cin //= stage_factor
        self.upBlks = nn.ModuleList(ups)
        self.out_costs = nn.Sequential(
            cm.ResizeConv3d(2 * self.in_chs, self.in_chs, deconv_k_sz, stride=2, activation=self.relu_type(),
                            norm_layer=self.norm_type(self.in_chs)),
            cm.BaseConvBlk3d(self.in_chs, final_chs, deconv_k_sz, bias_on=True, activation=cm.NoOp(),
                             norm_layer=cm.NoOp()))

    forward = regulator_forward
    takes_split = regulator_takes_split
    forward_split_in = regulator_forward_split_in
    __getstate__ = cm.module_getstate
