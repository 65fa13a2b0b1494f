"""Drop-in 2-D blocks and feature extractor: BaseConvBlk2d
(dsta_mvs/model/common/common_modules.py:18-80), ResConvBlk2d (:117-184) and SimpleFeatExtraction
(dsta_mvs/model/feature_extractor/simple_feature_extractor.py:8-84) with the reference's constructor
arguments, child names and state-dict keys; forward runs mvsgi_conv2d_f32.

forward(imgs [B*N, 3, H, W]) -> feats [B*N, chs, H/4, W/4], returned with the reference's shape and
channels-last strides (storage [B*N, H/4, W/4, chs]), which is exactly what the sweep kernel wants.
"""
from __future__ import annotations

import copy
import math
import os
from typing import Optional, Sequence, Tuple

import torch
from torch import nn, Tensor

from .. import hip_ops as H
from . import common_modules as cm
from .common_modules import module_getstate, NoOp, NORM2D_TYPE, RELU_TYPE, _is_identity


_SPLIT_CHAIN = H.exp_env("MVSGI_EXTRACTOR_SPLIT", "1") != "0"  # 0: the round-2 chain (fp32 activations between all layers)
_STEM_MFMA = H.exp_env("MVSGI_STEM_MFMA", "1") != "0"      # 0: the LDS-tiled VALU stem for uint8 images too


# The extractor's arithmetic: its split kernels are bf16-split in BOTH split modes of the library (bf16x3, f16x3) -- its output, the
# feature maps, is the INPUT of the path whose 1e-3 bar the modes are about (the sweep consumes them bit-exactly); MVSGI_CONV_MODE=f32
# runs it on the exact-fp32 kernels.
class Conv2dLaunch:
    __slots__ = ("w", "wp_b3", "wp_f32", "wp_stem", "wp_rs", "scale", "shift", "stride", "neg_slope", "k", "cin", "cout", "key")

    def run(self, x: Tensor, res: Optional[Tensor] = None, in_nchw: bool = False, out_split: Optional[Tensor] = None) -> Tensor:
        impl, wp = H.CONV_AUTO, None
        if H.split_mode() and self.k == 3 and self.cin % 16 == 0 and self.cout % 16 == 0 and not in_nchw:
            if self.wp_b3 is None:
                self.wp_b3 = H.pack_conv2d_weights_bf16x3(self.w)
            impl, wp = H.CONV_BF16X3, self.wp_b3
        elif self.k == 3 and self.cin % 16 == 0 and self.cout in (16, 32) and not in_nchw:
            if self.wp_f32 is None:                    # exact fp32 mode: fp32 MFMA kernel
                self.wp_f32 = H.pack_conv2d_weights_f32(self.w)
            impl, wp = H.CONV_MFMA, self.wp_f32
        elif x.dtype == torch.uint8 and (self.k, self.stride, self.cin, self.cout) == (5, 2, 3, 16) and _STEM_MFMA:
            if self.wp_stem is None:                   # camera images: the stem on the matrix cores (exact pixels, 24-bit weights)
                self.wp_stem = H.pack_conv2d_stem_weights(self.w)
            wp = self.wp_stem
        return H.conv2d(x, self.w, wp, self.scale, self.shift, res=res, stride=self.stride,
                        neg_slope=self.neg_slope, impl=impl, in_nchw=in_nchw, out_split=out_split)

    def rs_weights(self) -> Tensor:
        if self.wp_rs is None:
            self.wp_rs = H.pack_resblock2d_split_weights(self.w, self.scale)
        return self.wp_rs


def lower_conv2d_block(blk) -> Conv2dLaunch:
    conv: nn.Conv2d = blk.conv_layer
    norm = blk.norm_layer
    key = cm._fingerprint(blk)        # pointer and version as separate entries, bias, norm mode / eps, activation, stride
    cached = blk.__dict__.get("_mvsgi_launch")
    if cached is not None and cached.key == key:
        return cached
    k = conv.kernel_size[0]
    if not isinstance(conv, nn.Conv2d) or conv.kernel_size[0] != conv.kernel_size[1] or k % 2 == 0 or k > 7 \
            or tuple(conv.padding) != (k // 2, k // 2) or tuple(conv.dilation) != (1, 1) or conv.groups != 1 \
            or conv.stride[0] != conv.stride[1] or conv.stride[0] not in (1, 2) or conv.padding_mode != "zeros":
        raise NotImplementedError(f"HIP conv2d supports odd k <= 7, padding k//2, stride 1|2, dense; got {conv}")
    if getattr(blk, "out_pad", 0) != 0:
        raise NotImplementedError("BaseConvBlk2d out_pad != 0")
    w = conv.weight.detach()
    if not w.is_cuda:
        raise RuntimeError("mvs_gi_amd modules run on the GPU only: call .cuda() on the model "
                           "(there is no CPU fallback)")
    w = w.float().contiguous()
    cout = w.shape[0]
    if isinstance(norm, nn.BatchNorm2d):
        if norm.training:
            raise RuntimeError("HIP path implements eval-mode BatchNorm2d only: call model.eval()")
        gamma = norm.weight.detach().float() if norm.weight is not None else torch.ones(cout, device=w.device)
        beta = norm.bias.detach().float() if norm.bias is not None else torch.zeros(cout, device=w.device)
        alpha = gamma / torch.sqrt(norm.running_var.detach().float() + norm.eps)
        scale, shift = alpha, beta - norm.running_mean.detach().float() * alpha
        if conv.bias is not None:
            shift = shift + conv.bias.detach().float() * alpha
    elif _is_identity(norm):
        scale = torch.ones(cout, device=w.device)
        shift = conv.bias.detach().float().clone() if conv.bias is not None else torch.zeros(cout, device=w.device)
    else:
        raise NotImplementedError(f"norm layer {type(norm).__name__} has no HIP implementation")
    act = blk.activation
    if isinstance(act, nn.LeakyReLU):
        slope = float(act.negative_slope)
    elif isinstance(act, nn.ReLU):
        slope = 0.0
    elif _is_identity(act):
        slope = 1.0
    else:
        raise NotImplementedError(f"activation {type(act).__name__} has no HIP implementation")
    L = Conv2dLaunch()
    L.w, L.wp_b3, L.wp_f32, L.wp_stem, L.wp_rs = w, None, None, None, None
    L.scale, L.shift = scale.contiguous(), shift.contiguous()
    L.stride, L.neg_slope, L.k = int(conv.stride[0]), slope, int(k)
    L.cin, L.cout, L.key = int(w.shape[1]), int(cout), key
    blk.__dict__["_mvsgi_launch"] = L
    return L


def _nhwc(x: Tensor) -> Tensor:
    """[B, C, H, W] in either memory format -> contiguous [B, H, W, C] storage."""
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    lib_x = H._dev(x, "x")
    B, C, Hh, W = lib_x.shape
    from .. import _lib
    y = torch.empty((B, Hh, W, C), device=lib_x.device, dtype=torch.float32)
    _lib.check(_lib.load().mvsgi_ncv_to_nvc_f32(lib_x.data_ptr(), y.data_ptr(), B, C, Hh * W, H._stream_ptr(lib_x)),
               "mvsgi_ncv_to_nvc_f32")
    return y


def _nchw_view(y_nhwc: Tensor) -> Tensor:
    return y_nhwc.permute(0, 3, 1, 2)


def _calc(in_size, k, stride, pad):
    return tuple((s + 2 * pad - k) // stride + 1 for s in in_size)


class BaseConvBlk2d(nn.Module):
    __getstate__ = module_getstate

    def __init__(self, in_chs: int, out_chs: int, kernel_size: int, stride: int = 1, out_pad: int = 0,
                 extra_pad: int = 0, bias_on: bool = False, norm_layer: nn.Module = NoOp(),
                 activation: nn.Module = NoOp()):
        super().__init__()
        self.kernel_size, self.stride, self.out_pad, self.extra_pad = kernel_size, stride, out_pad, extra_pad
        self.conv_layer = nn.Conv2d(in_chs, out_chs, kernel_size, padding=(kernel_size // 2) + extra_pad,
                                    bias=bias_on, stride=stride)
        self.norm_layer = norm_layer
        self.activation = activation
        self.pad_layer = nn.ZeroPad2d((out_pad,) * 4) if out_pad > 0 else NoOp()

    def forward(self, x: Tensor, res: Optional[Tensor] = None) -> Tensor:
        r = None if res is None else _nhwc(res)
        return _nchw_view(lower_conv2d_block(self).run(_nhwc(x), r))

    def infer_size(self, in_size: Tuple[int, int]) -> Tuple[int, int]:
        h, w = _calc(in_size, self.kernel_size, self.stride, self.kernel_size // 2 + self.extra_pad)
        return h + 2 * self.out_pad, w + 2 * self.out_pad


_FUSE_RESBLOCK = H.exp_env("MVSGI_FUSE_RESBLOCK", "1") != "0"


def res_block2d_nhwc(blk, x: Tensor) -> Tensor:
    if not _is_identity(blk.one_by_one) or getattr(blk, "out_pad", 0) != 0:
        raise NotImplementedError("ResConvBlk2d with projection / out_pad is not on the extractor path")
    L1, L2 = lower_conv2d_block(blk.blk1), lower_conv2d_block(blk.blk2)
    if _FUSE_RESBLOCK and H.split_mode() and L1.k == 3 and L2.k == 3 and L1.stride == 1 and L2.stride == 1 \
            and (L1.cin, L1.cout, L2.cin, L2.cout) == (16, 16, 16, 16) and L1.neg_slope == L2.neg_slope:
        # both convs in one launch, the intermediate stays in LDS (mvsgi_resblock2d_f32)
        for L in (L1, L2):
            if L.wp_b3 is None:
                L.wp_b3 = H.pack_conv2d_weights_bf16x3(L.w)
        return H.resblock2d(x, L1.wp_b3, L1.scale, L1.shift, L2.wp_b3, L2.scale, L2.shift, L1.neg_slope)
    r = L1.run(x)
    return L2.run(r, res=x)


class ResConvBlk2d(nn.Module):
    def __init__(self, in_chs: int, out_chs: int, kernel_size: int = 3, in_stride: int = 1, out_stride: int = 1,
                 out_pad: int = 0, activation: nn.Module = NoOp(), norm_layer: nn.Module = NoOp()):
        super().__init__()
        self.in_chs, self.out_chs, self.k_sz = in_chs, out_chs, kernel_size
        self.in_stride, self.out_stride, self.out_pad = in_stride, out_stride, out_pad
        self.blk1 = BaseConvBlk2d(in_chs, out_chs, kernel_size, stride=in_stride,
                                  activation=copy.deepcopy(activation), norm_layer=copy.deepcopy(norm_layer))
        self.blk2 = BaseConvBlk2d(out_chs, out_chs, kernel_size, stride=out_stride,
                                  activation=copy.deepcopy(activation), norm_layer=copy.deepcopy(norm_layer))
        if in_chs != out_chs:
            self.one_by_one = BaseConvBlk2d(in_chs, out_chs, 1, stride=out_stride * in_stride,
                                            activation=copy.deepcopy(activation),
                                            norm_layer=copy.deepcopy(norm_layer))
        else:
            self.one_by_one = NoOp()
        self.pad_layer = nn.ZeroPad2d((out_pad,) * 4) if out_pad > 0 else NoOp()

    def forward(self, x: Tensor) -> Tensor:
        return _nchw_view(res_block2d_nhwc(self, _nhwc(x)))

    def infer_size(self, in_size):
        size = self.blk1.infer_size(in_size)
        size = self.one_by_one.infer_size(size)
        size = self.blk2.infer_size(size)
        return size[0] + 2 * self.out_pad, size[1] + 2 * self.out_pad


def _fusable_resblock(blk) -> bool:
    if not hasattr(blk, "blk1") or not _is_identity(blk.one_by_one) or getattr(blk, "out_pad", 0) != 0:
        return False
    L1, L2 = lower_conv2d_block(blk.blk1), lower_conv2d_block(blk.blk2)
    return L1.k == 3 and L2.k == 3 and L1.stride == 1 and L2.stride == 1 and \
        (L1.cin, L1.cout, L2.cin, L2.cout) == (16, 16, 16, 16) and L1.neg_slope == L2.neg_slope and 0.0 <= L1.neg_slope <= 1.0


def _split2d_pair(self, N: int, Hh: int, W: int, device):
    """Two zero-bordered 2-D split-padded buffers per (image count, resolution), owned by the module and NEVER replaced or freed
    while it lives: the kernels write interiors only, so the borders stay zero from one forward to the next, and a captured
    hipGraph (InferencePipeline.capture) holds their addresses."""
    cache = self.__dict__.setdefault("_mvsgi_split2d", {})
    key = (N, Hh, W, str(device))
    if key not in cache:
        cache[key] = (H.split2d_buffer(N, Hh, W, device), H.split2d_buffer(N, Hh, W, device))
    return cache[key]


def _s2_split_ok(L) -> bool:
    return L.k == 3 and L.stride == 2 and (L.cin, L.cout) == (16, 16) and 0.0 <= L.neg_slope <= 1.0


def _split_chain_forward(self, xin: Tensor, with_final: bool = True) -> Optional[Tensor]:
    """The extractor with its residual blocks on pre-split activations (csrc/resblock2d_rs.hip): the layer in front of a run of
    residual blocks writes the 2-D split-padded format, the blocks hand it on, a stride-2 layer between two runs reads and
    writes it (mvsgi_conv2d_s2_split), the last block in front of any other layer writes fp32.  None when this mode / recipe
    has no such run."""
    if not (_SPLIT_CHAIN and H.split_mode()):
        return None
    layers = [self.first] + list(self.blks) + ([self.final_layer] if with_final else [])
    fus = [hasattr(m, "blk1") and _fusable_resblock(m) for m in layers]

    def run_end(i):                                    # first index behind the run of fusable blocks that starts at i
        while i < len(layers) and fus[i]:
            i += 1
        return i
    y, nchw, idx = xin, True, 0                        # y: fp32 tensor (channels-last unless nchw) or a split-padded buffer
    y_is_split = False
    while idx < len(layers):
        m = layers[idx]
        if hasattr(m, "blk1"):                       # a residual block with no split-writing layer in front of it
            y = res_block2d_nhwc(m, y)
            idx += 1
            continue
        L = lower_conv2d_block(m)
        j = run_end(idx + 1)
        if y_is_split:                               # the stride-2 layer between two runs
            N, Hin, Win = y.shape[0], y.shape[1] - 4, y.shape[2] - 4
            can = True
        elif nchw:
            N, Hin, Win = (y.shape[0], y.shape[1], y.shape[2]) if y.dtype == torch.uint8 else (y.shape[0], y.shape[2], y.shape[3])
            can = (L.k, L.stride, L.cin, L.cout) == (5, 2, 3, 16) and N < 65536
        else:
            N, Hin, Win = y.shape[0], y.shape[1], y.shape[2]
            can = L.k == 3 and L.cin % 16 == 0 and L.cout == 16
        if j > idx + 1 and can:
            Ho, Wo = _calc((Hin, Win), L.k, L.stride, L.k // 2)
            cur, other = _split2d_pair(self, N, Ho, Wo, y.device)
            if y_is_split:
                cur = H.conv2d_s2_split(y, L.rs_weights(), L.shift, cur, L.neg_slope)
            else:
                cur = L.run(y, in_nchw=nchw, out_split=cur)
            # the run's last block hands on split activations when a stride-2 16 -> 16 layer with another run behind it follows
            nxt_split = j < len(layers) and not hasattr(layers[j], "blk1") and _s2_split_ok(lower_conv2d_block(layers[j])) \
                and run_end(j + 1) > j + 1
            for t in range(idx + 1, j):
                L1, L2 = lower_conv2d_block(layers[t].blk1), lower_conv2d_block(layers[t].blk2)
                last = t == j - 1
                out = H.resblock2d_split(cur, L1.rs_weights(), L1.shift, L2.rs_weights(), L2.shift, L1.neg_slope,
                                         out_split=None if (last and not nxt_split) else other)
                cur, other = out, cur
            y, idx, y_is_split = cur, j, nxt_split
        else:
            assert not y_is_split
            y = L.run(y, in_nchw=nchw)
            idx += 1
        nchw = False
    return y                                         # channels-last fp32


def extractor_forward(self, x: Tensor) -> Tensor:
    """simple_feature_extractor.py:81-84.  The RGB stem reads the caller's NCHW fp32 images directly, or
    uint8 [M, H, W, 3] camera images (converted /255 in the kernel, api/inference_class.py:104-107)."""
    xin = x if x.dtype == torch.uint8 else H._dev(x, "imgs")
    out = _split_chain_forward(self, xin)
    if out is not None:
        return _nchw_view(out)
    y = lower_conv2d_block(self.first).run(xin, in_nchw=True)
    for blk in self.blks:
        if hasattr(blk, "blk1"):
            y = res_block2d_nhwc(blk, y)
        else:
            y = lower_conv2d_block(blk).run(y)
    return _nchw_view(lower_conv2d_block(self.final_layer).run(y))


class SimpleFeatExtraction(nn.Module):
    __getstate__ = module_getstate

    def __init__(self, in_size: Tuple[int, int], in_chs=3, chs: int = 8, k_sz: int = 3,
                 layers: Sequence[int] = [5, 10], norm_type: str = "batch", relu_type: str = "leaky"):
        super().__init__()
        self.chs, self.k_sz, self.layers = chs, k_sz, layers
        self.num_steps = len(layers)
        self.in_size = in_size
        self.norm_type = NORM2D_TYPE[norm_type]
        self.relu_type = RELU_TYPE[relu_type]
        self.first = BaseConvBlk2d(in_chs=in_chs, out_chs=chs, kernel_size=5, stride=2,
                                   activation=self.relu_type(), norm_layer=self.norm_type(chs))
        blks = []
        for step, n in enumerate(layers):
            blks += [ResConvBlk2d(in_chs=chs, out_chs=chs, kernel_size=k_sz, activation=self.relu_type(),
                                  norm_layer=self.norm_type(chs)) for _ in range(n)]
            if step != self.num_steps - 1:
                blks.append(BaseConvBlk2d(in_chs=chs, out_chs=chs, kernel_size=3, stride=2,
                                          activation=self.relu_type(), norm_layer=self.norm_type(chs)))
        self.blks = nn.Sequential(*blks)
        self.final_layer = BaseConvBlk2d(in_chs=chs, out_chs=chs, kernel_size=k_sz, activation=self.relu_type(),
                                         norm_layer=self.norm_type(chs))

    forward = extractor_forward



# ------------------------------------------------------------------------------------------
# the sphere-convolution final layer of the G16VV extractor (config103)
# ------------------------------------------------------------------------------------------
def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def sphere_conv_offsets(input_size, kernel_size, stride, padding, dilation, lat_range=(-math.pi / 2, 0),
                        lon_range=(0, 2 * math.pi)) -> Tensor:
    """Offset field of SphereConvEquirect2d.gen_offset (common/common_modules.py:427-507): for every output pixel
    of an equirect image, where the taps of a kernel laid on the tangent plane of the sphere fall, as
    deform_conv2d offsets [1, 2*Kh*Kw, Ho, Wo] (channel 2*(i*Kw+j) = row offset, +1 = column offset).
    Same fp32 operation sequence as the reference, evaluated by broadcasting instead of per-latitude loops."""
    height, width = int(input_size[-2]), int(input_size[-1])
    Kh, Kw = kernel_size
    sh, sw = stride
    dh, dw = dilation
    lat_dist, lon_dist = lat_range[1] - lat_range[0], lon_range[1] - lon_range[0]
    lat_center, lon_center = lat_dist / 2 + lat_range[0], lon_dist / 2 + lon_range[0]
    delta_lat, delta_lon = lat_dist / height, lon_dist / width                                 # :438-439
    ry = torch.arange(-(Kh // 2), Kh // 2 + 1)
    rx = torch.arange(-(Kw // 2), Kw // 2 + 1)
    if Kw % 2 == 0:
        rx = rx[:-1]
    if Kh % 2 == 0:
        ry = ry[:-1]
    ker_x = torch.tan(rx * dw * delta_lon)                                                      # :452
    ker_y = torch.tan(ry * dh * delta_lat) / torch.cos(ry * delta_lon)                          # :453 (delta_lon, as upstream)
    ker_y, ker_x = torch.meshgrid(ker_y, ker_x, indexing="ij")                                  # [Kh, Kw]
    rho = torch.sqrt(ker_x ** 2 + ker_y ** 2)
    if Kh % 2 and Kw % 2:
        rho[Kh // 2][Kw // 2] = 1e-8                                                            # :462-463
    nu = torch.arctan(rho)
    cos_nu, sin_nu = torch.cos(nu), torch.sin(nu)
    h_range = torch.arange(0, height, sh) + 0.5
    w_range = torch.arange(0, width, sw) + 0.5
    lat_c = ((h_range / height) - 0.5) * lat_dist + lat_center                                  # [Ho]
    lon_c = ((w_range / width) - 0.5) * lon_dist + lon_center                                   # [Wo]
    sl, cl = torch.sin(lat_c)[:, None, None], torch.cos(lat_c)[:, None, None]
    lat = torch.arcsin(cos_nu * sl + ker_y * sin_nu * cl / rho)                                 # :476  [Ho, Kh, Kw]
    lon = torch.arctan2(ker_x * sin_nu, (rho * cl * cos_nu - ker_y * sl * sin_nu))              # :483
    lat = lat[:, None].expand(-1, lon_c.numel(), -1, -1)                                        # [Ho, Wo, Kh, Kw]
    lon = lon[:, None] + lon_c[None, :, None, None]                                             # :485
    lat = ((lat - lat_center) / lat_dist + 0.5) * height                                        # :489-490 -> pixels
    lon = ((lon - lon_center) / lon_dist + 0.5) * width
    lon = lon % (width - 1)                                                                     # :493 then :495, both as upstream
    lon = lon % width
    lat = lat - (h_range[:, None, None, None] - 0.5)                                            # :499-500 -> offsets
    lon = lon - (w_range[None, :, None, None] - 0.5)
    ll = torch.stack((lat, lon)).to(torch.float)                                                # [2, Ho, Wo, Kh, Kw]
    ll = ll.permute(3, 4, 0, 1, 2)                                                              # [Kh, Kw, 2, Ho, Wo]
    return ll.reshape(1, 2 * Kh * Kw, ll.shape[3], ll.shape[4])


class SphereConvEquirect2d(nn.Module):
    """common/common_modules.py:360-425: Conv2d whose taps follow the sphere; parameters `weight`
    [Cout, Cin/groups, Kh, Kw], optional `bias`, buffer `offset` (state-dict compatible)."""
    __getstate__ = module_getstate

    def __init__(self, in_size, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True):
        super().__init__()
        if in_channels % groups != 0:
            raise ValueError("in_channels must be divisible by groups")
        if out_channels % groups != 0:
            raise ValueError("out_channels must be divisible by groups")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation, self.groups = _pair(padding), _pair(dilation), groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            nn.init.uniform_(self.bias, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
        self.register_buffer("offset", sphere_conv_offsets(in_size, self.kernel_size, self.stride, self.padding,
                                                           self.dilation))

    def forward(self, input: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        return _nchw_view(sphere_conv_nhwc(self, _nhwc(input), mask=mask))


def _lower_sphere(conv, norm=None, act=None):
    """Launch record of a SphereConvEquirect2d (+ optional eval BatchNorm2d / activation), cached on the module."""
    parts = [conv.weight.data_ptr(), conv.weight._version, id(norm), id(act), type(act).__name__,
             float(getattr(act, "negative_slope", 0.0))]
    if conv.bias is not None:
        parts += [conv.bias.data_ptr(), conv.bias._version]
    if isinstance(norm, nn.BatchNorm2d):
        parts += [bool(norm.training), float(norm.eps)]
        for t in (norm.weight, norm.bias, norm.running_mean, norm.running_var):
            if t is not None:
                parts += [t.data_ptr(), t._version]
    key = tuple(parts)
    cached = conv.__dict__.get("_mvsgi_launch")
    if cached is not None and cached[0] == key:
        return cached[1]
    if getattr(conv, "groups", 1) != 1:
        raise NotImplementedError("SphereConvEquirect2d with groups != 1 has no HIP implementation")
    w = conv.weight.detach()
    if not w.is_cuda:
        raise RuntimeError("mvs_gi_amd modules run on the GPU only: call .cuda() on the model (there is no CPU fallback)")
    w = w.float().contiguous()
    cout = w.shape[0]
    bias = conv.bias.detach().float() if conv.bias is not None else None
    if isinstance(norm, nn.BatchNorm2d):
        if norm.training:
            raise RuntimeError("HIP path implements eval-mode BatchNorm2d only: call model.eval()")
        gamma = norm.weight.detach().float() if norm.weight is not None else torch.ones(cout, device=w.device)
        beta = norm.bias.detach().float() if norm.bias is not None else torch.zeros(cout, device=w.device)
        alpha = gamma / torch.sqrt(norm.running_var.detach().float() + norm.eps)
        scale, shift = alpha, beta - norm.running_mean.detach().float() * alpha
        if bias is not None:
            shift = shift + bias * alpha
    elif _is_identity(norm) or norm is None:
        scale = torch.ones(cout, device=w.device)
        shift = bias.clone() if bias is not None else torch.zeros(cout, device=w.device)
    else:
        raise NotImplementedError(f"norm layer {type(norm).__name__} has no HIP implementation")
    if isinstance(act, nn.LeakyReLU):
        slope = float(act.negative_slope)
    elif isinstance(act, nn.ReLU):
        slope = 0.0
    elif _is_identity(act) or act is None:
        slope = 1.0
    else:
        raise NotImplementedError(f"activation {type(act).__name__} has no HIP implementation")
    rec = dict(wp=H.pack_deform_conv2d_weights(w), scale=scale.contiguous(), shift=shift.contiguous(), slope=slope)
    conv.__dict__["_mvsgi_launch"] = (key, rec)
    return rec


def sphere_conv_nhwc(conv, x: Tensor, norm=None, act=None, res: Optional[Tensor] = None, mask=None) -> Tensor:
    if mask is not None:
        raise NotImplementedError("modulated deform_conv2d (mask) is not on the extractor path")
    rec = _lower_sphere(conv, norm, act)
    return H.deform_conv2d(x, conv.offset.float(), rec["wp"], rec["scale"], rec["shift"], _pair(conv.kernel_size),
                           _pair(conv.stride), _pair(conv.padding), _pair(conv.dilation), res=res,
                           neg_slope=rec["slope"])


def sphere_blk_nhwc(blk, x: Tensor, res: Optional[Tensor] = None) -> Tensor:
    """SphereConvBlk.forward (:538-547): blk = Sequential(SphereConvEquirect2d, norm); + res; activation."""
    return sphere_conv_nhwc(blk.blk[0], x, norm=blk.blk[1], act=blk.activation, res=res)


class SphereConvBlk(nn.Module):
    def __init__(self, in_size, in_chs: int, out_chs: int, k_sz: int, stride: int = 1, extra_pad: int = 0,
                 bias: bool = False, dilation: int = 1, norm_layer: nn.Module = NoOp(), activation: nn.Module = NoOp()):
        super().__init__()
        self.activation = activation
        self.blk = nn.Sequential(
            SphereConvEquirect2d(in_size=in_size, in_channels=in_chs, out_channels=out_chs, kernel_size=k_sz,
                                 stride=stride, padding=k_sz // 2 + extra_pad, dilation=dilation, bias=bias),
            norm_layer)

    def forward(self, x: Tensor, res: Optional[Tensor] = None) -> Tensor:
        r = None if res is None else _nhwc(res)
        return _nchw_view(sphere_blk_nhwc(self, _nhwc(x), r))


def sphere_extractor_forward(self, x: Tensor) -> Tensor:
    """SphereEquirectFeatExtraction.forward (feature_extractor/sphere_feature_extractor.py:80-83)."""
    xin = x if x.dtype == torch.uint8 else H._dev(x, "imgs")
    y = _split_chain_forward(self, xin, with_final=False)
    if y is None:
        y = lower_conv2d_block(self.first).run(xin, in_nchw=True)
        for blk in self.blks:
            if hasattr(blk, "blk1"):
                y = res_block2d_nhwc(blk, y)
            else:
                y = lower_conv2d_block(blk).run(y)
    return _nchw_view(sphere_blk_nhwc(self.final_layer, y))


class SphereEquirectFeatExtraction(nn.Module):
    __getstate__ = module_getstate

    """feature_extractor/sphere_feature_extractor.py:8-83: SimpleFeatExtraction whose final layer is a SphereConvBlk."""

    def __init__(self, in_size: Tuple[int, int], in_chs=3, chs: int = 8, k_sz: int = 3,
                 layers: Sequence[int] = [5, 10], norm_type: str = "batch", relu_type: str = "leaky"):
        super().__init__()
        self.chs, self.k_sz, self.layers = chs, k_sz, layers
        self.num_steps = len(layers)
        self.in_size = in_size
        self.norm_type = NORM2D_TYPE[norm_type]
        self.relu_type = RELU_TYPE[relu_type]
        self.first = BaseConvBlk2d(in_chs=in_chs, out_chs=chs, kernel_size=5, stride=2,
                                   activation=self.relu_type(), norm_layer=self.norm_type(chs))
        blks = []
        for step, n in enumerate(layers):
            blks += [ResConvBlk2d(in_chs=chs, out_chs=chs, kernel_size=k_sz, activation=self.relu_type(),
                                  norm_layer=self.norm_type(chs)) for _ in range(n)]
            if step != self.num_steps - 1:
                blks.append(BaseConvBlk2d(in_chs=chs, out_chs=chs, kernel_size=3, stride=2,
                                          activation=self.relu_type(), norm_layer=self.norm_type(chs)))
        self.blks = nn.Sequential(*blks)
        size = self.first.infer_size(self.in_size)
        for layer in self.blks:
            size = layer.infer_size(size)
        self.final_layer = SphereConvBlk(in_size=size, in_chs=chs, out_chs=chs, k_sz=k_sz,
                                         activation=self.relu_type(), norm_layer=self.norm_type(chs))

    forward = sphere_extractor_forward
