"""Drop-in 2-D blocks and feature extractor: BaseConvBlk2d
(dsta_mvs/model/common/common_modules.py:18-80), ResConvBlk2d (:117-184) and SimpleFeatExtraction
(dsta_mvs/model/feature_extractor/simple_feature_extractor.py:8-84) with the reference's constructor
arguments, child names and state-dict keys; forward runs mvsgi_conv2d_f32.

forward(imgs [B*N, 3, H, W]) -> feats [B*N, chs, H/4, W/4], returned with the reference's shape and
channels-last strides (storage [B*N, H/4, W/4, chs]), which is exactly what the sweep kernel wants.
"""
from __future__ import annotations

import copy
from typing import Optional, Sequence, Tuple

import torch
from torch import nn, Tensor

from .. import hip_ops as H
from .common_modules import NoOp, NORM2D_TYPE, RELU_TYPE, _is_identity


class Conv2dLaunch:
    __slots__ = ("w", "wp_b3", "wp_f32", "scale", "shift", "stride", "neg_slope", "k", "cin", "cout", "key")

    def run(self, x: Tensor, res: Optional[Tensor] = None, in_nchw: bool = False) -> Tensor:
        impl, wp = H.CONV_AUTO, None
        if H.get_conv_mode() == "bf16x3" and self.k == 3 and self.cin % 16 == 0 and self.cout % 16 == 0 and not in_nchw:
            if self.wp_b3 is None:
                self.wp_b3 = H.pack_conv2d_weights_bf16x3(self.w)
            impl, wp = H.CONV_BF16X3, self.wp_b3
        elif self.k == 3 and self.cin % 16 == 0 and self.cout in (16, 32) and not in_nchw:
            if self.wp_f32 is None:                    # exact fp32 mode: fp32 MFMA kernel
                self.wp_f32 = H.pack_conv2d_weights_f32(self.w)
            impl, wp = H.CONV_MFMA, self.wp_f32
        return H.conv2d(x, self.w, wp, self.scale, self.shift, res=res, stride=self.stride,
                        neg_slope=self.neg_slope, impl=impl, in_nchw=in_nchw)


def lower_conv2d_block(blk) -> Conv2dLaunch:
    conv: nn.Conv2d = blk.conv_layer
    parts = [conv.weight.data_ptr(), conv.weight._version]
    norm = blk.norm_layer
    if isinstance(norm, nn.BatchNorm2d):
        parts += [t.data_ptr() + t._version for t in (norm.weight, norm.bias, norm.running_mean, norm.running_var)
                  if t is not None]
    key = tuple(parts)
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
    L.w, L.wp_b3, L.wp_f32 = w, None, None
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


def res_block2d_nhwc(blk, x: Tensor) -> Tensor:
    if not _is_identity(blk.one_by_one) or getattr(blk, "out_pad", 0) != 0:
        raise NotImplementedError("ResConvBlk2d with projection / out_pad is not on the extractor path")
    r = lower_conv2d_block(blk.blk1).run(x)
    return lower_conv2d_block(blk.blk2).run(r, res=x)


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


def extractor_forward(self, x: Tensor) -> Tensor:
    """simple_feature_extractor.py:81-84.  The RGB stem reads the caller's NCHW fp32 images directly, or
    uint8 [M, H, W, 3] camera images (converted /255 in the kernel, api/inference_class.py:104-107)."""
    xin = x if x.dtype == torch.uint8 else H._dev(x, "imgs")
    y = lower_conv2d_block(self.first).run(xin, in_nchw=True)
    for blk in self.blks:
        if hasattr(blk, "blk1"):
            y = res_block2d_nhwc(blk, y)
        else:
            y = lower_conv2d_block(blk).run(y)
    return _nchw_view(lower_conv2d_block(self.final_layer).run(y))


class SimpleFeatExtraction(nn.Module):
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
