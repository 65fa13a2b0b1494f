"""Host-side mirror of the 3-D building blocks of dsta_mvs/model/common/common_modules.py
(NoOp :13-16, BaseConvBlk3d :82-115, ResConvBlk3d :186-244, ResizeConv3d :305-355) and of
the registries in dsta_mvs/model/common/__init__.py:7-23.

The classes keep the reference's constructor arguments, attribute names and child-module
names, so state dicts load unchanged and module objects pickled by the reference
(Lightning `save_hyperparameters()`, spherical_sweep_stereo.py:74) unpickle into them.
Their `forward` runs the hand-written HIP kernels through the C ABI; nothing here
computes on the CPU.
"""
from __future__ import annotations

import os

import copy
from typing import Dict, Optional, Type

import torch
from torch import nn, Tensor

from .. import hip_ops as H


def module_getstate(self):
    """__getstate__ of the drop-in modules (and, in patch mode, of the reference's classes): the launch records, packed
    weights, polyphase plans and module-owned activation buffers cached on a module under `_mvsgi_*` keys are derived data
    -- they are rebuilt on the next forward and must not travel with a pickled module or a whole-module checkpoint
    (Lightning's save_hyperparameters() pickles module OBJECTS, spherical_sweep_stereo.py:74)."""
    return {k: v for k, v in self.__dict__.items() if not k.startswith("_mvsgi_")}


class NoOp(nn.Identity):
    """Alias of nn.Identity, as in the reference (common_modules.py:13-16)."""
    def infer_size(self, in_size):
        return in_size


RELU_TYPE: Dict[str, Type[nn.Module]] = {"original": nn.ReLU, "leaky": nn.LeakyReLU, "none": NoOp}
NORM2D_TYPE: Dict[str, Type[nn.Module]] = {"batch": nn.BatchNorm2d, "instance": nn.InstanceNorm2d, "none": NoOp}
NORM3D_TYPE: Dict[str, Type[nn.Module]] = {"batch": nn.BatchNorm3d, "instance": nn.InstanceNorm3d, "none": NoOp}


# ------------------------------------------------------------------------------------------
# lowering of one conv block to the arguments of mvsgi_conv3d_f32
# ------------------------------------------------------------------------------------------
# MVSGI_V32=1: 32x32x16-MFMA kernels for the Cout % 32 == 0 layers (measured on par with the 16x16x32 kernels, so off by default)
_USE_V32 = H.exp_env("MVSGI_V32", "0") != "0"
# MVSGI_D32=0: keep the Cin % 32 == 0 stride-1 layers of large launches on the tap-pair layout (default: 32-channel slices,
# csrc/conv3d_bf16x3.hpp D32 -- 27 k-steps per 32 channels instead of 28 and half the slices per unit: 6-9 % of those layers)
_USE_D32 = os.environ.get("MVSGI_D32", "1") != "0"
_NO_D32U = H.exp_env("MVSGI_NO_D32U", "0") != "0"      # (tools: the fused-upsample layers alone back on tap pairs)
_D32_OK: Dict[tuple, bool] = {}       # (cin, cout, B, D, H, W) -> mvsgi_conv3d_d32_applies; ("up2", ...) -> mvsgi_conv3d_up2_d32_applies


class ConvLaunch:
    """Device-resident launch arguments of one BaseConvBlk3d: PyTorch-layout weight, packed
    MFMA weight (or None), per-channel scale/shift (eval BatchNorm3d or bias), stride, slope."""
    __slots__ = ("w", "wp", "wp_b3", "wp_c16", "wp_v32", "wp_d32", "wp_rs", "wp_s2", "wp_poly", "wp_head", "head_sc", "scale", "shift", "stride",
                 "neg_slope", "cin", "cout", "key", "f16")

    def run(self, x_ndhwc: Tensor, res: Optional[Tensor] = None, impl: Optional[int] = None) -> Tensor:
        wp = self.wp
        if impl is None and H.get_conv_mode() == "f16x3" and self.cin % 16 == 0 and self.cout % 16 == 0:
            B, D, Hh, W, _ = x_ndhwc.shape
            layout = H.CONV_BF16X3_C16 if self._c16() else \
                (H.CONV_BF16X3_V32 if _USE_V32 and self.cout % 32 == 0 and H.conv3d_v32_applies(B, self.cin, D, Hh, W, self.cout, self.stride)
                 else (H.CONV_BF16X3_D32 if self._d32(B, D, Hh, W) else H.CONV_BF16X3))
            wp16, sc16 = self._f16(layout)
            return H.conv3d(x_ndhwc, self.w, wp16, sc16, self.shift, res=res, stride=self.stride, neg_slope=self.neg_slope,
                            impl=layout | H.CONV_F16)
        if impl is None:
            impl = H.CONV_AUTO
            if H.get_conv_mode() == "bf16x3" and self.cin % 16 == 0 and self.cout % 16 == 0:
                B, D, Hh, W, _ = x_ndhwc.shape
                if self._c16():
                    impl, wp = H.CONV_BF16X3_C16, self._wp_c16()
                elif _USE_V32 and self.cout % 32 == 0 and H.conv3d_v32_applies(B, self.cin, D, Hh, W, self.cout, self.stride):
                    impl, wp = H.CONV_BF16X3_V32, self._wp_v32()
                elif self._d32(B, D, Hh, W):
                    if getattr(self, "wp_d32", None) is None:
                        self.wp_d32 = H.pack_conv_weights_bf16x3_d32(self.w)
                    impl, wp = H.CONV_BF16X3_D32, self.wp_d32
                else:
                    if self.wp_b3 is None:
                        self.wp_b3 = H.pack_conv_weights_bf16x3(self.w)
                    impl, wp = H.CONV_BF16X3, self.wp_b3
        elif impl == H.CONV_BF16X3:
            if self.wp_b3 is None:
                self.wp_b3 = H.pack_conv_weights_bf16x3(self.w)
            wp = self.wp_b3
        return H.conv3d(x_ndhwc, self.w, wp, self.scale, self.shift, res=res, stride=self.stride,
                        neg_slope=self.neg_slope, impl=impl)

    def _f16(self, layout: int):
        """(packed weights, per-channel scale) of the fp16 split in `layout`: the weights pre-scaled per output channel by a power
        of two, the epilogue's scale carrying the inverse (H.pack_conv_weights_f16x3)."""
        if self.f16 is None:
            self.f16 = {}
        if layout not in self.f16:
            wp, unscale = H.pack_conv_weights_f16x3(self.w, layout)
            self.f16[layout] = (wp, (self.scale * unscale).contiguous())
        return self.f16[layout]

    def _d32(self, B: int, D: int, Hh: int, W: int) -> bool:
        """32-channel slices (MVSGI_CONV_BF16X3_D32) serve this launch: Cin % 32 == 0, stride 1, a large launch."""
        if not (_USE_D32 and self.stride == 1 and self.cin % 32 == 0 and self.cout % 16 == 0):
            return False
        key = (B, D, Hh, W)                      # (one library query per launch shape, not per launch)
        ok = _D32_OK.get((self.cin, self.cout) + key)
        if ok is None:
            ok = _D32_OK[(self.cin, self.cout) + key] = H.conv3d_d32_applies(B, self.cin, D, Hh, W, self.cout, self.stride)
        return ok

    def _c16(self) -> bool:
        """Cout == 16, stride 1: the plane-schedule kernel (MVSGI_CONV_BF16X3_C16)."""
        return self.cout == 16 and self.stride == 1 and self.cin % 16 == 0

    def _wp_c16(self):
        if getattr(self, "wp_c16", None) is None:
            self.wp_c16 = H.pack_conv_weights_bf16x3_c16(self.w)
        return self.wp_c16

    def _wp_b3(self):
        if self.wp_b3 is None:
            self.wp_b3 = H.pack_conv_weights_bf16x3(self.w)
        return self.wp_b3

    def rs_ok(self) -> bool:
        """The register-stationary kernel (csrc/conv3d_rs.hip) serves this layer."""
        return H.conv3d_rs_applies(self.cin, self.cout, self.stride, self.neg_slope)

    def _wp_rs(self):
        if self.wp_rs is None:
            self.wp_rs = H.pack_conv_weights_rs(self.w)
        return self.wp_rs

    def _rs(self, fmt: str):
        """(packed weights, per-channel scale) of the register-stationary kernels in the split `fmt` of the activations."""
        if fmt == "bf16":
            return self._wp_rs(), self.scale
        if self.f16 is None:
            self.f16 = {}
        if "rs" not in self.f16:
            wp, unscale = H.pack_conv_weights_rs(self.w, "f16")
            self.f16["rs"] = (wp, (self.scale * unscale).contiguous())
        return self.f16["rs"]

    def wino_ok(self, D: int, Hh: int, W: int) -> bool:
        """The Winograd-form kernel (csrc/conv3d_wino.hip) serves this layer on a [D, Hh, W] volume (fp16 split only)."""
        return H.conv3d_wino_applies(self.cin, self.cout, D, Hh, W, self.stride, self.neg_slope)

    def _wino(self):
        """(packed weights, per-channel scale) of the Winograd-form kernel."""
        if self.f16 is None:
            self.f16 = {}
        if "wino" not in self.f16:
            wp, unscale = H.pack_conv_weights_wino(self.w)
            self.f16["wino"] = (wp, (self.scale * unscale).contiguous())
        return self.f16["wino"]

    def _b3(self, fmt: str):
        """(packed weights, per-channel scale) of the streaming split kernel (generic layout) in the split `fmt`."""
        return (self._wp_b3(), self.scale) if fmt == "bf16" else self._f16(H.CONV_BF16X3)

    def s2rs_ok(self) -> bool:
        """The stride-2 16 -> 32 kernel on split-padded activations (csrc/conv3d_s2rs.hip) serves this layer."""
        return H.split_mode() and H.conv3d_s2rs_applies(self.cin, self.cout, self.stride, self.neg_slope)

    def _wp_s2(self):
        if self.wp_s2 is None:
            self.wp_s2 = H.pack_conv_weights_s2rs(self.w, self.scale)
        return self.wp_s2

    def _s2(self, fmt: str):
        """(packed weights with the scale folded in, shift, unscale) of the stride-2 kernel on split-padded activations."""
        if fmt == "bf16":
            return self._wp_s2(), self.shift, 1.0
        if self.f16 is None:
            self.f16 = {}
        if "s2" not in self.f16:
            wp, up, un = H.pack_conv_weights_s2rs(self.w, self.scale, "f16")
            self.f16["s2"] = (wp, (self.shift * up).contiguous(), un)
        return self.f16["s2"]

    def _wp_v32(self):
        if self.wp_v32 is None:
            self.wp_v32 = H.pack_conv_weights_bf16x3_v32(self.w)
        return self.wp_v32

    def poly_ok(self) -> bool:
        """ResizeConv3d in polyphase form on the register-stationary kernel (csrc/conv3d_up2poly.hip)."""
        return H.split_mode() and self.stride == 1 and H.conv3d_up2_poly_applies(self.cin, self.cout, self.neg_slope)

    def _poly_plan(self, D: int, Hh: int, W: int, fmt: str = "bf16"):
        """(folded phase weights + face tables for a low-resolution input of D x Hh x W in the split `fmt`, the layer's scale):
        built once per size and split, kept."""
        if self.wp_poly is None:
            self.wp_poly = {}
        k = (int(D), int(Hh), int(W), fmt)
        if k not in self.wp_poly:
            if fmt == "f16":
                plan, unscale = H.conv3d_up2_poly_plan(self.w, *k[:3], fmt="f16")
                self.wp_poly[k] = (plan, (self.scale * unscale).contiguous())
            else:
                self.wp_poly[k] = (H.conv3d_up2_poly_plan(self.w, *k[:3]), self.scale)
        return self.wp_poly[k]

    def head_split_ok(self) -> bool:
        """The split cost head on a split-padded input (csrc/conv3d_headsplit.hip), in either 16-bit split."""
        return H.split_mode() and self.cout == 1 and self.cin % 16 == 0 and self.stride == 1

    def run_head_split(self, x_split) -> Tensor:
        if x_split.fmt == "f16":
            if self.f16 is None:
                self.f16 = {}
            if "head" not in self.f16:
                wp, unscale = H.pack_head_split_weights_f16(self.w)
                self.f16["head"] = (wp, float(self.scale[0]) * unscale, float(self.shift[0]))      # one host read at lowering time
            wp, sc, sh = self.f16["head"]
            return H.conv3d_head_split(x_split, wp, sc, sh, neg_slope=self.neg_slope, f16=True)
        if self.wp_head is None:
            self.wp_head = H.pack_head_split_weights(self.w)
            self.head_sc = (float(self.scale[0]), float(self.shift[0]))       # one host read at lowering time
        return H.conv3d_head_split(x_split, self.wp_head, self.head_sc[0], self.head_sc[1], neg_slope=self.neg_slope)

    def run_up2_poly_split(self, x_split, out) -> "H.SplitAct":
        plan, sc = self._poly_plan(x_split.D, x_split.H, x_split.W, x_split.fmt)
        return H.conv3d_up2_poly_split(x_split, plan, sc, self.shift, out=out, neg_slope=self.neg_slope)

    def run_up2_poly(self, x_split, out=None) -> Tensor:
        plan, sc = self._poly_plan(x_split.D, x_split.H, x_split.W, x_split.fmt)
        return H.conv3d_up2_poly(x_split, plan, sc, self.shift, neg_slope=self.neg_slope, out=out)

    def run_up2_split(self, x_lowres_ndhwc: Tensor, res: Optional[Tensor], out) -> "H.SplitAct":
        """conv(trilinear_x2(x)) (+ res) written split-padded into `out` (the polyphase layer's input, the split head's input)."""
        if H.get_conv_mode() == "f16x3":
            layout = H.CONV_BF16X3_C16 if self._c16() else H.CONV_BF16X3
            wp16, sc16 = self._f16(layout)
            return H.conv3d_up2_out_split(x_lowres_ndhwc, wp16, sc16, self.shift, out=out, res=res, neg_slope=self.neg_slope,
                                          w_layout=layout | H.CONV_F16)
        if self._c16():       # Cout == 16: the plane schedule (one cout tile)
            return H.conv3d_up2_out_split(x_lowres_ndhwc, self._wp_c16(), self.scale, self.shift, out=out, res=res,
                                          neg_slope=self.neg_slope, w_layout=H.CONV_BF16X3_C16)
        return H.conv3d_up2_out_split(x_lowres_ndhwc, self._wp_b3(), self.scale, self.shift, out=out, res=res, neg_slope=self.neg_slope)

    def can_fuse_up2(self) -> bool:
        return H.split_mode() and self.stride == 1 and self.cin % 16 == 0 and self.cout % 16 == 0

    def run_up2(self, x_lowres_ndhwc: Tensor, res: Optional[Tensor] = None) -> Tensor:
        """conv(trilinear_x2(x)) in one launch (mvsgi_conv3d_up2_f32)."""
        B, Dl, Hl, Wl, _ = x_lowres_ndhwc.shape
        d32 = False
        if _USE_D32 and not _NO_D32U and not self._c16() and self.cin % 32 == 0:
            key = ("up2", self.cin, self.cout, B, Dl, Hl, Wl)
            d32 = _D32_OK.get(key)
            if d32 is None:
                d32 = _D32_OK[key] = H.conv3d_up2_d32_applies(B, self.cin, Dl, Hl, Wl, self.cout)
        if H.get_conv_mode() == "f16x3":
            layout = H.CONV_BF16X3_C16 if self._c16() else (H.CONV_BF16X3_D32 if d32 else H.CONV_BF16X3)
            wp16, sc16 = self._f16(layout)
            return H.conv3d_up2(x_lowres_ndhwc, wp16, sc16, self.shift, res=res, neg_slope=self.neg_slope, w_layout=layout | H.CONV_F16)
        if self._c16():
            return H.conv3d_up2(x_lowres_ndhwc, self._wp_c16(), self.scale, self.shift, res=res,
                                neg_slope=self.neg_slope, w_layout=H.CONV_BF16X3_C16)
        if d32 and H.get_conv_mode() == "bf16x3":
            if getattr(self, "wp_d32", None) is None:
                self.wp_d32 = H.pack_conv_weights_bf16x3_d32(self.w)
            return H.conv3d_up2(x_lowres_ndhwc, self.wp_d32, self.scale, self.shift, res=res, neg_slope=self.neg_slope, w_layout=H.CONV_BF16X3_D32)
        if _USE_V32 and self.cout % 32 == 0 and H.conv3d_v32_applies(B, self.cin, 2 * Dl, 2 * Hl, 2 * Wl, self.cout, 1):
            return H.conv3d_up2(x_lowres_ndhwc, self._wp_v32(), self.scale, self.shift, res=res,
                                neg_slope=self.neg_slope, w_layout=H.CONV_BF16X3_V32)
        if self.wp_b3 is None:
            self.wp_b3 = H.pack_conv_weights_bf16x3(self.w)
        return H.conv3d_up2(x_lowres_ndhwc, self.wp_b3, self.scale, self.shift, res=res, neg_slope=self.neg_slope)


def _is_identity(m) -> bool:
    return m is None or isinstance(m, nn.Identity)


def _fingerprint(blk) -> tuple:
    """Everything the lowered launch record depends on: parameter identity and in-place version (pointer and
    version as SEPARATE entries), the norm layer's mode / eps, the activation and the stride.  `training` is part
    of the key, so a later `model.train()` re-lowers and raises instead of silently reusing eval statistics."""
    conv = blk.conv_layer
    parts = [conv.weight.data_ptr(), conv.weight._version, tuple(conv.stride)]
    if conv.bias is not None:
        parts += [conv.bias.data_ptr(), conv.bias._version]
    norm = blk.norm_layer
    parts.append(type(norm).__name__)
    if isinstance(norm, (nn.BatchNorm3d, nn.BatchNorm2d)):
        parts += [bool(norm.training), float(norm.eps)]
        for t in (norm.weight, norm.bias, norm.running_mean, norm.running_var):
            if t is not None:
                parts += [t.data_ptr(), t._version]
    act = blk.activation
    parts += [type(act).__name__, float(getattr(act, "negative_slope", 0.0))]
    return tuple(parts)


def lower_conv_block(blk) -> ConvLaunch:
    """Build (and cache on the module, keyed by parameter identity/version) the launch
    arguments of a BaseConvBlk3d-shaped module {conv_layer, norm_layer, activation}."""
    key = _fingerprint(blk)
    cached = blk.__dict__.get("_mvsgi_launch")
    if cached is not None and cached.key == key:
        return cached
    conv: nn.Conv3d = blk.conv_layer
    if not isinstance(conv, nn.Conv3d):
        raise NotImplementedError(f"conv_layer is {type(conv).__name__}, expected nn.Conv3d")
    if tuple(conv.kernel_size) != (3, 3, 3) or tuple(conv.padding) != (1, 1, 1) or tuple(conv.dilation) != (1, 1, 1) \
            or conv.groups != 1 or conv.padding_mode != "zeros" or len(set(conv.stride)) != 1 \
            or conv.stride[0] not in (1, 2):
        raise NotImplementedError(
            f"HIP conv3d supports kernel 3, padding 1, stride 1|2, dense; got kernel {tuple(conv.kernel_size)}, "
            f"padding {tuple(conv.padding)}, stride {tuple(conv.stride)}, groups {conv.groups}")
    w = conv.weight.detach()
    if not w.is_cuda:
        raise RuntimeError("mvs_gi_amd modules run on the GPU only: call .cuda() on the model "
                           "(there is no CPU fallback)")
    w = w.to(torch.float32).contiguous()
    cout = w.shape[0]
    norm = blk.norm_layer
    if isinstance(norm, nn.BatchNorm3d):
        if norm.training:
            raise RuntimeError("HIP path implements eval-mode BatchNorm3d only: call model.eval()")
        if norm.running_mean is None or norm.running_var is None:
            raise NotImplementedError("BatchNorm3d without running statistics")
        gamma = norm.weight.detach().float() if norm.weight is not None else torch.ones(cout, device=w.device)
        beta = norm.bias.detach().float() if norm.bias is not None else torch.zeros(cout, device=w.device)
        # ATen eval batch_norm: alpha = gamma / sqrt(var + eps); y = x * alpha + (beta - mean * alpha)
        alpha = gamma / torch.sqrt(norm.running_var.detach().float() + norm.eps)
        scale = alpha
        shift = beta - norm.running_mean.detach().float() * alpha
        if conv.bias is not None:
            shift = shift + conv.bias.detach().float() * alpha
    elif _is_identity(norm):
        scale = torch.ones(cout, device=w.device, dtype=torch.float32)
        shift = conv.bias.detach().float().clone() if conv.bias is not None \
            else torch.zeros(cout, device=w.device, dtype=torch.float32)
    else:
        raise NotImplementedError(f"norm layer {type(norm).__name__} has no HIP implementation "
                                  "(only BatchNorm3d in eval mode and NoOp)")
    act = blk.activation
    if isinstance(act, nn.LeakyReLU):
        slope = float(act.negative_slope)
    elif isinstance(act, nn.ReLU):
        slope = 0.0
    elif _is_identity(act):
        slope = 1.0
    else:
        raise NotImplementedError(f"activation {type(act).__name__} has no HIP implementation")
    L = ConvLaunch()
    L.w = w
    L.wp = H.pack_conv_weights(w)
    L.wp_b3 = None
    L.wp_c16 = None
    L.wp_v32 = None
    L.wp_d32 = None
    L.wp_rs = None
    L.wp_s2 = None
    L.wp_poly = None
    L.wp_head = None
    L.head_sc = None
    L.f16 = None
    L.scale = scale.contiguous()
    L.shift = shift.contiguous()
    L.stride = int(conv.stride[0])
    L.neg_slope = slope
    L.cin, L.cout = int(w.shape[1]), int(cout)
    L.key = key
    blk.__dict__["_mvsgi_launch"] = L
    return L


def _to_ncdhw_view(y_ndhwc: Tensor) -> Tensor:
    """[B, D, H, W, C] storage presented with the reference's [B, C, D, H, W] shape
    (channels_last_3d strides; no copy)."""
    return y_ndhwc.permute(0, 4, 1, 2, 3)


# ------------------------------------------------------------------------------------------
# blocks (ndhwc-level helpers are used by the regulator / builder forwards)
# ------------------------------------------------------------------------------------------
class BaseConvBlk3d(nn.Module):
    def __init__(self, in_chs: int, out_chs: int, kernel_size: int, stride: int = 1, extra_pad: int = 0,
                 bias_on: bool = False, norm_layer: nn.Module = NoOp(), activation: nn.Module = NoOp()):
        super().__init__()
        self.conv_layer = nn.Conv3d(in_chs, out_chs, kernel_size, padding=(kernel_size // 2) + extra_pad,
                                    bias=bias_on, stride=stride)
        self.norm_layer = norm_layer
        self.activation = activation

    __getstate__ = module_getstate

    def forward_ndhwc(self, x: Tensor, res: Optional[Tensor] = None) -> Tensor:
        return lower_conv_block(self).run(x, res)

    def forward(self, x: Tensor, res: Optional[Tensor] = None) -> Tensor:
        r = None if res is None else H.as_ndhwc(res)
        return _to_ncdhw_view(self.forward_ndhwc(H.as_ndhwc(x), r))


def res_block_ndhwc(blk, x: Tensor) -> Tensor:
    """ResConvBlk3d.forward (common_modules.py:231-244) for in_chs == out_chs."""
    if not _is_identity(blk.one_by_one):
        raise NotImplementedError("ResConvBlk3d with a 1x1x1 projection (in_chs != out_chs) is not on the hot path")
    if getattr(blk, "out_pad", 0) != 0:
        raise NotImplementedError("ResConvBlk3d out_pad != 0")
    r = lower_conv_block(blk.blk1).run(x)
    return lower_conv_block(blk.blk2).run(r, res=x)


class ResConvBlk3d(nn.Module):
    def __init__(self, in_chs: int, out_chs: int, kernel_size: int = 3, in_stride: int = 1, out_stride: int = 1,
                 out_pad: int = 0, activation: nn.Module = NoOp(), norm_layer: nn.Module = NoOp()):
        super().__init__()
        self.in_chs, self.out_chs, self.k_sz = in_chs, out_chs, kernel_size
        self.in_stride, self.out_stride, self.out_pad = in_stride, out_stride, out_pad
        self.blk1 = BaseConvBlk3d(in_chs, out_chs, kernel_size, stride=in_stride,
                                  activation=copy.deepcopy(activation), norm_layer=copy.deepcopy(norm_layer))
        self.blk2 = BaseConvBlk3d(out_chs, out_chs, kernel_size, stride=out_stride,
                                  activation=copy.deepcopy(activation), norm_layer=copy.deepcopy(norm_layer))
        if in_chs != out_chs:
            self.one_by_one = BaseConvBlk3d(in_chs, out_chs, 1, stride=out_stride * in_stride,
                                            activation=copy.deepcopy(activation),
                                            norm_layer=copy.deepcopy(norm_layer))
        else:
            self.one_by_one = NoOp()

    def forward(self, x: Tensor) -> Tensor:
        return _to_ncdhw_view(res_block_ndhwc(self, H.as_ndhwc(x)))


def resize_conv_ndhwc(blk, x: Tensor, res: Optional[Tensor] = None) -> Tensor:
    """ResizeConv3d.forward (common_modules.py:332-355): trilinear to int(scale*s)+out_pad,
    optional second resize to the skip's size, then conv(+res)."""
    up = [int(blk.scale * s) + blk.out_pad for s in x.shape[1:4]]
    L = lower_conv_block(blk.conv)
    if up == [2 * s for s in x.shape[1:4]] and (res is None or tuple(res.shape[1:4]) == tuple(up)) and L.can_fuse_up2():
        return L.run_up2(x, res)                  # upsample evaluated inside the conv's staging path
    x = H.resize_trilinear(x, up)
    if res is not None and tuple(x.shape[1:4]) != tuple(res.shape[1:4]):
        x = H.resize_trilinear(x, res.shape[1:4])
    return lower_conv_block(blk.conv).run(x, res)


class ResizeConv3d(nn.Module):
    def __init__(self, in_chs: int, out_chs: int, kernel_size: int, stride: int = 1, extra_pad: int = 0,
                 out_pad: int = 0, activation: nn.Module = NoOp(), norm_layer: nn.Module = NoOp()):
        super().__init__()
        self.scale = stride
        self.out_pad = out_pad
        self.conv = BaseConvBlk3d(in_chs, out_chs, kernel_size, stride=1, extra_pad=extra_pad,
                                  activation=copy.deepcopy(activation), norm_layer=copy.deepcopy(norm_layer))

    def forward(self, x: Tensor, res: Optional[Tensor] = None) -> Tensor:
        r = None if res is None else H.as_ndhwc(res)
        return _to_ncdhw_view(resize_conv_ndhwc(self, H.as_ndhwc(x), r))
