"""Tensor-level wrappers over the C ABI: argument checking, output allocation (PyTorch is
only the device allocator and stream provider here) and the call through ctypes.

All activations are channels-last fp32 ("NDHWC": [B, D, H, W, C]) unless stated.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

import os

CONV_AUTO, CONV_DIRECT, CONV_MFMA, CONV_BF16X3, CONV_BF16X3_C16, CONV_BF16X3_V32, CONV_BF16X3_D32 = 0, 1, 2, 3, 4, 5, 6
SPLIT_F16 = 1          # `fmt` of the *_fmt entry points (include/mvsgi.h MVSGI_SPLIT_F16): split-padded data / weights hold fp16 pairs
CONV_F16 = 0x100       # flag OR-ed into CONV_BF16X3 / _C16 / _V32: the same kernel in the fp16 split (include/mvsgi.h MVSGI_CONV_F16)

# Arithmetic of the conv layers:
#   "f16x3" (default) = split-fp16 MFMA (x = hi + lo, hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16, fp32 accumulate; 11 + 11
#                       significant bits per operand; operands saturate at +-65504 (never inf / nan); weights pre-scaled per output
#                       channel by a power of two).  Inverse distance ~8x closer to the reference than the bf16 split, and since the
#                       level-0 residual convs of the (16, 32) regulator run in Winograd form in this split (csrc/conv3d_wino.hip; exact
#                       for |activation| <= 16376) also the fastest mode;
#   "bf16x3"          = the same three products in the bf16 split (8 + 8 bits per operand, fp32's range): the answer for activations
#                       beyond fp16's range (HotPath.precision_check tells the two apart);
#   "f32"             = exact fp32 MFMA (v_mfma_f32_16x16x4_f32; bit-for-bit an fp32 fmaf chain, ~4x slower).
CONV_MODES = ("f32", "bf16x3", "f16x3")
_CONV_MODE = os.environ.get("MVSGI_CONV_MODE", "f16x3")
if _CONV_MODE not in CONV_MODES:
    raise ValueError(f"MVSGI_CONV_MODE={_CONV_MODE!r} not in {CONV_MODES}")


def exp_env(name: str, default: str) -> str:
    """Experiment switches -- A/B knobs whose measurement is recorded in DESIGN.md / DESIGN_HISTORY.md as neutral or slower -- are
    read only when MVSGI_EXPERIMENTAL=1; the product configuration surface is MVSGI_CONV_MODE, MVSGI_RIG_CACHE, MVSGI_POLY,
    MVSGI_S2RS, MVSGI_HEAD_SPLIT, MVSGI_FRONT_CHUNK (each covered by tests/test_gpu_parity.py::test_product_switches_off), MVSGI_WINO
    (test_full_size_winograd_level0_vs_direct_kernel_and_reference_golden) and MVSGI_LIB."""
    if os.environ.get("MVSGI_EXPERIMENTAL") == "1":
        return os.environ.get(name, default)
    if name in os.environ and name not in _EXP_WARNED:      # a probe / A-B script that forgot the gate would measure the default twice
        _EXP_WARNED.add(name)
        import warnings
        warnings.warn(f"{name} is an experiment switch: it is read only with MVSGI_EXPERIMENTAL=1 (ignored)", RuntimeWarning, stacklevel=2)
    return default


_EXP_WARNED = set()


def set_conv_mode(mode: str) -> None:
    global _CONV_MODE
    if mode not in CONV_MODES:
        raise ValueError(f"conv mode {mode!r} not in {CONV_MODES}")
    _CONV_MODE = mode


def get_conv_mode() -> str:
    return _CONV_MODE


# ---- range report of the fp16 split (include/mvsgi.h: mvsgi_saturation_flags) ----
# The reference computes in fp32 with no clamp (common_modules.py:105-115); the default arithmetic saturates what it writes or stages
# in fp16 pieces.  Every kernel that clamps raises a sticky flag when a clamp ENGAGED; the wrappers below read it -- a load from
# pinned host memory, no stream operation -- and turn it into an exception (MVSGI_RANGE_CHECK=raise, the default), a warning (=warn,
# once per kind) or nothing (=off).  A flag says something about launches that have COMPLETED: the path's calls are asynchronous, so
# the entry checks of HotPath / the drop-in modules report the frames before the current one; HotPath.check_range() and
# InferencePipeline (which synchronise) report the frame itself.
SAT_SWEEP, SAT_SPLIT, SAT_WINO = 1, 2, 4
_SAT_TEXT = {SAT_SWEEP: "the sweep's cost volume reached +-65504 (un-normalised features?)",
             SAT_SPLIT: "an activation written or staged in fp16 pieces reached +-65504",
             SAT_WINO: "an activation of the Winograd-form level reached its +-16376 range (or a transformed sum +-65504)"}
RANGE_CHECKS = ("raise", "warn", "off")
_RANGE_CHECK = os.environ.get("MVSGI_RANGE_CHECK", "raise")
if _RANGE_CHECK not in RANGE_CHECKS:
    raise ValueError(f"MVSGI_RANGE_CHECK={_RANGE_CHECK!r} not in {RANGE_CHECKS}")
_RANGE_WARNED = set()
_RANGE_SEEN = 0             # OR of every flag check_range() has reported (and cleared) in this process


class MvsgiRangeError(RuntimeError):
    """The fp16 split left its range: results since the last check are saturated, not the reference's."""


def set_range_check(policy: str) -> None:
    global _RANGE_CHECK
    if policy not in RANGE_CHECKS:
        raise ValueError(f"range check policy {policy!r} not in {RANGE_CHECKS}")
    _RANGE_CHECK = policy


def get_range_check() -> str:
    return _RANGE_CHECK


def saturation_flags(clear: bool = False) -> int:
    """OR of the SAT_* bits raised by completed launches since the last clear (sticky; no synchronisation)."""
    import ctypes
    out = ctypes.c_uint(0)
    if _lib.load().mvsgi_saturation_flags(1 if clear else 0, ctypes.byref(out)) != 0:
        # the pinned words could not be allocated (no usable device): then no kernel has run either -- every launcher of a kernel
        # that can clamp refuses to launch without them -- and there is nothing to report
        return 0
    return int(out.value)


def saturation_words():
    """The raw report words (diagnostics): a kernel only ever stores 1 into the word of its kind."""
    import ctypes
    out = (ctypes.c_uint * 8)()
    if _lib.load().mvsgi_saturation_words(out) != 0:
        return [0] * 8
    return [int(v) for v in out]


def range_flags_seen() -> int:
    """OR of the flags check_range() has reported so far in this process (it clears the library's sticky words when it reports)."""
    return _RANGE_SEEN


def check_range(where: str = "", sync_device=None) -> int:
    """Apply the MVSGI_RANGE_CHECK policy to the flags raised so far (after synchronising `sync_device`, if given) and clear them.
    -> the flags that were raised."""
    if _RANGE_CHECK == "off":
        return 0
    if sync_device is not None:
        torch.cuda.synchronize(sync_device)
    f = saturation_flags(clear=False)
    if not f:
        return 0
    words = saturation_words()
    saturation_flags(clear=True)
    global _RANGE_SEEN
    _RANGE_SEEN |= f
    what = "; ".join(t for b, t in _SAT_TEXT.items() if f & b)
    msg = (f"mvs_gi_amd{' (' + where + ')' if where else ''}: the fp16 split (MVSGI_CONV_MODE=f16x3, the default) left its range (flags 0x{f:x}, words {words[:4]}) -- {what}. "
           "Results computed since the last check are saturated, not the reference's fp32 results "
           "(dsta_mvs/model/common/common_modules.py:105-115 has no clamp). Re-run with MVSGI_CONV_MODE=bf16x3 "
           "(hip_ops.set_conv_mode('bf16x3'): fp32's range, ~8x the rounding error) or 'f32'; HotPath.precision_check(frames) measures "
           "which arithmetic a checkpoint needs. MVSGI_RANGE_CHECK=warn|off relaxes this check.")
    if _RANGE_CHECK == "raise":
        raise MvsgiRangeError(msg)
    if f not in _RANGE_WARNED:
        _RANGE_WARNED.add(f)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
    return f


def split_mode() -> bool:
    """The library's mode is one of the two 16-bit splits (the streaming split kernels serve the layer)."""
    return _CONV_MODE in ("bf16x3", "f16x3")


def mode_fmt() -> str:
    """Element type of the (hi | lo) pieces the library's mode writes into split-padded buffers."""
    return "f16" if _CONV_MODE == "f16x3" else "bf16"


def _fmt_code(fmt: str) -> int:
    if fmt not in ("bf16", "f16"):
        raise ValueError(f"split format {fmt!r} not in ('bf16', 'f16')")
    return SPLIT_F16 if fmt == "f16" else 0


def _pow2_unscale(amax: torch.Tensor):
    """k per entry such that amax * 2^k lies in (512, 1024] (0 for a zero entry) -> (2^k, 2^-k) as fp32 tensors."""
    k = torch.where(amax > 0, torch.floor(torch.log2(1024.0 / amax.clamp_min(1e-37))), torch.zeros_like(amax)).clamp(-100.0, 100.0)
    return torch.exp2(k), torch.exp2(-k)


def _stream_ptr(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: mvs_gi_amd runs on the GPU only (tensor is on {t.device}); "
                           "there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------------------
def _feats_nhwc(feats):
    """[B, N, C, Hi, Wi] -> channels-last storage [B, N, Hi, Wi, C] (no copy if it already is)."""
    if feats.dim() != 5:
        raise AssertionError(f"feats must be [B, N, C, Hi, Wi], got {tuple(feats.shape)}")
    v = feats.permute(0, 1, 3, 4, 2)
    if v.is_contiguous() and v.data_ptr() % 16 == 0 and v.is_cuda and v.dtype == torch.float32:
        return v
    lib = _lib.load()
    f = _dev(feats, "feats")
    B, N, C, Hi, Wi = f.shape
    y = torch.empty((B, N, Hi, Wi, C), device=f.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_ncv_to_nvc_f32(f.data_ptr(), y.data_ptr(), B * N, C, Hi * Wi, _stream_ptr(f)),
               "mvsgi_ncv_to_nvc_f32")
    return y


def nhwc_sweep_ok(feats) -> bool:
    return feats.dim() == 5 and feats.shape[2] % 4 == 0 and feats.shape[1] <= 4


def sweep_validity(grids, grid_masks, masks) -> torch.Tensor:
    """Rig-constant validity byte per voxel, [B, D, Ho, Wo] uint8 (bit cam = camera cam valid):
    (bilinear_grid_sample(masks) > 0) & grid_masks of spherical_sweep_avg.py:92-102."""
    lib = _lib.load()
    grids = _dev(grids, "grids")
    masks = _dev(masks, "masks")
    gm, gm_f32 = _gm_arg(grid_masks)
    B, N, D, Ho, Wo, two = grids.shape
    if two != 2 or N > 8:
        raise AssertionError(f"grids must be [B, N<=8, D, Ho, Wo, 2], got {tuple(grids.shape)}")
    if tuple(gm.shape[:5]) != (B, N, D, Ho, Wo) or gm.numel() != B * N * D * Ho * Wo:
        raise AssertionError(f"grid_masks {tuple(gm.shape)} do not match grids {tuple(grids.shape)}")
    if masks.shape[0] != B or masks.shape[1] != N or masks.numel() != B * N * masks.shape[-2] * masks.shape[-1]:
        raise AssertionError(f"masks {tuple(masks.shape)} do not match grids {tuple(grids.shape)}")
    Hm, Wm = masks.shape[-2:]
    vmask = torch.empty((B, D, Ho, Wo), device=grids.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_sweep_validity_u8(grids.data_ptr(), gm.data_ptr(), gm_f32, masks.data_ptr(), vmask.data_ptr(),
                                           B, N, Hm, Wm, D, Ho, Wo, _stream_ptr(grids)), "mvsgi_sweep_validity_u8")
    return vmask


def sweep_std_valid(feats, grids, vmask) -> torch.Tensor:
    """sweep_std with the cached validity byte (channels-last kernel only) -> vol_raw [B, D, Ho, Wo, C].
    grids / vmask with batch 1 against feats with batch B > 1 = one rig shared by the whole batch."""
    lib = _lib.load()
    if not nhwc_sweep_ok(feats):
        raise AssertionError(f"sweep_std_valid needs C % 4 == 0 and N <= 4, got feats {tuple(feats.shape)}")
    B, N, C, Hi, Wi = feats.shape
    f = _feats_nhwc(feats)
    grids = _dev(grids, "grids")
    vmask = _dev(vmask, "vmask", torch.uint8)
    Bg, Ng, D, Ho, Wo, two = grids.shape
    if (Ng, two) != (N, 2) or Bg not in (1, B):
        raise AssertionError(f"grids {tuple(grids.shape)} do not match feats {tuple(feats.shape)}")
    if tuple(vmask.shape) != (Bg, D, Ho, Wo):
        raise AssertionError(f"vmask {tuple(vmask.shape)} does not match grids {tuple(grids.shape)}")
    vol = torch.empty((B, D, Ho, Wo, C), device=f.device, dtype=torch.float32)
    fn = lib.mvsgi_sweep_std_nhwc_valid_rig_f32 if (Bg == 1 and B > 1) else lib.mvsgi_sweep_std_nhwc_valid_f32
    _lib.check(fn(f.data_ptr(), grids.data_ptr(), vmask.data_ptr(), vol.data_ptr(), B, N, C, Hi, Wi, D, Ho, Wo, _stream_ptr(f)),
               "mvsgi_sweep_std_nhwc_valid_f32")
    return vol


def sweep_std_valid_split(feats, grids, vmask, out: "SplitAct", fmt: str = "bf16") -> "SplitAct":
    """sweep_std_valid with vol_raw written split-padded (C == 16) into `out` (B, D, Ho, Wo, 16), its pieces in the split `fmt`."""
    lib = _lib.load()
    B, N, C, Hi, Wi = feats.shape
    f = _feats_nhwc(feats)
    grids = _dev(grids, "grids")
    vmask = _dev(vmask, "vmask", torch.uint8)
    Bg, Ng, D, Ho, Wo, two = grids.shape
    if (Ng, two) != (N, 2) or Bg not in (1, B) or tuple(vmask.shape) != (Bg, D, Ho, Wo) or C != 16:
        raise AssertionError(f"grids {tuple(grids.shape)} / vmask {tuple(vmask.shape)} do not match feats {tuple(feats.shape)}")
    if out.shape != (B, D, Ho, Wo, C):
        raise AssertionError(f"split output {out.shape} does not match {(B, D, Ho, Wo, C)}")
    _lib.check(lib.mvsgi_sweep_std_nhwc_valid_split_fmt(f.data_ptr(), grids.data_ptr(), vmask.data_ptr(), out.buf.data_ptr(), B, N, C, Hi,
                                                        Wi, D, Ho, Wo, Bg, _fmt_code(fmt), _stream_ptr(f)), "mvsgi_sweep_std_nhwc_valid_split")
    out.fmt = fmt
    return out


def sweep_std(feats, grids, grid_masks, masks, layout: str = "auto") -> torch.Tensor:
    """-> vol_raw [B, D, Ho, Wo, C] (masked variance over cameras).  layout: 'auto' uses the
    channels-last kernel when C % 4 == 0 and N <= 4 (transposing NCHW feats once), 'nchw'
    forces the plane-gather kernel."""
    lib = _lib.load()
    if layout == "auto" and nhwc_sweep_ok(feats):
        return _sweep_std_nhwc(feats, grids, grid_masks, masks)
    feats = _dev(feats, "feats")
    grids = _dev(grids, "grids")
    masks = _dev(masks, "masks")
    if grid_masks.dtype == torch.bool:
        gm = _dev(grid_masks, "grid_masks", torch.bool)
        gm_f32 = 0
    elif grid_masks.dtype == torch.uint8:
        gm = _dev(grid_masks, "grid_masks", torch.uint8)
        gm_f32 = 0
    else:
        gm = _dev(grid_masks.to(torch.float32) if grid_masks.dtype != torch.float32 else grid_masks, "grid_masks")
        gm_f32 = 1
    B, N, C, Hi, Wi = feats.shape
    Bg, Ng, D, Ho, Wo, two = grids.shape
    if (Bg, Ng, two) != (B, N, 2):
        raise AssertionError(f"grids {tuple(grids.shape)} do not match feats {tuple(feats.shape)}")
    if tuple(gm.shape[:5]) != (B, N, D, Ho, Wo) or gm.numel() != B * N * D * Ho * Wo:
        raise AssertionError(f"grid_masks {tuple(gm.shape)} do not match grids {tuple(grids.shape)}")
    if masks.shape[0] != B or masks.shape[1] != N or masks.numel() != B * N * masks.shape[-2] * masks.shape[-1]:
        raise AssertionError(f"masks {tuple(masks.shape)} do not match feats {tuple(feats.shape)}")
    Hm, Wm = masks.shape[-2:]
    vol = torch.empty((B, D, Ho, Wo, C), device=feats.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_sweep_std_f32(feats.data_ptr(), grids.data_ptr(), gm.data_ptr(), gm_f32, masks.data_ptr(),
                                       vol.data_ptr(), B, N, C, Hi, Wi, Hm, Wm, D, Ho, Wo, _stream_ptr(feats)),
               "mvsgi_sweep_std_f32")
    return vol


def _gm_arg(grid_masks):
    if grid_masks.dtype == torch.bool:
        return _dev(grid_masks, "grid_masks", torch.bool), 0
    if grid_masks.dtype == torch.uint8:
        return _dev(grid_masks, "grid_masks", torch.uint8), 0
    return _dev(grid_masks.to(torch.float32) if grid_masks.dtype != torch.float32 else grid_masks, "grid_masks"), 1


def _sweep_std_nhwc(feats, grids, grid_masks, masks) -> torch.Tensor:
    lib = _lib.load()
    B, N, C, Hi, Wi = feats.shape
    f = _feats_nhwc(feats)
    grids = _dev(grids, "grids")
    masks = _dev(masks, "masks")
    gm, gm_f32 = _gm_arg(grid_masks)
    Bg, Ng, D, Ho, Wo, two = grids.shape
    if (Bg, Ng, two) != (B, N, 2):
        raise AssertionError(f"grids {tuple(grids.shape)} do not match feats {tuple(feats.shape)}")
    if tuple(gm.shape[:5]) != (B, N, D, Ho, Wo) or gm.numel() != B * N * D * Ho * Wo:
        raise AssertionError(f"grid_masks {tuple(gm.shape)} do not match grids {tuple(grids.shape)}")
    if masks.shape[0] != B or masks.shape[1] != N or masks.numel() != B * N * masks.shape[-2] * masks.shape[-1]:
        raise AssertionError(f"masks {tuple(masks.shape)} do not match feats {tuple(feats.shape)}")
    Hm, Wm = masks.shape[-2:]
    vol = torch.empty((B, D, Ho, Wo, C), device=f.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_sweep_std_nhwc_f32(f.data_ptr(), grids.data_ptr(), gm.data_ptr(), gm_f32, masks.data_ptr(),
                                            vol.data_ptr(), B, N, C, Hi, Wi, Hm, Wm, D, Ho, Wo, _stream_ptr(f)),
               "mvsgi_sweep_std_nhwc_f32")
    return vol


def sweep_cat(feats, grids, layout: str = "auto") -> torch.Tensor:
    """-> vol_raw [B, D, Ho, Wo, N*C] (channel = cam*C + c)."""
    lib = _lib.load()
    if layout == "auto" and feats.dim() == 5 and feats.shape[2] % 4 == 0:
        B, N, C, Hi, Wi = feats.shape
        f = _feats_nhwc(feats)
        grids = _dev(grids, "grids")
        Bg, Ng, D, Ho, Wo, two = grids.shape
        if (Bg, Ng, two) != (B, N, 2):
            raise AssertionError(f"grids {tuple(grids.shape)} do not match feats {tuple(feats.shape)}")
        vol = torch.empty((B, D, Ho, Wo, N * C), device=f.device, dtype=torch.float32)
        _lib.check(lib.mvsgi_sweep_cat_nhwc_f32(f.data_ptr(), grids.data_ptr(), vol.data_ptr(), B, N, C, Hi, Wi, D,
                                                Ho, Wo, _stream_ptr(f)), "mvsgi_sweep_cat_nhwc_f32")
        return vol
    feats = _dev(feats, "feats")
    grids = _dev(grids, "grids")
    B, N, C, Hi, Wi = feats.shape
    Bg, Ng, D, Ho, Wo, two = grids.shape
    if (Bg, Ng, two) != (B, N, 2):
        raise AssertionError(f"grids {tuple(grids.shape)} do not match feats {tuple(feats.shape)}")
    vol = torch.empty((B, D, Ho, Wo, N * C), device=feats.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_sweep_cat_f32(feats.data_ptr(), grids.data_ptr(), vol.data_ptr(), B, N, C, Hi, Wi, D, Ho, Wo,
                                       _stream_ptr(feats)), "mvsgi_sweep_cat_f32")
    return vol


def pack_conv_weights(w_oidhw: torch.Tensor) -> Optional[torch.Tensor]:
    """[Cout, Cin, 3, 3, 3] -> MFMA lane-ordered layout, or None when the channel counts
    are not multiples of 16 (the direct kernel then reads w_oidhw itself)."""
    lib = _lib.load()
    w = _dev(w_oidhw, "conv weight")
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3, 3):
        raise NotImplementedError(f"only 3x3x3 kernels are supported, got {tuple(w.shape[2:])}")
    if Cin % 16 or not (Cout % 16 == 0 or Cout == 1):
        return None
    wp = torch.empty(lib.mvsgi_conv3d_packed_weight_floats(Cout, Cin), device=w.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_conv3d_pack_weights_f32(w.data_ptr(), wp.data_ptr(), Cout, Cin, _stream_ptr(w)),
               "mvsgi_conv3d_pack_weights_f32")
    return wp


def pack_conv_weights_bf16x3(w_oidhw: torch.Tensor) -> Optional[torch.Tensor]:
    """[Cout, Cin, 3, 3, 3] -> split-bf16 (hi | lo) MFMA layout, or None when unsupported."""
    lib = _lib.load()
    w = _dev(w_oidhw, "conv weight")
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3, 3) or Cin % 16 or Cout % 16:
        return None
    wp = torch.empty(lib.mvsgi_conv3d_packed_weight_bytes_bf16x3(Cout, Cin), device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv3d_pack_weights_bf16x3(w.data_ptr(), wp.data_ptr(), Cout, Cin, _stream_ptr(w)),
               "mvsgi_conv3d_pack_weights_bf16x3")
    return wp


def pack_conv_weights_f16x3(w_oidhw: torch.Tensor, layout: int = CONV_BF16X3):
    """[Cout, Cin, 3, 3, 3] -> (packed weights of the fp16 split in `layout` (CONV_BF16X3 | _C16 | _V32), unscale [Cout]) or None.
    Every output channel's weights are pre-scaled by a power of two so that its largest weight lies in (512, 1024] -- the lo parts
    (2^-11 of the weight) are then normal fp16 numbers instead of subnormals with an absolute quantum of 2^-24; `unscale` = 2^-k
    per channel goes into the epilogue's per-channel scale (exact: powers of two)."""
    lib = _lib.load()
    w = _dev(w_oidhw, "conv weight")
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3, 3) or Cin % 16 or Cout % 16 or (layout == CONV_BF16X3_C16 and Cout != 16) or \
            (layout == CONV_BF16X3_V32 and Cout % 32) or (layout == CONV_BF16X3_D32 and Cin % 32):
        return None
    amax = w.abs().amax(dim=(1, 2, 3, 4))
    k = torch.where(amax > 0, torch.floor(torch.log2(1024.0 / amax.clamp_min(1e-37))), torch.zeros_like(amax)).clamp(-100.0, 100.0)
    ws = (w * torch.exp2(k).view(-1, 1, 1, 1, 1)).contiguous()
    n = {CONV_BF16X3: lambda: lib.mvsgi_conv3d_packed_weight_bytes_bf16x3(Cout, Cin),
         CONV_BF16X3_C16: lambda: lib.mvsgi_conv3d_packed_weight_bytes_bf16x3_c16(Cin),
         CONV_BF16X3_D32: lambda: lib.mvsgi_conv3d_packed_weight_bytes_bf16x3(Cout, Cin),
         CONV_BF16X3_V32: lambda: lib.mvsgi_conv3d_packed_weight_bytes_bf16x3_v32(Cout, Cin)}[layout]()
    wp = torch.empty(n, device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv3d_pack_weights_split(ws.data_ptr(), wp.data_ptr(), Cout, Cin, layout | CONV_F16, _stream_ptr(w)),
               "mvsgi_conv3d_pack_weights_split")
    return wp, torch.exp2(-k).contiguous()


def conv3d_d32_applies(B, Cin, Din, Hin, Win, Cout, stride=1) -> bool:
    """Whether the split kernel on 32-channel slices (impl CONV_BF16X3_D32: 27 k-steps per 32 channels instead of 28, half the
    slices per unit) serves this problem: Cin % 32 == 0, stride 1, a launch large enough for its 128- / 160-voxel bricks."""
    return bool(_lib.load().mvsgi_conv3d_d32_applies(B, Cin, Din, Hin, Win, Cout, stride))


def conv3d_up2_d32_applies(B, Cin, Dl, Hl, Wl, Cout) -> bool:
    """... and the fused upsample + conv (conv3d_up2 with w_layout CONV_BF16X3_D32; low-resolution sizes)."""
    return bool(_lib.load().mvsgi_conv3d_up2_d32_applies(B, Cin, Dl, Hl, Wl, Cout))


def pack_conv_weights_bf16x3_d32(w_oidhw: torch.Tensor) -> Optional[torch.Tensor]:
    """[Cout, Cin, 3, 3, 3] -> the bf16 split's weights in the 32-channel-slice layout (CONV_BF16X3_D32), or None (Cin % 32, Cout % 16)."""
    lib = _lib.load()
    w = _dev(w_oidhw, "conv weight")
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3, 3) or Cin % 32 or Cout % 16:
        return None
    wp = torch.empty(lib.mvsgi_conv3d_packed_weight_bytes_bf16x3(Cout, Cin), device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv3d_pack_weights_split(w.data_ptr(), wp.data_ptr(), Cout, Cin, CONV_BF16X3_D32, _stream_ptr(w)),
               "mvsgi_conv3d_pack_weights_split")
    return wp


def conv3d_v32_applies(B, Cin, Din, Hin, Win, Cout, stride=1) -> bool:
    """Whether the 32x32x16-MFMA kernel (impl / w_layout CONV_BF16X3_V32) serves this problem
    (for conv3d_up2 pass the upsampled input size)."""
    return bool(_lib.load().mvsgi_conv3d_v32_applies(B, Cin, Din, Hin, Win, Cout, stride))


def pack_conv_weights_bf16x3_v32(w_oidhw: torch.Tensor) -> Optional[torch.Tensor]:
    """[Cout % 32 == 0, Cin % 16 == 0, 3, 3, 3] -> 32x32x16-MFMA split-bf16 layout, or None when unsupported."""
    lib = _lib.load()
    w = _dev(w_oidhw, "conv weight")
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3, 3) or Cin % 16 or Cout % 32:
        return None
    wp = torch.empty(lib.mvsgi_conv3d_packed_weight_bytes_bf16x3_v32(Cout, Cin), device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv3d_pack_weights_bf16x3_v32(w.data_ptr(), wp.data_ptr(), Cout, Cin, _stream_ptr(w)),
               "mvsgi_conv3d_pack_weights_bf16x3_v32")
    return wp


def pack_conv_weights_bf16x3_c16(w_oidhw: torch.Tensor) -> Optional[torch.Tensor]:
    """[16, Cin, 3, 3, 3] -> plane-schedule split-bf16 layout (impl CONV_BF16X3_C16), or None when unsupported."""
    lib = _lib.load()
    w = _dev(w_oidhw, "conv weight")
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3, 3) or Cin % 16 or Cout != 16:
        return None
    wp = torch.empty(lib.mvsgi_conv3d_packed_weight_bytes_bf16x3_c16(Cin), device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv3d_pack_weights_bf16x3_c16(w.data_ptr(), wp.data_ptr(), Cin, _stream_ptr(w)),
               "mvsgi_conv3d_pack_weights_bf16x3_c16")
    return wp


def conv3d(x, w_oidhw, w_packed, scale, shift, res=None, stride=1, neg_slope=0.01, impl=CONV_AUTO, out=None):
    """x [B, D, H, W, Cin] -> y [B, Do, Ho, Wo, Cout] = act(conv(x) * scale + shift (+ res));
    act(v) = v if v > 0 else v * neg_slope (1.0 = no activation)."""
    lib = _lib.load()
    x = _dev(x, "x")
    B, Din, Hin, Win, Cin = x.shape
    Cout = scale.numel()
    Do, Ho, Wo = (Din - 1) // stride + 1, (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    if res is not None:
        res = _dev(res, "res")
        if tuple(res.shape) != (B, Do, Ho, Wo, Cout):
            raise AssertionError(f"residual {tuple(res.shape)} does not match output {(B, Do, Ho, Wo, Cout)}")
    y = out if out is not None else torch.empty((B, Do, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_conv3d_f32(x.data_ptr(), _ptr(w_oidhw), _ptr(w_packed), scale.data_ptr(), shift.data_ptr(),
                                    _ptr(res), y.data_ptr(), B, Cin, Din, Hin, Win, Cout, stride, float(neg_slope),
                                    impl, _stream_ptr(x)), "mvsgi_conv3d_f32")
    return y


def conv3d_up2(x, w_packed_b3, scale, shift, res=None, neg_slope=0.01, out=None, w_layout=CONV_BF16X3):
    """Fused trilinear x2 upsample + conv3d (split-bf16): x [B, Dl, Hl, Wl, Cin] -> y [B, 2Dl, 2Hl, 2Wl, Cout].
    w_layout: CONV_BF16X3 (pack_conv_weights_bf16x3) or CONV_BF16X3_C16 (pack_conv_weights_bf16x3_c16, Cout == 16)."""
    lib = _lib.load()
    x = _dev(x, "x")
    B, Dl, Hl, Wl, Cin = x.shape
    Cout = scale.numel()
    shp = (B, 2 * Dl, 2 * Hl, 2 * Wl, Cout)
    if res is not None:
        res = _dev(res, "res")
        if tuple(res.shape) != shp:
            raise AssertionError(f"residual {tuple(res.shape)} does not match output {shp}")
    y = out if out is not None else torch.empty(shp, device=x.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_conv3d_up2_f32(x.data_ptr(), _ptr(w_packed_b3), w_layout, scale.data_ptr(), shift.data_ptr(), _ptr(res),
                                        y.data_ptr(), B, Cin, Dl, Hl, Wl, Cout, float(neg_slope), _stream_ptr(x)),
               "mvsgi_conv3d_up2_f32")
    return y


def conv3d_up2_out_split(x, w_packed_b3, scale, shift, out: "SplitAct", res=None, neg_slope=0.01, w_layout=CONV_BF16X3) -> "SplitAct":
    """conv3d_up2 with the result written split-padded into `out` (B, 2Dl, 2Hl, 2Wl, Cout)."""
    lib = _lib.load()
    x = _dev(x, "x")
    B, Dl, Hl, Wl, Cin = x.shape
    Cout = scale.numel()
    if out.shape != (B, 2 * Dl, 2 * Hl, 2 * Wl, Cout):
        raise AssertionError(f"split output {out.shape} does not match {(B, 2 * Dl, 2 * Hl, 2 * Wl, Cout)}")
    if res is not None:
        res = _dev(res, "res")
        if tuple(res.shape) != out.shape:
            raise AssertionError(f"residual {tuple(res.shape)} does not match output {out.shape}")
    _lib.check(lib.mvsgi_conv3d_up2_f32_out_split(x.data_ptr(), _ptr(w_packed_b3), w_layout, scale.data_ptr(), shift.data_ptr(), _ptr(res),
                                                  out.buf.data_ptr(), B, Cin, Dl, Hl, Wl, Cout, float(neg_slope), _stream_ptr(x)),
               "mvsgi_conv3d_up2_f32_out_split")
    out.fmt = "f16" if w_layout & CONV_F16 else "bf16"
    return out


def conv3d_up2_poly_applies(cin: int, cout: int, neg_slope: float) -> bool:
    """The polyphase ResizeConv3d kernel (csrc/conv3d_up2poly.hip) serves this layer (no skip input)."""
    return cin == 32 and cout == 16 and 0.0 <= neg_slope <= 1.0


def conv3d_up2_poly_plan(w_oidhw: torch.Tensor, D: int, H: int, W: int, fmt: str = "bf16"):
    """Lowering of a [16, 32, 3, 3, 3] ResizeConv3d weight for a low-resolution input of D x H x W voxels: folded phase weights in
    the register-stationary layout, face-correction weights and role tables (built on the host by the library, float64).
    fmt = 'f16': the plan in the fp16 split -> (plan, unscale [16]): the weights are pre-scaled per output channel by a power of two
    (the folded phase weights are linear in them) and `unscale` goes into the layer's scale."""
    lib = _lib.load()
    if tuple(w_oidhw.shape) != (16, 32, 3, 3, 3):
        raise AssertionError(f"polyphase plan needs a [16, 32, 3, 3, 3] weight, got {tuple(w_oidhw.shape)}")
    n = lib.mvsgi_conv3d_up2_poly_plan_bytes(D, H, W)
    if not n:
        raise RuntimeError("mvsgi_conv3d_up2_poly_plan_bytes: bad dims")
    wh = w_oidhw.detach().to("cpu", torch.float32).contiguous()
    unscale = None
    if fmt == "f16":
        up, un = _pow2_unscale(wh.abs().amax(dim=(1, 2, 3, 4)))
        wh = (wh * up.view(-1, 1, 1, 1, 1)).contiguous()
        unscale = un.to(w_oidhw.device)
    plan = torch.empty(n, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv3d_up2_poly_plan_fmt(wh.data_ptr(), plan.data_ptr(), D, H, W, _fmt_code(fmt)), "mvsgi_conv3d_up2_poly_plan")
    plan = plan.to(w_oidhw.device)
    return (plan, unscale) if fmt == "f16" else plan


def conv3d_up2_poly(x: "SplitAct", plan: torch.Tensor, scale, shift, neg_slope=0.01, out=None) -> torch.Tensor:
    """act(conv(trilinear_x2(x)) * scale + shift) for Cin = 32, Cout = 16 in polyphase form: x split-padded (low resolution)
    -> fp32 [B, 2D, 2H, 2W, 16]."""
    lib = _lib.load()
    if x.C != 32 or scale.numel() != 16:
        raise AssertionError("conv3d_up2_poly is the 32 -> 16 kernel")
    shp = (x.B, 2 * x.D, 2 * x.H, 2 * x.W, 16)
    y = out if out is not None else torch.empty(shp, device=x.buf.device, dtype=torch.float32)
    if tuple(y.shape) != shp or not y.is_contiguous():
        raise AssertionError(f"output {tuple(y.shape)} does not match {shp}")
    _lib.check(lib.mvsgi_conv3d_up2_poly_fmt(x.buf.data_ptr(), plan.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), 0,
                                             x.B, x.D, x.H, x.W, float(neg_slope), _fmt_code(x.fmt), _stream_ptr(x.buf)),
               "mvsgi_conv3d_up2_poly_f32")
    return y


POLY_WINO = os.environ.get("MVSGI_POLY_WINO", "1") != "0"      # 0: the polyphase layer's main kernel stays the direct register-stationary one in the fp16 split too


def conv3d_up2_poly_wino_pays(B: int, D: int, Hh: int, W: int) -> bool:
    """conv3d_up2_poly_split on fp16 pairs of this (low-resolution) geometry and batch runs its main kernel in Winograd form
    (csrc/conv3d_wino_up2.hip): D == 8, H even, W a multiple of 32 and enough units to fill the chip."""
    return POLY_WINO and bool(_lib.load().mvsgi_conv3d_up2_poly_wino_pays(int(B), int(D), int(Hh), int(W)))


def conv3d_up2_poly_split(x: "SplitAct", plan: torch.Tensor, scale, shift, out: "SplitAct", neg_slope=0.01, direct: bool = False,
                          wino: bool = False) -> "SplitAct":
    """conv3d_up2_poly with the result written split-padded into `out` (B, 2D, 2H, 2W, 16).  In the fp16 split the library picks
    the main kernel: the Winograd form where conv3d_up2_poly_wino_pays(), else the direct kernel; `direct` (or MVSGI_POLY_WINO=0)
    keeps the direct kernel, `wino` forces the Winograd form (an error where it does not apply)."""
    lib = _lib.load()
    if x.C != 32 or scale.numel() != 16 or out.shape != (x.B, 2 * x.D, 2 * x.H, 2 * x.W, 16):
        raise AssertionError(f"conv3d_up2_poly_split: input {x.shape}, output {out.shape}")
    _lib.check(lib.mvsgi_conv3d_up2_poly_fmt(x.buf.data_ptr(), plan.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.buf.data_ptr(),
                                             5 if wino else (3 if (direct or not POLY_WINO) else 1),
                                             x.B, x.D, x.H, x.W, float(neg_slope), _fmt_code(x.fmt), _stream_ptr(x.buf)),
               "mvsgi_conv3d_up2_poly_split")
    out.fmt = x.fmt
    return out


def pack_head_split_weights(w_oidhw: torch.Tensor) -> Optional[torch.Tensor]:
    """[1, Cin % 16 == 0, 3, 3, 3] -> the fragment layout of conv3d_head_split, or None when unsupported."""
    lib = _lib.load()
    w = _dev(w_oidhw, "conv weight")
    if w.shape[0] != 1 or tuple(w.shape[2:]) != (3, 3, 3) or w.shape[1] % 16:
        return None
    wp = torch.empty(lib.mvsgi_conv3d_head_split_packed_weight_bytes(int(w.shape[1])), device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv3d_head_split_pack_weights(w.data_ptr(), wp.data_ptr(), int(w.shape[1]), _stream_ptr(w)),
               "mvsgi_conv3d_head_split_pack_weights")
    return wp


def pack_head_split_weights_f16(w_oidhw: torch.Tensor):
    """[1, Cin % 16 == 0, 3, 3, 3] -> (fragment layout of conv3d_head_split in the fp16 split, unscale) or None: the weights
    pre-scaled by a power of two (largest in (512, 1024]), `unscale` its inverse for the head's scale."""
    lib = _lib.load()
    w = _dev(w_oidhw, "conv weight")
    if w.shape[0] != 1 or tuple(w.shape[2:]) != (3, 3, 3) or w.shape[1] % 16:
        return None
    amax = float(w.abs().max())
    k = 0.0 if amax == 0.0 else float(torch.floor(torch.log2(torch.tensor(1024.0 / amax))))
    ws = (w * (2.0 ** k)).contiguous()
    wp = torch.empty(lib.mvsgi_conv3d_head_split_packed_weight_bytes(int(w.shape[1])), device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv3d_head_split_pack_weights_f16(ws.data_ptr(), wp.data_ptr(), int(w.shape[1]), _stream_ptr(w)),
               "mvsgi_conv3d_head_split_pack_weights_f16")
    return wp, 2.0 ** -k


def conv3d_head_split(x: "SplitAct", w_packed, scale: float, shift: float, neg_slope=1.0, out=None, f16: bool = False) -> torch.Tensor:
    """Cost head (Cout = 1) on a split-padded input -> fp32 [B, D, H, W, 1] = act(conv(x) * scale + shift).  f16: input and weights
    in the fp16 split."""
    lib = _lib.load()
    if x.fmt != ("f16" if f16 else "bf16"):
        raise AssertionError(f"conv3d_head_split(f16={f16}) on a split-padded buffer holding {x.fmt} pieces")
    y = out if out is not None else torch.empty((x.B, x.D, x.H, x.W, 1), device=x.buf.device, dtype=torch.float32)
    if tuple(y.shape) != (x.B, x.D, x.H, x.W, 1) or not y.is_contiguous():
        raise AssertionError(f"head output {tuple(y.shape)} does not match {(x.B, x.D, x.H, x.W, 1)}")
    fn = lib.mvsgi_conv3d_head_split_f16 if f16 else lib.mvsgi_conv3d_head_split
    _lib.check(fn(x.buf.data_ptr(), w_packed.data_ptr(), float(scale), float(shift), y.data_ptr(), x.B, x.C,
                  x.D, x.H, x.W, float(neg_slope), _stream_ptr(x.buf)), "mvsgi_conv3d_head_split")
    return y


def conv3d_up2_variant(B, Cin, Dl, Hl, Wl, Cout, w_layout=CONV_BF16X3) -> str:
    lib = _lib.load()
    name = lib.mvsgi_conv3d_up2_variant_f32(B, Cin, Dl, Hl, Wl, Cout, w_layout)
    if name is None:
        raise RuntimeError("mvsgi_conv3d_up2_variant_f32: " + lib.mvsgi_last_error().decode())
    return name.decode()


def conv3d_variant(B, Cin, Din, Hin, Win, Cout, stride=1, impl=CONV_AUTO) -> str:
    """Name of the kernel mvsgi_conv3d_f32 will launch for this problem (as rocprofv3 prints it)."""
    lib = _lib.load()
    name = lib.mvsgi_conv3d_variant_f32(B, Cin, Din, Hin, Win, Cout, stride, impl)
    if name is None:
        raise RuntimeError("mvsgi_conv3d_variant_f32: " + lib.mvsgi_last_error().decode())
    return name.decode()


# --------------------------------------------------------------------------------------
# split-padded activations (csrc/conv3d_rs.hip): [B, D+2, H+2, W+2, C] int32 words, each word = (hi | lo) bf16 pieces laid
# out per 16-channel slice as [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15]; zero border
# --------------------------------------------------------------------------------------
class SplitAct:
    """A split-padded activation buffer plus its logical geometry (B, D, H, W, C)."""
    __slots__ = ("buf", "B", "D", "H", "W", "C", "fmt")

    def __init__(self, B, D, H, W, C, device, buf=None):
        self.B, self.D, self.H, self.W, self.C = int(B), int(D), int(H), int(W), int(C)
        self.fmt = "bf16"          # element type of the (hi | lo) pieces: set by the kernel that writes the buffer, checked by its reader
        if buf is None:
            buf = torch.zeros((self.B, self.D + 2, self.H + 2, self.W + 2, self.C), device=device, dtype=torch.int32)
        self.buf = buf

    @property
    def shape(self):
        return (self.B, self.D, self.H, self.W, self.C)

    @property
    def device(self):
        return self.buf.device


def act_to_split(x_ndhwc: torch.Tensor, out: Optional[SplitAct] = None, fmt: str = "bf16") -> SplitAct:
    lib = _lib.load()
    x = _dev(x_ndhwc, "x")
    B, D, Hh, W, C = x.shape
    y = out if out is not None else SplitAct(B, D, Hh, W, C, x.device)
    if y.shape != (B, D, Hh, W, C):
        raise AssertionError(f"split buffer {y.shape} does not match {tuple(x.shape)}")
    _lib.check(lib.mvsgi_act_f32_to_split_fmt(x.data_ptr(), y.buf.data_ptr(), B, C, D, Hh, W, _fmt_code(fmt), _stream_ptr(x)),
               "mvsgi_act_f32_to_split")
    y.fmt = fmt
    return y


def act_from_split(x: SplitAct, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    y = out if out is not None else torch.empty(x.shape, device=x.buf.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_act_split_to_f32_fmt(x.buf.data_ptr(), y.data_ptr(), x.B, x.C, x.D, x.H, x.W, _fmt_code(x.fmt), _stream_ptr(x.buf)),
               "mvsgi_act_split_to_f32")
    return y


def pack_conv_weights_rs(w_oidhw: torch.Tensor, fmt: str = "bf16"):
    """[32, 32, 3, 3, 3] or [16, 16, 3, 3, 3] -> register-stationary layout (tap pairs in the kernel's own order), or None.
    fmt = 'f16': -> (packed weights of the fp16 split, unscale [Cout]) with the per-channel power-of-two pre-scaling of
    pack_conv_weights_f16x3."""
    lib = _lib.load()
    w = _dev(w_oidhw, "conv weight")
    unscale = None
    if fmt == "f16":
        up, unscale = _pow2_unscale(w.abs().amax(dim=(1, 2, 3, 4)))
        w = (w * up.view(-1, 1, 1, 1, 1)).contiguous()
    Cout, Cin = w.shape[:2]
    nbytes = lib.mvsgi_conv3d_rs_packed_weight_bytes(Cout, Cin) if tuple(w.shape[2:]) == (3, 3, 3) else 0
    if not nbytes:
        return None
    wp = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv3d_rs_pack_weights_fmt(w.data_ptr(), wp.data_ptr(), Cout, Cin, _fmt_code(fmt), _stream_ptr(w)),
               "mvsgi_conv3d_rs_pack_weights")
    return (wp, unscale.contiguous()) if fmt == "f16" else wp


def conv3d_rs(x: SplitAct, w_packed_rs, scale, shift, res: Optional[SplitAct] = None, neg_slope=0.01,
              out=None, out_f32: bool = False):
    """Register-stationary split-bf16 conv (Cin = Cout = 32, stride 1) on split-padded activations.  `out_f32`: the
    result is a plain fp32 [B, D, H, W, 32] tensor instead of a SplitAct (hand-over to an fp32-reading kernel)."""
    lib = _lib.load()
    Cout = scale.numel()
    if out_f32:
        y = out if out is not None else torch.empty((x.B, x.D, x.H, x.W, Cout), device=x.buf.device, dtype=torch.float32)
        yp = y.data_ptr()
        if tuple(y.shape) != (x.B, x.D, x.H, x.W, Cout) or not y.is_contiguous():
            raise AssertionError(f"fp32 output {tuple(y.shape)} does not match {(x.B, x.D, x.H, x.W, Cout)}")
    else:
        y = out if out is not None else SplitAct(x.B, x.D, x.H, x.W, Cout, x.buf.device)
        yp = y.buf.data_ptr()
        if y.shape != (x.B, x.D, x.H, x.W, Cout):
            raise AssertionError(f"split output {y.shape} does not match {(x.B, x.D, x.H, x.W, Cout)}")
    if res is not None and (res.shape != (x.B, x.D, x.H, x.W, Cout) or res.fmt != x.fmt):
        raise AssertionError(f"residual {res.shape} ({res.fmt}) does not match the output ({x.fmt})")
    _lib.check(lib.mvsgi_conv3d_rs_split_fmt(x.buf.data_ptr(), w_packed_rs.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                             None if res is None else res.buf.data_ptr(), yp, int(out_f32), x.B, x.C, x.D, x.H,
                                             x.W, Cout, float(neg_slope), _fmt_code(x.fmt), _stream_ptr(x.buf)), "mvsgi_conv3d_rs_split")
    if not out_f32:
        y.fmt = x.fmt
    return y


def conv3d_wino_applies(cin, cout, D, Hh, W, stride, neg_slope) -> bool:
    """The Winograd-form 32 -> 32 kernel (csrc/conv3d_wino.hip) serves this layer on this geometry (fp16 split only)."""
    return bool(_lib.load().mvsgi_conv3d_wino32_applies(int(cin), int(cout), int(D), int(Hh), int(W), int(stride), float(neg_slope)))


def pack_conv_weights_wino(w_oidhw: torch.Tensor):
    """[32, 32, 3, 3, 3] -> (packed weights of the Winograd F(2x2, 3x3) x direct-D kernel in the fp16 split, unscale [32]), or None.
    U[a, b, kd] = sum_kh,kw G[a, kh] G[b, kw] w[.., kd, kh, kw], pre-scaled per cout by a power of two so that max |U| lies in
    (512, 1024] (the caller folds `unscale` into the epilogue's scale), split hi = fp16(U), lo = fp16(U - hi)."""
    lib = _lib.load()
    w = _dev(w_oidhw, "conv weight")
    if tuple(w.shape) != (32, 32, 3, 3, 3):
        return None
    wp = torch.empty(lib.mvsgi_conv3d_wino32_packed_weight_bytes(), device=w.device, dtype=torch.uint8)
    unscale = torch.empty(32, device=w.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_conv3d_wino32_pack_weights(w.data_ptr(), wp.data_ptr(), unscale.data_ptr(), _stream_ptr(w)),
               "mvsgi_conv3d_wino32_pack_weights")
    return wp, unscale


F32P_MAX = 16376.0       # range of fp32-padded activations: a Winograd layer sums four of them in front of its fp16 split (65504 / 4)


def act_to_f32p(x_ndhwc: torch.Tensor, out: Optional[SplitAct] = None) -> SplitAct:
    """fp32 NDHWC -> "fp32-padded": the split-padded geometry with plain fp32 records (tests / tools; torch copies)."""
    x = _dev(x_ndhwc, "x")
    B, D, Hh, W, C = x.shape
    y = out if out is not None else SplitAct(B, D, Hh, W, C, x.device)
    y.buf.view(torch.float32)[:, 1:-1, 1:-1, 1:-1, :] = x.clamp(-F32P_MAX, F32P_MAX)      # the format's range (its writers clamp to it)
    y.fmt = "f32p"
    return y


def act_from_f32p(x: SplitAct) -> torch.Tensor:
    if x.fmt != "f32p":
        raise AssertionError(f"buffer holds {x.fmt} records")
    return x.buf.view(torch.float32)[:, 1:-1, 1:-1, 1:-1, :].contiguous()


def conv3d_wino(x: SplitAct, w_packed, scale, shift, res: Optional[SplitAct] = None, neg_slope=0.01, out=None, out_f32: bool = False,
                ):
    """Winograd-form 32 -> 32 conv (stride 1) on padded activations, arithmetic in the fp16 split; D == 8 or 16, H even, W % 32 == 0.
    x (and res) are split-padded fp16 pairs (fmt 'f16') or fp32-padded ('f32p': plain fp32 records in the same padded geometry --
    the cheaper hand-over between Winograd layers); the output takes x's format.  `out_f32`: the result is a plain fp32
    [B, D, H, W, 32] tensor instead of a SplitAct."""
    lib = _lib.load()
    if x.fmt not in ("f16", "f32p") or x.C != 32:
        raise AssertionError(f"conv3d_wino: needs a 32-channel input in the fp16 split or fp32-padded (got {x.C}, {x.fmt})")
    if out_f32:
        y = out if out is not None else torch.empty(x.shape, device=x.buf.device, dtype=torch.float32)
        if tuple(y.shape) != x.shape or not y.is_contiguous() or y.dtype != torch.float32:
            raise AssertionError(f"fp32 output {tuple(y.shape)} does not match {x.shape}")
        yp = y.data_ptr()
    else:
        y = out if out is not None else SplitAct(x.B, x.D, x.H, x.W, 32, x.buf.device)
        if y.shape != x.shape:
            raise AssertionError(f"split output {y.shape} does not match {x.shape}")
        yp = y.buf.data_ptr()
    if res is not None and (res.shape != x.shape or res.fmt != x.fmt):
        raise AssertionError(f"residual {res.shape} ({res.fmt}) does not match the input ({x.fmt})")
    if w_packed.numel() != lib.mvsgi_conv3d_wino32_packed_weight_bytes():
        raise AssertionError("conv3d_wino: packed weights of the wrong size")
    _lib.check(lib.mvsgi_conv3d_wino32_f16(x.buf.data_ptr(), w_packed.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                           None if res is None else res.buf.data_ptr(), yp, int(out_f32), int(x.fmt == "f32p"),
                                           x.B, x.D, x.H, x.W, float(neg_slope), _stream_ptr(x.buf)), "mvsgi_conv3d_wino32_f16")
    if not out_f32:
        y.fmt = x.fmt
    return y


def conv3d_out_split(x, w_packed_b3, scale, shift, out: "SplitAct", res=None, stride=1, neg_slope=0.01, fmt: str = "bf16") -> "SplitAct":
    """mvsgi_conv3d_f32 (streaming split kernel) with the output written split-padded into `out`; fmt = 'f16': weights packed by
    pack_conv_weights_f16x3 (scale carrying the unscale) and the output in the fp16 split."""
    lib = _lib.load()
    x = _dev(x, "x")
    B, Din, Hin, Win, Cin = x.shape
    Cout = scale.numel()
    Do, Ho, Wo = (Din - 1) // stride + 1, (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    if out.shape != (B, Do, Ho, Wo, Cout):
        raise AssertionError(f"split output {out.shape} does not match {(B, Do, Ho, Wo, Cout)}")
    if res is not None:
        res = _dev(res, "res")
    _lib.check(lib.mvsgi_conv3d_f32_out_split_fmt(x.data_ptr(), w_packed_b3.data_ptr(), scale.data_ptr(), shift.data_ptr(), _ptr(res),
                                                  out.buf.data_ptr(), B, Cin, Din, Hin, Win, Cout, stride, float(neg_slope),
                                                  _fmt_code(fmt), _stream_ptr(x)), "mvsgi_conv3d_f32_out_split")
    out.fmt = fmt
    return out


def conv3d_rs16(x: "SplitAct", w_packed_rs, scale, shift, neg_slope=0.01, out=None, out_split: Optional["SplitAct"] = None):
    """Register-stationary 16 -> 16 conv (post_vol) on a split-padded volume -> fp32 [B, D, H, W, 16], or (out_split) a
    split-padded volume of the same geometry (the hand-over to conv3d_s2rs)."""
    lib = _lib.load()
    if x.C != 16 or scale.numel() != 16:
        raise AssertionError("conv3d_rs16 is the 16 -> 16 kernel")
    if out_split is not None:
        if out_split.shape != x.shape or out_split.buf.data_ptr() == x.buf.data_ptr():
            raise AssertionError(f"split output {out_split.shape} must match the input {x.shape} and be another buffer")
        _lib.check(lib.mvsgi_conv3d_rs16_split_fmt(x.buf.data_ptr(), w_packed_rs.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                   out_split.buf.data_ptr(), 1, x.B, x.D, x.H, x.W, float(neg_slope), _fmt_code(x.fmt),
                                                   _stream_ptr(x.buf)), "mvsgi_conv3d_rs16_split_out_split")
        out_split.fmt = x.fmt
        return out_split
    y = out if out is not None else torch.empty((x.B, x.D, x.H, x.W, 16), device=x.buf.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_conv3d_rs16_split_fmt(x.buf.data_ptr(), w_packed_rs.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), 0,
                                               x.B, x.D, x.H, x.W, float(neg_slope), _fmt_code(x.fmt), _stream_ptr(x.buf)),
               "mvsgi_conv3d_rs16_split")
    return y


def conv3d_s2rs_applies(cin: int, cout: int, stride: int, neg_slope: float) -> bool:
    return cin == 16 and cout == 32 and stride == 2 and 0.0 <= neg_slope <= 1.0


def pack_conv_weights_s2rs(w_oidhw: torch.Tensor, scale: torch.Tensor, fmt: str = "bf16"):
    """[32, 16, 3, 3, 3] weights with the per-channel scale folded in, in the lane order of csrc/conv3d_s2rs.hip.
    fmt = 'f16': -> (packed weights, up, unscale): the kernel's epilogue has no per-channel multiplier, so ONE power of two `up` for
    the layer (the largest |weight * scale| in (512, 1024]) is folded into the packed weights; the caller multiplies `shift` by `up`
    and passes `unscale` = 1 / up to conv3d_s2rs."""
    lib = _lib.load()
    w = _dev(w_oidhw, "w")
    scale = _dev(scale, "scale")
    if tuple(w.shape) != (32, 16, 3, 3, 3) or scale.numel() != 32:
        raise AssertionError(f"conv3d_s2rs is the 16 -> 32 channel stride-2 layer, got weights {tuple(w.shape)}")
    up = un = None
    if fmt == "f16":
        up_t, un_t = _pow2_unscale((w.abs().amax(dim=(1, 2, 3, 4)) * scale.abs()).max().reshape(1))
        up, un = float(up_t[0]), float(un_t[0])
        scale = (scale * up).contiguous()
    wp = torch.empty(lib.mvsgi_conv3d_s2rs_packed_weight_bytes(), device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv3d_s2rs_pack_weights_fmt(w.data_ptr(), scale.data_ptr(), wp.data_ptr(), _fmt_code(fmt), _stream_ptr(w)),
               "mvsgi_conv3d_s2rs_pack_weights")
    return (wp, up, un) if fmt == "f16" else wp


def conv3d_s2rs(x: "SplitAct", w_packed, shift, out: "SplitAct", neg_slope=0.01, unscale: float = 1.0, out_f32p: bool = False) -> "SplitAct":
    """16 -> 32 channel 3x3x3 stride-2 conv + scale / shift + LeakyReLU, split-padded in and out (LDS-DMA staging).
    `out_f32p` (fp16 split): the output is fp32-padded (plain fp32 records, fmt 'f32p') for a Winograd-form level 0 behind it."""
    lib = _lib.load()
    Do, Ho, Wo = (x.D - 1) // 2 + 1, (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
    if x.C != 16 or out.shape != (x.B, Do, Ho, Wo, 32) or shift.numel() != 32:
        raise AssertionError(f"conv3d_s2rs: input {x.shape} -> output {out.shape}, expected {(x.B, Do, Ho, Wo, 32)}")
    _lib.check(lib.mvsgi_conv3d_s2rs_out_fmt(x.buf.data_ptr(), w_packed.data_ptr(), shift.data_ptr(), out.buf.data_ptr(), x.B, x.D, x.H, x.W,
                                             float(neg_slope), float(unscale), _fmt_code(x.fmt), int(out_f32p), _stream_ptr(x.buf)),
               "mvsgi_conv3d_s2rs")
    out.fmt = "f32p" if out_f32p else x.fmt
    return out


def conv3d_rs_applies(cin: int, cout: int, stride: int, neg_slope: float) -> bool:
    return cin == 32 and cout == 32 and stride == 1 and 0.0 <= neg_slope <= 1.0


def pack_conv2d_weights_bf16x3(w_oihw: torch.Tensor) -> Optional[torch.Tensor]:
    """[Cout, Cin, 3, 3] -> split-bf16 MFMA layout, or None when unsupported."""
    lib = _lib.load()
    w = _dev(w_oihw, "conv weight")
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3) or Cin % 16 or Cout % 16:
        return None
    wp = torch.empty(lib.mvsgi_conv2d_packed_weight_bytes_bf16x3(Cout, Cin), device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv2d_pack_weights_bf16x3(w.data_ptr(), wp.data_ptr(), Cout, Cin, _stream_ptr(w)),
               "mvsgi_conv2d_pack_weights_bf16x3")
    return wp


def pack_conv2d_stem_weights(w_oihw: torch.Tensor) -> Optional[torch.Tensor]:
    """[16, 3, 5, 5] RGB-stem weights -> the matrix-core layout for uint8 images (w / 255 in three bf16 pieces), or None."""
    lib = _lib.load()
    w = _dev(w_oihw, "conv weight")
    if tuple(w.shape) != (16, 3, 5, 5):
        return None
    wp = torch.empty(lib.mvsgi_conv2d_stem_packed_weight_bytes(), device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_conv2d_stem_pack_weights(w.data_ptr(), wp.data_ptr(), _stream_ptr(w)), "mvsgi_conv2d_stem_pack_weights")
    return wp


def pack_conv2d_weights_f32(w_oihw: torch.Tensor) -> Optional[torch.Tensor]:
    """[Cout, Cin, 3, 3] -> exact-fp32 MFMA layout (Cout in {16, 32}), or None when unsupported."""
    lib = _lib.load()
    w = _dev(w_oihw, "conv weight")
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3) or Cin % 16 or Cout not in (16, 32):
        return None
    wp = torch.empty(lib.mvsgi_conv2d_packed_weight_floats(Cout, Cin), device=w.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_conv2d_pack_weights_f32(w.data_ptr(), wp.data_ptr(), Cout, Cin, _stream_ptr(w)),
               "mvsgi_conv2d_pack_weights_f32")
    return wp


def conv2d(x, w_oihw, w_packed, scale, shift, res=None, stride=1, neg_slope=0.01, impl=CONV_AUTO, in_nchw=False, out_split=None):
    """x [B, H, W, Cin] (or [B, Cin, H, W] with in_nchw) -> y [B, Ho, Wo, Cout] = act(conv(x)*scale + shift (+res)).
    out_split: a zero-bordered 2-D split-padded buffer (split2d_buffer) that receives the output instead (Cout == 16)."""
    lib = _lib.load()
    layout = int(bool(in_nchw))
    if x.dtype == torch.uint8:                 # camera images [B, H, W, 3]: converted (/255) inside the stem kernel
        x = _dev(x, "imgs", torch.uint8)
        B, Hin, Win, Cin = x.shape
        layout = 2
    else:
        x = _dev(x, "x")
        if in_nchw:
            B, Cin, Hin, Win = x.shape
        else:
            B, Hin, Win, Cin = x.shape
    Cout, k = scale.numel(), int(w_oihw.shape[-1])
    pad = k // 2
    Ho, Wo = (Hin + 2 * pad - k) // stride + 1, (Win + 2 * pad - k) // stride + 1
    if res is not None:
        res = _dev(res, "res")
        if tuple(res.shape) != (B, Ho, Wo, Cout):
            raise AssertionError(f"residual {tuple(res.shape)} does not match output {(B, Ho, Wo, Cout)}")
    if out_split is not None:
        _check_split2d(out_split, B, Ho, Wo, x.device)
        _lib.check(lib.mvsgi_conv2d_f32_out_split2d(x.data_ptr(), _ptr(w_oihw), _ptr(w_packed), scale.data_ptr(), shift.data_ptr(),
                                                    _ptr(res), out_split.data_ptr(), B, Cin, Hin, Win, Cout, k, stride,
                                                    float(neg_slope), impl, layout, _stream_ptr(x)), "mvsgi_conv2d_f32_out_split2d")
        return out_split
    y = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_conv2d_f32(x.data_ptr(), _ptr(w_oihw), _ptr(w_packed), scale.data_ptr(), shift.data_ptr(),
                                    _ptr(res), y.data_ptr(), B, Cin, Hin, Win, Cout, k, stride, float(neg_slope), impl,
                                    layout, _stream_ptr(x)), "mvsgi_conv2d_f32")
    return y


def conv2d_variant(Cin, Cout, k=3, stride=1, impl=CONV_AUTO, in_nchw=False) -> str:
    lib = _lib.load()
    name = lib.mvsgi_conv2d_variant_f32(Cin, Cout, k, stride, impl, int(bool(in_nchw)))
    if name is None:
        raise RuntimeError("mvsgi_conv2d_variant_f32: " + lib.mvsgi_last_error().decode())
    return name.decode()


def resize_trilinear(x, size) -> torch.Tensor:
    lib = _lib.load()
    x = _dev(x, "x")
    B, Di, Hi, Wi, C = x.shape
    Do, Ho, Wo = (int(s) for s in size)
    y = torch.empty((B, Do, Ho, Wo, C), device=x.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_resize_trilinear_f32(x.data_ptr(), y.data_ptr(), B, C, Di, Hi, Wi, Do, Ho, Wo,
                                              _stream_ptr(x)), "mvsgi_resize_trilinear_f32")
    return y


def softargmin(costs_bdhw, inv_idx, scale: int, want_norm_costs: bool, post_div: float = 1.0):
    """costs [B, D, H, W], inv_idx [D] -> inv_dist [B, 1, sH, sW] (divided by post_div), norm_costs
    [B, D, sH, sW] | None."""
    lib = _lib.load()
    c = _dev(costs_bdhw, "costs")
    inv_idx = _dev(inv_idx.reshape(-1), "inv_dist_idx")
    B, D, H, W = c.shape
    if inv_idx.numel() != D:
        raise AssertionError(f"{D} cost planes but {inv_idx.numel()} distance candidates")
    inv = torch.empty((B, 1, H * scale, W * scale), device=c.device, dtype=torch.float32)
    pr = torch.empty((B, D, H * scale, W * scale), device=c.device, dtype=torch.float32) if want_norm_costs else None
    _lib.check(lib.mvsgi_softargmin_div_f32(c.data_ptr(), inv_idx.data_ptr(), inv.data_ptr(), _ptr(pr), B, D, H, W,
                                            scale, float(post_div), _stream_ptr(c)), "mvsgi_softargmin_div_f32")
    return inv, pr


def ncdhw_to_ndhwc(x) -> torch.Tensor:
    """contiguous [B, C, D, H, W] -> [B, D, H, W, C]."""
    lib = _lib.load()
    x = _dev(x, "x")
    B, C, D, H, W = x.shape
    y = torch.empty((B, D, H, W, C), device=x.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_ncv_to_nvc_f32(x.data_ptr(), y.data_ptr(), B, C, D * H * W, _stream_ptr(x)),
               "mvsgi_ncv_to_nvc_f32")
    return y


def ndhwc_to_ncdhw(x) -> torch.Tensor:
    lib = _lib.load()
    x = _dev(x, "x")
    B, D, H, W, C = x.shape
    y = torch.empty((B, C, D, H, W), device=x.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_nvc_to_ncv_f32(x.data_ptr(), y.data_ptr(), B, C, D * H * W, _stream_ptr(x)),
               "mvsgi_nvc_to_ncv_f32")
    return y


def as_ndhwc(vol: torch.Tensor) -> torch.Tensor:
    """Accepts the module-boundary volume [B, C, D, H, W] in either memory format and returns the
    [B, D, H, W, C] view (no copy when it already is channels-last, which is what our
    cv_builder hands over) or a transposed copy (contiguous NCDHW from another producer)."""
    if vol.dim() != 5:
        raise AssertionError(f"expected a [B, C, D, H, W] volume, got {tuple(vol.shape)}")
    v = vol.permute(0, 2, 3, 4, 1)
    if v.is_contiguous():
        return v
    return ncdhw_to_ndhwc(vol.contiguous())


# --------------------------------------------------------------------------------------
def pack_deform_conv2d_weights(w_oihw: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, Kh, Kw] -> [Kh*Kw, Cin, Cout] for mvsgi_deform_conv2d_f32."""
    lib = _lib.load()
    w = _dev(w_oihw, "deform conv weight")
    Cout, Cin, Kh, Kw = w.shape
    wp = torch.empty((Kh * Kw, Cin, Cout), device=w.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_deform_conv2d_pack_weights_f32(w.data_ptr(), wp.data_ptr(), Cout, Cin, Kh, Kw, _stream_ptr(w)),
               "mvsgi_deform_conv2d_pack_weights_f32")
    return wp


def deform_conv2d(x_nhwc, offset, w_packed, scale, shift, kernel_size, stride=(1, 1), padding=(0, 0), dilation=(1, 1),
                  res=None, neg_slope=1.0) -> torch.Tensor:
    """x [N, H, W, Cin], offset [1 | N, 2*Kh*Kw, Ho, Wo] -> y [N, Ho, Wo, Cout] (torchvision.ops.deform_conv2d
    semantics + per-channel scale / shift, residual, LeakyReLU)."""
    lib = _lib.load()
    x = _dev(x_nhwc, "x")
    offset = _dev(offset, "offset")
    N, Hh, W, Cin = x.shape
    Kh, Kw = kernel_size
    Cout = scale.numel()
    Ho = (Hh + 2 * padding[0] - (dilation[0] * (Kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (Kw - 1) + 1)) // stride[1] + 1
    if offset.dim() != 4 or tuple(offset.shape[1:]) != (2 * Kh * Kw, Ho, Wo) or offset.shape[0] not in (1, N):
        raise AssertionError(f"offset {tuple(offset.shape)} does not match [1|{N}, {2 * Kh * Kw}, {Ho}, {Wo}]")
    if tuple(w_packed.shape) != (Kh * Kw, Cin, Cout):
        raise AssertionError(f"w_packed {tuple(w_packed.shape)} does not match {(Kh * Kw, Cin, Cout)}")
    if res is not None:
        res = _dev(res, "res")
        if tuple(res.shape) != (N, Ho, Wo, Cout):
            raise AssertionError(f"residual {tuple(res.shape)} does not match output {(N, Ho, Wo, Cout)}")
    y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_deform_conv2d_f32(x.data_ptr(), offset.data_ptr(), int(offset.shape[0] == N and N > 1),
                                           w_packed.data_ptr(), scale.data_ptr(), shift.data_ptr(), _ptr(res),
                                           y.data_ptr(), N, Cin, Hh, W, Cout, Kh, Kw, stride[0], stride[1], padding[0],
                                           padding[1], dilation[0], dilation[1], float(neg_slope), _stream_ptr(x)),
               "mvsgi_deform_conv2d_f32")
    return y


# ---- the extractor's residual blocks on pre-split activations (csrc/resblock2d_rs.hip) ----
def split2d_buffer(N: int, H: int, W: int, device) -> torch.Tensor:
    """A zeroed 2-D split-padded activation buffer [N, H + 4, W + 4, 64] uint8 (the kernels write the interior only: the
    two-pixel zero border is the convolutions' padding and stays as allocated)."""
    return torch.zeros((N, H + 4, W + 4, 64), device=device, dtype=torch.uint8)


def _check_split2d(t: torch.Tensor, N: int, H: int, W: int, device) -> None:
    if t.dtype != torch.uint8 or tuple(t.shape) != (N, H + 4, W + 4, 64) or not t.is_contiguous() or t.device != device:
        raise AssertionError(f"split-padded 2-D buffer {tuple(t.shape)} {t.dtype} {t.device} does not match {(N, H + 4, W + 4, 64)} uint8 on {device}")


def f32_to_split2d(x_nhwc: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    x = _dev(x_nhwc, "x")
    N, Hh, W, C = x.shape
    if C != 16:
        raise AssertionError(f"the 2-D split-padded format holds 16 channels, got {tuple(x.shape)}")
    if out is None:
        out = split2d_buffer(N, Hh, W, x.device)
    _check_split2d(out, N, Hh, W, x.device)
    _lib.check(lib.mvsgi_f32_to_split2d(x.data_ptr(), out.data_ptr(), N, Hh, W, _stream_ptr(x)), "mvsgi_f32_to_split2d")
    return out


def split2d_to_f32(x_split: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    N, Hp, Wp, _ = x_split.shape
    _check_split2d(x_split, N, Hp - 4, Wp - 4, x_split.device)
    y = torch.empty((N, Hp - 4, Wp - 4, 16), device=x_split.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_split2d_to_f32(x_split.data_ptr(), y.data_ptr(), N, Hp - 4, Wp - 4, _stream_ptr(x_split)), "mvsgi_split2d_to_f32")
    return y


def pack_resblock2d_split_weights(w_oihw: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """Lane-ordered split-bf16 weights of one conv of the block with the BatchNorm scale folded in."""
    lib = _lib.load()
    w = _dev(w_oihw, "w")
    scale = _dev(scale, "scale")
    if tuple(w.shape) != (16, 16, 3, 3) or scale.numel() != 16:
        raise AssertionError(f"resblock2d_split is the 16 -> 16 channel 3x3 block, got weights {tuple(w.shape)}")
    wp = torch.empty(lib.mvsgi_resblock2d_split_packed_weight_bytes(), device=w.device, dtype=torch.uint8)
    _lib.check(lib.mvsgi_resblock2d_split_pack_weights(w.data_ptr(), scale.data_ptr(), wp.data_ptr(), _stream_ptr(w)),
               "mvsgi_resblock2d_split_pack_weights")
    return wp


def resblock2d_split(x_split, wp1, shift1, wp2, shift2, neg_slope=0.01, out_split=None) -> torch.Tensor:
    """Fused 16 -> 16 residual block on a 2-D split-padded input [N, H + 4, W + 4, 64] uint8 (wp1 / wp2 carry the scales).
    out_split: the split-padded buffer that receives the output (returned); None: the output is a plain fp32 [N, H, W, 16]
    tensor."""
    lib = _lib.load()
    N, Hp, Wp, _ = x_split.shape
    Hh, W = Hp - 4, Wp - 4
    _check_split2d(x_split, N, Hh, W, x_split.device)
    if shift1.numel() != 16 or shift2.numel() != 16:
        raise AssertionError("resblock2d_split is the 16-channel block")
    if out_split is not None:
        _check_split2d(out_split, N, Hh, W, x_split.device)
        y = out_split
    else:
        y = torch.empty((N, Hh, W, 16), device=x_split.device, dtype=torch.float32)
    _lib.check(lib.mvsgi_resblock2d_split(x_split.data_ptr(), wp1.data_ptr(), shift1.data_ptr(), wp2.data_ptr(), shift2.data_ptr(),
                                          y.data_ptr(), int(out_split is not None), N, Hh, W, float(neg_slope),
                                          _stream_ptr(x_split)), "mvsgi_resblock2d_split")
    return y


def conv2d_s2_split(x_split, w_packed, shift, out_split, neg_slope=0.01) -> torch.Tensor:
    """16 -> 16 channel 3x3 stride-2 layer on 2-D split-padded activations (weights from pack_resblock2d_split_weights)."""
    lib = _lib.load()
    N, Hp, Wp, _ = x_split.shape
    Hh, W = Hp - 4, Wp - 4
    _check_split2d(x_split, N, Hh, W, x_split.device)
    _check_split2d(out_split, N, (Hh - 1) // 2 + 1, (W - 1) // 2 + 1, x_split.device)
    if shift.numel() != 16:
        raise AssertionError("conv2d_s2_split is the 16-channel layer")
    _lib.check(lib.mvsgi_conv2d_s2_split(x_split.data_ptr(), w_packed.data_ptr(), shift.data_ptr(), out_split.data_ptr(), N, Hh, W,
                                         float(neg_slope), _stream_ptr(x_split)), "mvsgi_conv2d_s2_split")
    return out_split


def resblock2d(x_nhwc, wp1, scale1, shift1, wp2, scale2, shift2, neg_slope=0.01) -> torch.Tensor:
    """Fused 16 -> 16 residual block (two 3x3 convs + BN + LeakyReLU + skip): x [N, H, W, 16] -> y, same shape."""
    lib = _lib.load()
    x = _dev(x_nhwc, "x")
    N, Hh, W, C = x.shape
    if C != 16 or scale1.numel() != 16 or scale2.numel() != 16:
        raise AssertionError(f"resblock2d is the 16-channel block, got x {tuple(x.shape)}")
    y = torch.empty_like(x)
    _lib.check(lib.mvsgi_resblock2d_f32(x.data_ptr(), wp1.data_ptr(), scale1.data_ptr(), shift1.data_ptr(),
                                        wp2.data_ptr(), scale2.data_ptr(), shift2.data_ptr(), y.data_ptr(), N, Hh, W,
                                        float(neg_slope), _stream_ptr(x)), "mvsgi_resblock2d_f32")
    return y
