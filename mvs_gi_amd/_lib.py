"""ctypes binding of libmvsgi_hip.so (include/mvsgi.h).

The library is the product path: there is no CPU or PyTorch fallback.  If the shared
object has not been built (`python -c "import __graft_entry__ as g; g.build()"` or
`make -C mvs_gi_amd/csrc`) every op raises MvsgiLibraryMissing.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVSGI_LIB", os.path.join(_HERE, "libmvsgi_hip.so"))   # MVSGI_LIB: diagnostic builds

ABI_VERSION = 3


class MvsgiLibraryMissing(RuntimeError):
    pass


# name -> (restype, argtypes); must list every symbol declared in include/mvsgi.h
_P = c_void_p
SIGNATURES = {
    "mvsgi_abi_version": (c_int, []),
    "mvsgi_last_error": (c_char_p, []),
    "mvsgi_saturation_flags": (c_int, [c_int, _P]),
    "mvsgi_saturation_words": (c_int, [_P]),
    "mvsgi_sweep_std_f32": (c_int, [_P, _P, _P, c_int, _P, _P] + [c_int] * 10 + [_P]),
    "mvsgi_sweep_cat_f32": (c_int, [_P, _P, _P] + [c_int] * 8 + [_P]),
    "mvsgi_sweep_std_nhwc_f32": (c_int, [_P, _P, _P, c_int, _P, _P] + [c_int] * 10 + [_P]),
    "mvsgi_sweep_cat_nhwc_f32": (c_int, [_P, _P, _P] + [c_int] * 8 + [_P]),
    "mvsgi_sweep_validity_u8": (c_int, [_P, _P, c_int, _P, _P] + [c_int] * 7 + [_P]),
    "mvsgi_sweep_std_nhwc_valid_f32": (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "mvsgi_sweep_std_nhwc_valid_split": (c_int, [_P, _P, _P, _P] + [c_int] * 9 + [_P]),
    "mvsgi_sweep_std_nhwc_valid_rig_f32": (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "mvsgi_conv3d_packed_weight_floats": (c_size_t, [c_int, c_int]),
    "mvsgi_conv3d_pack_weights_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "mvsgi_conv3d_packed_weight_bytes_bf16x3": (c_size_t, [c_int, c_int]),
    "mvsgi_conv3d_pack_weights_bf16x3": (c_int, [_P, _P, c_int, c_int, _P]),
    "mvsgi_conv3d_pack_weights_split": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "mvsgi_conv3d_f32": (c_int, [_P] * 7 + [c_int] * 7 + [c_float, c_int, _P]),
    "mvsgi_conv3d_variant_f32": (c_char_p, [c_int] * 8),
    "mvsgi_conv3d_up2_f32": (c_int, [_P, _P, c_int] + [_P] * 4 + [c_int] * 6 + [c_float, _P]),
    "mvsgi_conv3d_up2_variant_f32": (c_char_p, [c_int] * 7),
    "mvsgi_conv3d_v32_applies": (c_int, [c_int] * 7),
    "mvsgi_conv3d_d32_applies": (c_int, [c_int] * 7),
    "mvsgi_conv3d_up2_d32_applies": (c_int, [c_int] * 6),
    "mvsgi_conv3d_packed_weight_bytes_bf16x3_v32": (c_size_t, [c_int, c_int]),
    "mvsgi_conv3d_pack_weights_bf16x3_v32": (c_int, [_P, _P, c_int, c_int, _P]),
    "mvsgi_conv3d_packed_weight_bytes_bf16x3_c16": (c_size_t, [c_int]),
    "mvsgi_conv3d_pack_weights_bf16x3_c16": (c_int, [_P, _P, c_int, _P]),
    "mvsgi_conv2d_stem_packed_weight_bytes": (c_size_t, []),
    "mvsgi_conv2d_stem_pack_weights": (c_int, [_P, _P, _P]),
    "mvsgi_conv2d_packed_weight_floats": (c_size_t, [c_int, c_int]),
    "mvsgi_conv2d_pack_weights_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "mvsgi_conv2d_packed_weight_bytes_bf16x3": (c_size_t, [c_int, c_int]),
    "mvsgi_conv2d_pack_weights_bf16x3": (c_int, [_P, _P, c_int, c_int, _P]),
    "mvsgi_conv2d_f32": (c_int, [_P] * 7 + [c_int] * 7 + [c_float, c_int, c_int, _P]),
    "mvsgi_conv2d_variant_f32": (c_char_p, [c_int] * 6),
    "mvsgi_resblock2d_f32": (c_int, [_P] * 8 + [c_int] * 3 + [c_float, _P]),
    "mvsgi_split2d_bytes": (c_size_t, [c_int] * 3),
    "mvsgi_f32_to_split2d": (c_int, [_P, _P] + [c_int] * 3 + [_P]),
    "mvsgi_split2d_to_f32": (c_int, [_P, _P] + [c_int] * 3 + [_P]),
    "mvsgi_resblock2d_split_packed_weight_bytes": (c_size_t, []),
    "mvsgi_resblock2d_split_pack_weights": (c_int, [_P, _P, _P, _P]),
    "mvsgi_resblock2d_split": (c_int, [_P] * 6 + [c_int] * 4 + [c_float, _P]),
    "mvsgi_conv2d_s2_split": (c_int, [_P] * 4 + [c_int] * 3 + [c_float, _P]),
    "mvsgi_conv2d_f32_out_split2d": (c_int, [_P] * 7 + [c_int] * 7 + [c_float, c_int, c_int, _P]),
    "mvsgi_resize_trilinear_f32": (c_int, [_P, _P] + [c_int] * 8 + [_P]),
    "mvsgi_softargmin_f32": (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [_P]),
    "mvsgi_softargmin_div_f32": (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [c_float, _P]),
    "mvsgi_rays_panorama_f32": (c_int, [_P, _P, c_int, c_int, c_int] + [c_float] * 4 + [_P]),
    "mvsgi_transform_points_f32": (c_int, [_P, _P, _P, c_int, c_longlong, _P]),
    "mvsgi_grid_double_sphere_f32": (c_int, [_P, _P, _P, c_int, c_longlong] + [c_float] * 6 + [c_int, c_int, c_float, _P]),
    "mvsgi_grid_equirect_f32": (c_int, [_P, _P, c_int, c_longlong, _P]),
    "mvsgi_deform_conv2d_pack_weights_f32": (c_int, [_P, _P] + [c_int] * 4 + [_P]),
    "mvsgi_deform_conv2d_f32": (c_int, [_P, _P, c_int] + [_P] * 5 + [c_int] * 13 + [c_float, _P]),
    "mvsgi_act_split_bytes": (c_size_t, [c_int] * 5),
    "mvsgi_act_f32_to_split": (c_int, [_P, _P] + [c_int] * 5 + [_P]),
    "mvsgi_act_split_to_f32": (c_int, [_P, _P] + [c_int] * 5 + [_P]),
    "mvsgi_conv3d_f32_out_split": (c_int, [_P] * 6 + [c_int] * 7 + [c_float, _P]),
    "mvsgi_conv3d_rs16_split": (c_int, [_P] * 5 + [c_int] * 4 + [c_float, _P]),
    "mvsgi_conv3d_rs16_split_out_split": (c_int, [_P] * 5 + [c_int] * 4 + [c_float, _P]),
    "mvsgi_conv3d_s2rs_packed_weight_bytes": (c_size_t, []),
    "mvsgi_conv3d_s2rs_pack_weights": (c_int, [_P, _P, _P, _P]),
    "mvsgi_conv3d_s2rs": (c_int, [_P] * 4 + [c_int] * 4 + [c_float, _P]),
    "mvsgi_conv3d_rs_packed_weight_bytes": (c_size_t, [c_int, c_int]),
    "mvsgi_conv3d_rs_pack_weights": (c_int, [_P, _P, c_int, c_int, _P]),
    "mvsgi_conv3d_rs_split": (c_int, [_P] * 6 + [c_int] * 7 + [c_float, _P]),
    "mvsgi_conv3d_up2_f32_out_split": (c_int, [_P, _P, c_int] + [_P] * 4 + [c_int] * 6 + [c_float, _P]),
    "mvsgi_conv3d_up2_poly_split": (c_int, [_P] * 5 + [c_int] * 4 + [c_float, _P]),
    "mvsgi_conv3d_head_split_packed_weight_bytes": (c_size_t, [c_int]),
    "mvsgi_conv3d_head_split_pack_weights": (c_int, [_P, _P, c_int, _P]),
    "mvsgi_conv3d_head_split": (c_int, [_P, _P, c_float, c_float, _P] + [c_int] * 5 + [c_float, _P]),
    "mvsgi_conv3d_head_split_pack_weights_f16": (c_int, [_P, _P, c_int, _P]),
    "mvsgi_conv3d_head_split_f16": (c_int, [_P, _P, c_float, c_float, _P] + [c_int] * 5 + [c_float, _P]),
    "mvsgi_conv3d_up2_poly_plan_bytes": (c_size_t, [c_int] * 3),
    "mvsgi_conv3d_up2_poly_plan": (c_int, [_P, _P] + [c_int] * 3),
    "mvsgi_conv3d_up2_poly_f32": (c_int, [_P] * 5 + [c_int] * 4 + [c_float, _P]),
    "mvsgi_act_f32_to_split_fmt": (c_int, [_P, _P] + [c_int] * 6 + [_P]),
    "mvsgi_act_split_to_f32_fmt": (c_int, [_P, _P] + [c_int] * 6 + [_P]),
    "mvsgi_sweep_std_nhwc_valid_split_fmt": (c_int, [_P, _P, _P, _P] + [c_int] * 10 + [_P]),
    "mvsgi_conv3d_f32_out_split_fmt": (c_int, [_P] * 6 + [c_int] * 7 + [c_float, c_int, _P]),
    "mvsgi_conv3d_rs_pack_weights_fmt": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "mvsgi_conv3d_rs_split_fmt": (c_int, [_P] * 6 + [c_int] * 7 + [c_float, c_int, _P]),
    "mvsgi_conv3d_rs16_split_fmt": (c_int, [_P] * 5 + [c_int] * 5 + [c_float, c_int, _P]),
    "mvsgi_conv3d_s2rs_pack_weights_fmt": (c_int, [_P, _P, _P, c_int, _P]),
    "mvsgi_conv3d_s2rs_fmt": (c_int, [_P] * 4 + [c_int] * 4 + [c_float, c_float, c_int, _P]),
    "mvsgi_conv3d_s2rs_out_fmt": (c_int, [_P] * 4 + [c_int] * 4 + [c_float, c_float, c_int, c_int, _P]),
    "mvsgi_conv3d_up2_poly_plan_fmt": (c_int, [_P, _P] + [c_int] * 4),
    "mvsgi_conv3d_wino32_packed_weight_bytes": (c_size_t, []),
    "mvsgi_conv3d_wino32_applies": (c_int, [c_int] * 6 + [c_float]),
    "mvsgi_conv3d_wino32_pack_weights": (c_int, [_P] * 4),
    "mvsgi_conv3d_wino32_f16": (c_int, [_P] * 6 + [c_int] * 6 + [c_float, _P]),
    "mvsgi_conv3d_up2_poly_fmt": (c_int, [_P] * 5 + [c_int] * 5 + [c_float, c_int, _P]),
    "mvsgi_conv3d_up2_poly_wino_pays": (c_int, [c_int] * 4),
    "mvsgi_ncv_to_nvc_f32": (c_int, [_P, _P, c_int, c_int, c_longlong, _P]),
    "mvsgi_nvc_to_ncv_f32": (c_int, [_P, _P, c_int, c_int, c_longlong, _P]),
}

_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise MvsgiLibraryMissing(
            f"{LIB_PATH} not found: build the HIP library first (__graft_entry__.build()). "
            "mvs_gi_amd has no CPU fallback.")
    # torch first: libmvsgi_hip.so links libamdhip64 by SONAME and must bind to the HIP runtime torch has
    # loaded (its own copy); loaded the other way round the process ends up with two runtimes and every launch of
    # this library fails with "no ROCm-capable device is detected"
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    # the version first: a stale library (an old diagnostic build named by MVSGI_LIB, say) must fail with the rebuild hint, not
    # with an AttributeError on the first symbol it lacks
    try:
        lib.mvsgi_abi_version.restype = c_int
        lib.mvsgi_abi_version.argtypes = []
        got = lib.mvsgi_abi_version()
    except AttributeError as e:
        raise MvsgiLibraryMissing(f"{LIB_PATH}: not a libmvsgi_hip ({e}); rebuild it (__graft_entry__.build())") from None
    if got != ABI_VERSION:
        raise MvsgiLibraryMissing(f"{LIB_PATH}: ABI version {got}, expected {ABI_VERSION}; rebuild it (__graft_entry__.build())")
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise MvsgiLibraryMissing(f"{LIB_PATH}: symbol {name} missing although the ABI version matches; rebuild it "
                                      "(__graft_entry__.build())") from None
        fn.restype = res
        fn.argtypes = args
    # allocates the range report's pinned words now, so that no allocation falls into a stream capture (include/mvsgi.h).  Without a
    # usable device (the CPU-only symbol test) this fails and is asked again by the first launch, which then fails loudly.
    lib.mvsgi_saturation_flags(0, None)
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().mvsgi_last_error()
        raise RuntimeError(f"{what}: {msg.decode() if msg else 'unknown error'}")
