// The fp16-split ("f16x3") instantiations of the streaming split kernel (conv3d_bf16x3.hpp): the same variants, bricks, schedules
// and packed-weight layouts as the bf16 split, with hi = fp16(x), lo = fp16(x - hi) and v_mfma_f32_16x16x32_f16 /
// v_mfma_f32_32x32x16_f16 -- 11 + 11 significant bits per operand instead of 8 + 8 at the same matrix rate.  A translation unit of its
// own so that the two arithmetics compile side by side.  Entered from conv3d.hip's launch_variant() when ConvArgs::f16 is set
// (impl | MVSGI_CONV_F16 at the C ABI); replaces the same reference ops as conv3d.hip (BaseConvBlk3d / ResizeConv3d forward,
// dsta_mvs/model/common/common_modules.py:107-115, 332-355).
#include "common.hpp"
#include "conv_common.hpp"
#include "conv3d_variants.hpp"
#ifdef MVSGI_STAMPS
#include <cstdio>
#include <cstdlib>
#endif

namespace {
#include "conv3d_bf16x3.hpp"
}  // namespace

namespace mvsgi {

int conv3d_launch_b3_f16(int variant, const void* args, hipStream_t st) {
    const ConvArgs& a = *static_cast<const ConvArgs*>(args);
    switch (variant) {
#define MVSGI_B3(V, ...) case V: return launch_bf16x3<__VA_ARGS__, true>(a, st);
#include "conv3d_b3_variants.inc"
#undef MVSGI_B3
        case B3_N16_T: return launch_bf16x3<1, 1, 4, 1, 1, 4, 16, 1, 3, false, false, false, true, true>(a, st);
#define MVSGI_B3D(V, ...) case V: return launch_bf16x3<__VA_ARGS__, 1, 3, false, false, false, false, true, true>(a, st);
#define MVSGI_B3DK(V, ...) case V: return launch_bf16x3<__VA_ARGS__, 1, 3, false, false, false, false, true, true, true>(a, st);
#define MVSGI_B3DK2(V, ...) case V: return launch_bf16x3<__VA_ARGS__, 1, 3, false, false, false, false, true, true, true, 3>(a, st);
#include "conv3d_b3d_variants.inc"
#undef MVSGI_B3D
#undef MVSGI_B3DK
#undef MVSGI_B3DK2
#define MVSGI_B3DU(V, ...) case V: return launch_bf16x3<__VA_ARGS__, 1, 3, true, false, false, false, true, true>(a, st);
#define MVSGI_B3DUK(V, ...) case V: return launch_bf16x3<__VA_ARGS__, 1, 3, true, false, false, false, true, true, true>(a, st);
#include "conv3d_b3du_variants.inc"
#undef MVSGI_B3DU
#undef MVSGI_B3DUK
    }
    return fail("mvsgi_conv3d_f32: variant %d has no fp16-split form (MVSGI_CONV_F16 goes with MVSGI_CONV_BF16X3 / _C16 / _V32)", variant);
}

const char* conv3d_b3_f16_name(int variant) {
    switch (variant) {
#define MVSGI_B3(V, ...) case V: return "conv3d_f16x3_kernel<" #__VA_ARGS__ ">";
#include "conv3d_b3_variants.inc"
#undef MVSGI_B3
        case B3_N16_T: return "conv3d_f16x3_kernel<1, 1, 4, 1, 1, 4, 16, 1, 3, false, false, false, true>";
#define MVSGI_B3D(V, ...) case V: return "conv3d_f16x3_d32_kernel<" #__VA_ARGS__ ">";
#define MVSGI_B3DK(V, ...) case V: return "conv3d_f16x3_d32_dk_kernel<" #__VA_ARGS__ ">";
#define MVSGI_B3DK2(V, ...) case V: return "conv3d_f16x3_d32_dk2_kernel<" #__VA_ARGS__ ">";
#include "conv3d_b3d_variants.inc"
#undef MVSGI_B3D
#undef MVSGI_B3DK
#undef MVSGI_B3DK2
#define MVSGI_B3DU(V, ...) case V: return "conv3d_f16x3_d32u_kernel<" #__VA_ARGS__ ">";
#define MVSGI_B3DUK(V, ...) case V: return "conv3d_f16x3_d32u_dk_kernel<" #__VA_ARGS__ ">";
#include "conv3d_b3du_variants.inc"
#undef MVSGI_B3DU
#undef MVSGI_B3DUK
    }
    return nullptr;
}

}  // namespace mvsgi
