/*
 * mvsgi.h -- C ABI of the MI355X (gfx950) plane-sweep hot path of castacks/mvs_gi.
 *
 * The reference is pure Python/PyTorch and has no FFI of its own; the precedent for
 * handing this path raw device pointers is its TensorRT back end
 * (api/inference_trt.py:128-142, context.execute_async_v2 with device addresses).
 * Each entry point below replaces one stage of
 *   dsta_mvs/model/mvs_model/torch_only.py:32-34
 *     vol   = cv_builder(feats, grids, grid_masks, masks)
 *     costs = cv_regulator(vol)
 *     inv_dist, norm_costs = dist_regressor(costs)
 * and is what a ctypes / cgo / JNI binding on the reference side would bind
 * (INTEGRATION.md shows the ctypes stub that mvs_gi_amd/_lib.py uses).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM), caller-owned, never freed or retained;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work is
 *     enqueued on it and nothing synchronises the host;
 *   - activations are fp32, channels-last: [B][D][H][W][C] ("NDHWC"); the 2-D inputs of
 *     the sweep keep the reference's own layouts (stated per function);
 *   - return value: 0 = enqueued, non-zero = rejected (nothing enqueued);
 *     mvsgi_last_error() returns a thread-local description of the last failure
 *     (the Python layer raises RuntimeError with it, mirroring the reference's
 *     assert/exception behaviour, e.g. backports.py:32).
 *   - no global mutable state but the range report's sticky words (mvsgi_saturation_flags):
 *     every call is re-entrant per stream, one process per GPU.
 */
#ifndef MVSGI_H
#define MVSGI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVSGI_ABI_VERSION 3   /* 3: mvsgi_conv3d_d32_applies, mvsgi_conv3d_up2_d32_applies + MVSGI_CONV_BF16X3_D32; 2: mvsgi_saturation_flags; the symbol set of round 5 */

typedef void* mvsgi_stream_t;

/* conv3d implementation selector */
#define MVSGI_CONV_AUTO   0   /* MFMA implicit GEMM when Cin%16==0 && Cout%16==0, else direct */
#define MVSGI_CONV_DIRECT 1   /* VALU direct convolution (any channel counts; Cout==1 head)   */
#define MVSGI_CONV_MFMA   2   /* LDS-tiled: v_mfma_f32_16x16x4_f32 implicit GEMM (exact fp32),
                                 or the LDS-tiled VALU head kernel when Cout == 1               */
#define MVSGI_CONV_BF16X3 3   /* split-bf16 implicit GEMM on v_mfma_f32_16x16x32_bf16: x = hi + lo,
                                 hi*hi + hi*lo + lo*hi, fp32 accumulate (~2^-16 per product);
                                 w_packed must come from mvsgi_conv3d_pack_weights_bf16x3.
                                 Falls back to the exact paths for Cout == 1 / odd channel counts */
#define MVSGI_CONV_BF16X3_C16 4 /* the same arithmetic for Cout == 16, stride 1 (post_vol, out_costs.0): "plane"
                                   schedule -- one activation fragment feeds the kd = 0, 1, 2 taps of three output
                                   planes -- with weights from mvsgi_conv3d_pack_weights_bf16x3_c16            */
#define MVSGI_CONV_BF16X3_V32 5 /* the same arithmetic on v_mfma_f32_32x32x16_bf16 (one tap x 16 channels per MFMA, half the
                                   issue-port time per flop) where mvsgi_conv3d_v32_applies(); weights from
                                   mvsgi_conv3d_pack_weights_bf16x3_v32                                                  */
#define MVSGI_CONV_BF16X3_D32 6 /* the same arithmetic on 32-channel slices: the K = 32 of an MFMA is ONE tap of 32 input channels
                                   (27 k-steps per 32 channels; the tap-pair layout of MVSGI_CONV_BF16X3 needs 28, its 14th pair
                                   half empty) and a unit has half as many slices to synchronise on.  Cin % 32 == 0, stride 1, launches
                                   large enough for a 128- / 160-voxel brick: where mvsgi_conv3d_d32_applies(); weights from
                                   mvsgi_conv3d_pack_weights_split(layout = MVSGI_CONV_BF16X3_D32 [| MVSGI_CONV_F16]), sized by
                                   mvsgi_conv3d_packed_weight_bytes_bf16x3.  Same products, another summation order.          */

#define MVSGI_CONV_F16 0x100   /* FLAG, OR-ed into MVSGI_CONV_BF16X3 / _C16 / _V32 (impl of mvsgi_conv3d_f32, w_layout of
                                  mvsgi_conv3d_up2_f32, layout of mvsgi_conv3d_pack_weights_split): the same kernels and packed
                                  layouts in the fp16 split ("f16x3") -- hi = fp16(x), lo = fp16(x - hi), v_mfma_*_f16: 11 + 11
                                  significant bits per operand instead of 8 + 8 at the same matrix rate (~10x closer to the
                                  fp32 reference); operands are clamped to fp16's range (+-65504), and callers pre-scale each
                                  output channel's weights by a power of two (undone in `scale`) so that their lo parts are
                                  normal fp16 numbers                                                                    */

#define MVSGI_SPLIT_F16 1      /* `fmt` of the *_fmt entry points: split-padded activations / packed weights hold fp16 pairs (the
                                  "f16x3" arithmetic, see MVSGI_CONV_F16) instead of bf16 pairs (fmt 0)                     */

int         mvsgi_abi_version(void);
const char* mvsgi_last_error(void);

/* ---- range report of the fp16 split ------------------------------------------------------
 * The reference computes in fp32 with no clamp (dsta_mvs/model/common/common_modules.py:105-115,
 * distance_regressor/distance_regressor.py:51-79); the "f16x3" arithmetic (MVSGI_CONV_F16 / MVSGI_SPLIT_F16) saturates
 * what it writes or stages in fp16 pieces at +-65504, and the Winograd level's fp32-padded records at +-16376.  Every
 * kernel that clamps reports a clamp that ENGAGED (a value at or beyond the bound) into sticky per-process words of pinned
 * host memory; this call reads them (no stream operation, no copy: ask after synchronising the streams whose launches
 * the question is about) and optionally clears them.  Results computed while a bit was raised are NOT the reference's:
 * re-run in the bf16 split (fmt 0 / no MVSGI_CONV_F16: fp32's range) or in the exact fp32 mode.
 *   flags  out, may be NULL: OR of the MVSGI_SAT_* bits raised since the last clear
 */
#define MVSGI_SAT_SWEEP 1u     /* the sweep's split-padded cost volume (the one un-normalised tensor of the path) reached +-65504 */
#define MVSGI_SAT_SPLIT 2u     /* an activation written or staged in fp16 pieces by a conv layer reached +-65504                */
#define MVSGI_SAT_WINO  4u     /* an activation of the Winograd level reached +-16376, or a transformed sum of four +-65504      */
int mvsgi_saturation_flags(int clear, unsigned* flags);
int mvsgi_saturation_words(unsigned* words8);      /* diagnostics: the 8 raw words (a kernel only ever stores 1 into one of them) */

/* ---- K1: fused spherical sweep -------------------------------------------------------
 * Replaces SphericalSweepStdMasked.sweep (cost_volume_builder/spherical_sweep_avg.py:38-136)
 * including both bilinear_grid_sample calls per candidate (backports/backports.py:34-86).
 *   feats      [B][N][C][Hi][Wi]      fp32 (the feature extractor's NCHW output)
 *   grids      [B][N][D][Ho][Wo][2]   fp32 normalised (x, y), align_corners=False
 *   grid_masks [B][N][D][Ho][Wo][1]   uint8/bool (grid_mask_is_f32 == 0) or fp32 (== 1)
 *   masks      [B][N][1][Hm][Wm]      fp32
 *   vol        [B][D][Ho][Wo][C]      fp32 out: population variance over the valid cameras,
 *                                     0 where fewer than two cameras are valid
 */
int mvsgi_sweep_std_f32(const float* feats, const float* grids, const void* grid_masks,
                        int grid_mask_is_f32, const float* masks, float* vol,
                        int B, int N, int C, int Hi, int Wi, int Hm, int Wm,
                        int D, int Ho, int Wo, mvsgi_stream_t stream);

/* Replaces SphericalSweep.sweep (cost_volume_builder/spherical_sweep.py:38-68):
 *   vol [B][D][Ho][Wo][N*C], channel = cam*C + c (spherical_sweep.py:60-61). */
int mvsgi_sweep_cat_f32(const float* feats, const float* grids, float* vol,
                        int B, int N, int C, int Hi, int Wi,
                        int D, int Ho, int Wo, mvsgi_stream_t stream);

/* The same two sweeps on channels-last feature maps, feats [B][N][Hi][Wi][C] (C % 4 == 0; std: N <= 4):
 * one 64-byte texel per tap instead of C strided planes.  The Python layer transposes the
 * feature extractor's NCHW output once (mvsgi_ncv_to_nvc_f32) unless it already is channels-last. */
int mvsgi_sweep_std_nhwc_f32(const float* feats, const float* grids, const void* grid_masks,
                             int grid_mask_is_f32, const float* masks, float* vol,
                             int B, int N, int C, int Hi, int Wi, int Hm, int Wm,
                             int D, int Ho, int Wo, mvsgi_stream_t stream);
int mvsgi_sweep_cat_nhwc_f32(const float* feats, const float* grids, float* vol,
                             int B, int N, int C, int Hi, int Wi,
                             int D, int Ho, int Wo, mvsgi_stream_t stream);

/* Rig-constant validity.  The mask half of SphericalSweepStdMasked.sweep -- (bilinear_grid_sample(
 * masks) > 0) & grid_masks, spherical_sweep_avg.py:92-102 -- depends only on grids / grid_masks /
 * masks, which the reference builds once per camera rig (api/inference_class.py:40-45).
 * mvsgi_sweep_validity_u8 evaluates it once into vmask [B][D][Ho][Wo] (bit cam set = camera cam is
 * valid; N <= 8); mvsgi_sweep_std_nhwc_valid_f32 is mvsgi_sweep_std_nhwc_f32 reading that byte
 * instead of re-sampling the masks every frame.  Bit-identical output. */
int mvsgi_sweep_validity_u8(const float* grids, const void* grid_masks, int grid_mask_is_f32,
                            const float* masks, unsigned char* vmask,
                            int B, int N, int Hm, int Wm, int D, int Ho, int Wo, mvsgi_stream_t stream);
int mvsgi_sweep_std_nhwc_valid_f32(const float* feats, const float* grids, const unsigned char* vmask,
                                   float* vol, int B, int N, int C, int Hi, int Wi,
                                   int D, int Ho, int Wo, mvsgi_stream_t stream);
/* The same with ONE rig for the whole batch: grids [1][N][D][Ho][Wo][2], vmask [1][D][Ho][Wo] (the rig constants are
 * frame-independent, api/inference_class.py:40-45); feats / vol keep their batch. */
int mvsgi_sweep_std_nhwc_valid_rig_f32(const float* feats, const float* grids, const unsigned char* vmask,
                                       float* vol, int B, int N, int C, int Hi, int Wi, int D, int Ho, int Wo,
                                       mvsgi_stream_t stream);

/* ---- K2: 3x3x3 convolution block -----------------------------------------------------
 * Replaces BaseConvBlk3d.forward (common/common_modules.py:107-115):
 *   y = act( conv3d(x, w, pad=1, stride) * scale[co] + shift[co] (+ res) )
 * eval-mode BatchNorm3d is the per-channel (scale, shift); a conv bias is shift with
 * scale = 1; act(v) = v > 0 ? v : v * neg_slope  (LeakyReLU: 0.01, ReLU: 0, identity: 1).
 *   x   [B][Din][Hin][Win][Cin]   y / res [B][Do][Ho][Wo][Cout],  Do = (Din-1)/stride+1 ...
 *   w_packed: output of mvsgi_conv3d_pack_weights_f32 (needed by the tiled paths -- MFMA, and
 *             the Cout == 1 cost head; may be NULL when impl == MVSGI_CONV_DIRECT);
 *   w_oidhw:  the PyTorch [Cout][Cin][3][3][3] tensor (needed by the direct path; may be NULL
 *             when impl == MVSGI_CONV_MFMA).
 * mvsgi_conv3d_variant_f32 names the kernel the dispatcher will launch for a problem (the
 * demangled kernel name as rocprofv3 prints it; NULL + last_error when unservable), so that
 * a benchmark can attribute time to the kernel that actually ran.
 */
size_t mvsgi_conv3d_packed_weight_floats(int Cout, int Cin);
int mvsgi_conv3d_pack_weights_f32(const float* w_oidhw, float* w_packed, int Cout, int Cin,
                                  mvsgi_stream_t stream);
size_t mvsgi_conv3d_packed_weight_bytes_bf16x3(int Cout, int Cin);
int mvsgi_conv3d_pack_weights_bf16x3(const float* w_oidhw, void* w_packed, int Cout, int Cin,
                                     mvsgi_stream_t stream);
/* weights for a split kernel in either split: layout = MVSGI_CONV_BF16X3 | _C16 | _V32, optionally | MVSGI_CONV_F16; w_packed sized by
 * mvsgi_conv3d_packed_weight_bytes_bf16x3 / _c16 / _v32 (the two splits share the layouts) */
int mvsgi_conv3d_pack_weights_split(const float* w_oidhw, void* w_packed, int Cout, int Cin, int layout, mvsgi_stream_t stream);
int mvsgi_conv3d_f32(const float* x, const float* w_oidhw, const float* w_packed,
                     const float* scale, const float* shift, const float* res, float* y,
                     int B, int Cin, int Din, int Hin, int Win, int Cout,
                     int stride, float neg_slope, int impl, mvsgi_stream_t stream);
const char* mvsgi_conv3d_variant_f32(int B, int Cin, int Din, int Hin, int Win, int Cout,
                                     int stride, int impl);

/* ResizeConv3d.forward (common/common_modules.py:332-355) in ONE launch: the trilinear x2 upsample
 * (F.interpolate, align_corners=False) of x [B][Dl][Hl][Wl][Cin] is evaluated inside the convolution's
 * staging path, so neither the resize kernel nor the upsampled tensor exist:
 *   y [B][2Dl][2Hl][2Wl][Cout] = act( conv3d(upsample2(x), w, pad=1) * scale + shift (+ res) )
 * Split-bf16 MFMA arithmetic; w_packed from mvsgi_conv3d_pack_weights_bf16x3; Cin, Cout % 16 == 0.
 * (Odd target sizes -- the second re-interpolation of :343-350 -- use mvsgi_resize_trilinear_f32 + mvsgi_conv3d_f32.) */
int mvsgi_conv3d_up2_f32(const float* x, const void* w_packed, int w_layout /* MVSGI_CONV_BF16X3 | _C16 | _V32 */,
                         const float* scale, const float* shift, const float* res, float* y,
                         int B, int Cin, int Dl, int Hl, int Wl, int Cout, float neg_slope, mvsgi_stream_t stream);
const char* mvsgi_conv3d_up2_variant_f32(int B, int Cin, int Dl, int Hl, int Wl, int Cout, int w_layout);
/* 32x32x16-MFMA schedule (MVSGI_CONV_BF16X3_V32): Cout % 32 == 0, stride 1, enough bricks to fill the chip */
int mvsgi_conv3d_v32_applies(int B, int Cin, int Din, int Hin, int Win, int Cout, int stride);
/* 1 when mvsgi_conv3d_f32 accepts impl = MVSGI_CONV_BF16X3_D32 [| MVSGI_CONV_F16] for this problem (replaces nothing in the reference:
 * a second schedule of BaseConvBlk3d's convolution, dsta_mvs/model/common/common_modules.py:107-115) */
int mvsgi_conv3d_d32_applies(int B, int Cin, int Din, int Hin, int Win, int Cout, int stride);
/* ... and mvsgi_conv3d_up2_f32 with w_layout = MVSGI_CONV_BF16X3_D32 (ResizeConv3d, common_modules.py:332-355; low-resolution sizes) */
int mvsgi_conv3d_up2_d32_applies(int B, int Cin, int Dl, int Hl, int Wl, int Cout);
size_t mvsgi_conv3d_packed_weight_bytes_bf16x3_v32(int Cout, int Cin);
int mvsgi_conv3d_pack_weights_bf16x3_v32(const float* w_oidhw, void* w_packed, int Cout, int Cin, mvsgi_stream_t stream);
/* weights of a Cout == 16 layer in the plane-schedule layout (MVSGI_CONV_BF16X3_C16) */
size_t mvsgi_conv3d_packed_weight_bytes_bf16x3_c16(int Cin);
int mvsgi_conv3d_pack_weights_bf16x3_c16(const float* w_oidhw, void* w_packed, int Cin, mvsgi_stream_t stream);

/* ---- K2-2D: convolution block of the feature extractor ----------
 * Replaces BaseConvBlk2d.forward (common/common_modules.py:56-70) on channels-last images:
 *   y = act( conv2d(x, w, pad k/2, stride) * scale[co] + shift[co] (+ res) )
 *   x [B][Hin][Win][Cin]; in_nchw == 1: the caller's fp32 NCHW images (direct path only);
 *   in_nchw == 2: uint8 [B][Hin][Win][3] camera images, converted as `.float() / 255.0`
 *   (api/inference_class.py:104-107) inside the 5x5 stride-2 3->16 stem kernel,
 *   y / res [B][Ho][Wo][Cout], w_oihw [Cout][Cin][k][k], k odd <= 7, stride 1|2.
 * impl: MVSGI_CONV_BF16X3 = split-bf16 MFMA kernel (k == 3, Cin, Cout multiples of 16, w_packed from
 * mvsgi_conv2d_pack_weights_bf16x3); MVSGI_CONV_MFMA = exact fp32 MFMA kernel (k == 3, Cin % 16 == 0,
 * Cout in {16, 32}, w_packed from mvsgi_conv2d_pack_weights_f32); anything else, or AUTO/DIRECT = exact
 * fp32 direct kernels (incl. the LDS-tiled 5x5 stride-2 RGB stem). */
size_t mvsgi_conv2d_packed_weight_floats(int Cout, int Cin);            /* exact-fp32 MFMA path (MVSGI_CONV_MFMA) */
int mvsgi_conv2d_pack_weights_f32(const float* w_oihw, float* w_packed, int Cout, int Cin,
                                  mvsgi_stream_t stream);
/* uint8 HWC camera images into the 5x5 stride-2 3 -> 16 stem (simple_feature_extractor.py:21-31 after the /255 of
 * api/inference_class.py:104-107): weights / 255 cut into three bf16 pieces for the matrix cores.  Pass the packed buffer
 * as w_packed of mvsgi_conv2d_f32 together with in_nchw == 2 (needs Win % 4 == 0; otherwise the LDS-tiled kernel runs). */
size_t mvsgi_conv2d_stem_packed_weight_bytes(void);
int mvsgi_conv2d_stem_pack_weights(const float* w_oihw, void* w_packed, mvsgi_stream_t stream);
size_t mvsgi_conv2d_packed_weight_bytes_bf16x3(int Cout, int Cin);
int mvsgi_conv2d_pack_weights_bf16x3(const float* w_oihw, void* w_packed, int Cout, int Cin,
                                     mvsgi_stream_t stream);
int mvsgi_conv2d_f32(const float* x, const float* w_oihw, const void* w_packed,
                     const float* scale, const float* shift, const float* res, float* y,
                     int B, int Cin, int Hin, int Win, int Cout, int ksize, int stride,
                     float neg_slope, int impl, int in_nchw, mvsgi_stream_t stream);
const char* mvsgi_conv2d_variant_f32(int Cin, int Cout, int ksize, int stride, int impl, int in_nchw);

/* ---- K3: trilinear resize ------------------------------------------------------------
 * Replaces F.interpolate(mode='trilinear', align_corners=False, size=...) inside
 * ResizeConv3d.forward (common/common_modules.py:333-350), for x2 and for the odd-size
 * re-interpolation to the skip tensor's shape.  x [B][Di][Hi][Wi][C] -> y [B][Do][Ho][Wo][C]. */
int mvsgi_resize_trilinear_f32(const float* x, float* y, int B, int C,
                               int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                               mvsgi_stream_t stream);

/* ---- K4: fused upsample + soft-argmin ------------------------------------------------
 * Replaces DistanceRegressorWithFixedCandidates.forward (distance_regressor/
 * distance_regressor.py:51-79): bilinear x`scale` (scale in {1,2}; 1 = no interpolation),
 * softmax over D, expectation of inv_idx[d] = bf / dist[d].
 *   costs [B][D][H][W] (the C==1 volume), inv_idx [D],
 *   inv_dist [B][1][sH][sW], norm_costs [B][D][sH][sW] or NULL (inference discards it). */
int mvsgi_softargmin_f32(const float* costs, const float* inv_idx, float* inv_dist,
                         float* norm_costs, int B, int D, int H, int W, int scale,
                         mvsgi_stream_t stream);

/* Same with the post-processing of api/inference_class.py:111-114 folded in: inv_dist is divided by
 * post_div (= bf) after the expectation (post_div == 1 reproduces mvsgi_softargmin_f32). */
int mvsgi_softargmin_div_f32(const float* costs, const float* inv_idx, float* inv_dist,
                             float* norm_costs, int B, int D, int H, int W, int scale,
                             float post_div, mvsgi_stream_t stream);

/* ---- layout helpers for the module boundary ------------------------------------------
 * [B][C][D*H*W] <-> [B][D*H*W][C]; V = D*H*W.  Used when a caller hands the regulator a
 * contiguous NCDHW tensor (e.g. produced by the reference's own cv_builder). */
int mvsgi_ncv_to_nvc_f32(const float* x, float* y, int B, int C, long long V, mvsgi_stream_t stream);
int mvsgi_nvc_to_ncv_f32(const float* x, float* y, int B, int C, long long V, mvsgi_stream_t stream);

/* ResConvBlk2d.forward (common/common_modules.py:165-176) for 16 -> 16 channels, 3x3, stride 1 in ONE launch
 * (the extractor's residual blocks are HBM-bound; fused, the intermediate never leaves the CU):
 *   y = act( conv2( act( conv1(x) * scale1 + shift1 ) ) * scale2 + shift2 + x ),   x / y [N][H][W][16], x != y
 * w_packed1 / w_packed2 from mvsgi_conv2d_pack_weights_bf16x3(Cout = Cin = 16); split-bf16 MFMA arithmetic. */
int mvsgi_resblock2d_f32(const float* x, const void* w_packed1, const float* scale1, const float* shift1,
                         const void* w_packed2, const float* scale2, const float* shift2, float* y,
                         int N, int H, int W, float neg_slope, mvsgi_stream_t stream);

/* The same block on PRE-SPLIT activations (csrc/resblock2d_rs.hip): the extractor's chain of residual blocks hands its
 * activations on in the 2-D split-padded format
 *   [N][H + 4][W + 4][64 B],  pixel record = [hi(c 0-7) | hi(c 8-15) | lo(c 0-7) | lo(c 8-15)] bf16,  x = hi + lo,
 * with a two-pixel zero border that no kernel writes (the caller zeroes a buffer once; mvsgi_split2d_bytes gives its size).
 * Staging is then a pure copy (LDS-DMA), two workgroups share a CU, and the skip connection is read from LDS.
 * w_packed1 / w_packed2 from mvsgi_resblock2d_split_pack_weights ([16][16][3][3] fp32 each, the per-channel scale folded
 * in: y = act(conv2'(act(conv1'(x) + shift1)) + shift2 + x)).  y: the same format
 * (y_is_split != 0) or plain fp32 [N][H][W][16] (the hand-over to a kernel that stages fp32).  x_split != y.
 * mvsgi_conv2d_f32_out_split2d = mvsgi_conv2d_f32 writing that format (Cout == 16: the RGB stem and the split-bf16 3x3
 * kernels); mvsgi_f32_to_split2d / mvsgi_split2d_to_f32 convert at module boundaries and in tests. */
size_t mvsgi_split2d_bytes(int N, int H, int W);
int mvsgi_f32_to_split2d(const float* x, void* y_split, int N, int H, int W, mvsgi_stream_t stream);
int mvsgi_split2d_to_f32(const void* x_split, float* y, int N, int H, int W, mvsgi_stream_t stream);
size_t mvsgi_resblock2d_split_packed_weight_bytes(void);
int mvsgi_resblock2d_split_pack_weights(const float* w_oihw, const float* scale, void* w_packed, mvsgi_stream_t stream);
int mvsgi_resblock2d_split(const void* x_split, const void* w_packed1, const float* shift1,
                           const void* w_packed2, const float* shift2, void* y, int y_is_split,
                           int N, int H, int W, float neg_slope, mvsgi_stream_t stream);
/* BaseConvBlk2d 16 -> 16, 3x3, stride 2 between two runs of residual blocks, split-padded in and out (w_packed from
 * mvsgi_resblock2d_split_pack_weights): y_split [N][Ho+4][Wo+4][64 B], Ho = (H - 1) / 2 + 1 */
int mvsgi_conv2d_s2_split(const void* x_split, const void* w_packed, const float* shift, void* y_split,
                          int N, int H, int W, float neg_slope, mvsgi_stream_t stream);
int mvsgi_conv2d_f32_out_split2d(const float* x, const float* w_oihw, const void* w_packed,
                                 const float* scale, const float* shift, const float* res, void* y_split,
                                 int B, int Cin, int Hin, int Win, int Cout, int ksize, int stride,
                                 float neg_slope, int impl, int in_nchw, mvsgi_stream_t stream);

/* ---- sampling-grid generator ---------------------------------------
 * The closed forms of dsta_mvs/support/dataset/torch_cuda_sweep.py, composed as
 * MultiViewCameraModelDataset.make_sweep_grid_cuda does (support/dataset/multi_view_camera_model_dataset.py:474-521):
 *   rays   = RayMaker_UEPanorama.make_rays_for_candidates        (:76-132)   [3][N][H][W]
 *   points = transform_3D_points_torch(inverse camera pose, rays) (:385-408)  [B][3][M]
 *   grid, mask = DoubleSphereSampleGridMaker.make_grid(points)    (:262-298)  [B][M][2], [B][M] u8
 *   grid       = EquirectangularSampleGridMaker.make_grid(points) (:305-335)  [B][M][2]
 * fp32, operation order of the reference's torch expressions.  Run once per rig. */
int mvsgi_rays_panorama_f32(const float* dist, float* rays, int N, int H, int W,
                            float lat0, float lat1, float lon0, float lon1, mvsgi_stream_t stream);
int mvsgi_transform_points_f32(const float* T /* [B][4][4] */, const float* points, float* out,
                               int B, long long M, mvsgi_stream_t stream);
int mvsgi_grid_double_sphere_f32(const float* points, float* grid, unsigned char* mask, int B, long long M,
                                 float xi, float alpha, float fx, float fy, float cx, float cy,
                                 int calib_h, int calib_w, float w2, mvsgi_stream_t stream);
int mvsgi_grid_equirect_f32(const float* points, float* grid, int B, long long M, mvsgi_stream_t stream);

/* ---- deformable 2-D convolution with a given offset field ------------
 * SphereConvEquirect2d.forward + SphereConvBlk (common/common_modules.py:411-425, :509-547):
 *   y = act( deform_conv2d(x, offset, w, stride, padding, dilation) * scale + shift (+ res) )
 * with torchvision.ops.deform_conv2d's definition (offset channel 2*(i*Kw+j) = dy, +1 = dx of tap (i, j);
 * bilinear sampling, zero outside the image; no modulation mask).  Channels-last:
 *   x [N][H][W][Cin], offset [2*Kh*Kw][Ho][Wo] (shared) or [N][2*Kh*Kw][Ho][Wo] (offset_per_image),
 *   w_packed [Kh*Kw][Cin][Cout] from mvsgi_deform_conv2d_pack_weights_f32, y / res [N][Ho][Wo][Cout]. */
int mvsgi_deform_conv2d_pack_weights_f32(const float* w_oihw, float* w_packed, int Cout, int Cin, int Kh, int Kw,
                                         mvsgi_stream_t stream);
int mvsgi_deform_conv2d_f32(const float* x, const float* offset, int offset_per_image, const float* w_packed,
                            const float* scale, const float* shift, const float* res, float* y,
                            int N, int Cin, int H, int W, int Cout, int Kh, int Kw, int stride_h, int stride_w,
                            int pad_h, int pad_w, int dil_h, int dil_w, float neg_slope, mvsgi_stream_t stream);

/* mvsgi_sweep_std_nhwc_valid_f32 with the volume written split-padded (C == 16; see below), rig_batch in {1, B} */
int mvsgi_sweep_std_nhwc_valid_split(const float* feats, const float* grids, const unsigned char* vmask,
                                     void* vol_split, int B, int N, int C, int Hi, int Wi, int D, int Ho, int Wo,
                                     int rig_batch, mvsgi_stream_t stream);

/* ---- split-padded activations and the register-stationary conv (csrc/conv3d_rs.hip) ----------------------
 * Split-padded format of a channels-last activation tensor:  [B][D+2][H+2][W+2][C/16][4][8] bf16, where a voxel's
 * 16-channel slice is [hi(c 0-7) | hi(c 8-15) | lo(c 0-7) | lo(c 8-15)], x = hi + lo (hi = bf16(x), lo = bf16(x - hi));
 * the one-voxel border is zero (Conv3d's padding=1, common_modules.py:97-101) and is never written by any kernel:
 * allocate the buffer zero-filled once.  Same 4 bytes per element as fp32.
 *   mvsgi_act_split_bytes      size of such a buffer
 *   mvsgi_act_f32_to_split     fp32 [B][D][H][W][C] -> interior of a split-padded buffer
 *   mvsgi_act_split_to_f32     the inverse (hi + lo)
 *   mvsgi_conv3d_rs_split      BaseConvBlk3d.forward (common_modules.py:107-115) for Cin = Cout = 32, stride 1, on
 *                              split-padded x / res / y with the layer's weights resident in registers
 *                              (w_packed from mvsgi_conv3d_rs_pack_weights); same arithmetic as MVSGI_CONV_BF16X3;
 *                              neg_slope in [0, 1]; y_is_f32 != 0: y is a plain fp32 [B][D][H][W][32] tensor (the
 *                              hand-over to a kernel that reads fp32), otherwise split-padded
 */
int mvsgi_conv3d_f32_out_split(const float* x, const float* w_packed_b3, const float* scale, const float* shift,
                               const float* res, void* y_split, int B, int Cin, int Din, int Hin, int Win, int Cout,
                               int stride, float neg_slope, mvsgi_stream_t stream);   /* mvsgi_conv3d_f32 (MVSGI_CONV_BF16X3) writing a split-padded y */
/* post_vol (Cin = Cout = 16, stride 1, no residual) on a split-padded volume, fp32 [B][D][H][W][16] out; weights from
 * mvsgi_conv3d_rs_pack_weights(16, 16) */
int mvsgi_conv3d_rs16_split(const void* x_split, const void* w_packed_rs, const float* scale, const float* shift, float* y,
                            int B, int D, int H, int W, float neg_slope, mvsgi_stream_t stream);
/* the same layer writing a split-padded [B][D+2][H+2][W+2][16] tensor (zero-bordered by the caller, interior written): the
 * hand-over to mvsgi_conv3d_s2rs */
int mvsgi_conv3d_rs16_split_out_split(const void* x_split, const void* w_packed_rs, const float* scale, const float* shift,
                                      void* y_split, int B, int D, int H, int W, float neg_slope, mvsgi_stream_t stream);
/* UNetDownBlk.first of level 0 (unet_regulator.py:286-303: BaseConvBlk3d 16 -> 32, 3x3x3, stride 2, padding 1 + BN + LeakyReLU)
 * on split-padded activations (csrc/conv3d_s2rs.hip): x_split [B][D+2][H+2][W+2][64 B] -> y_split [B][Do+2][Ho+2][Wo+2][128 B],
 * Do = (D - 1) / 2 + 1 (likewise Ho, Wo), y = act(conv(x) * scale + shift); the scale is folded into w_packed by
 * mvsgi_conv3d_s2rs_pack_weights ([32][16][27] fp32 weights, scale[32]).  Staging by LDS-DMA: the layer is HBM-bound. */
size_t mvsgi_conv3d_s2rs_packed_weight_bytes(void);
int mvsgi_conv3d_s2rs_pack_weights(const float* w_oidhw, const float* scale, void* w_packed, mvsgi_stream_t stream);
int mvsgi_conv3d_s2rs(const void* x_split, const void* w_packed, const float* shift, void* y_split,
                      int B, int D, int H, int W, float neg_slope, mvsgi_stream_t stream);
size_t mvsgi_conv3d_rs_packed_weight_bytes(int Cout, int Cin);
int mvsgi_conv3d_rs_pack_weights(const float* w_oidhw, void* w_packed, int Cout, int Cin, mvsgi_stream_t stream);
size_t mvsgi_act_split_bytes(int B, int C, int D, int H, int W);
int mvsgi_act_f32_to_split(const float* x, void* y_split, int B, int C, int D, int H, int W, mvsgi_stream_t stream);
int mvsgi_act_split_to_f32(const void* x_split, float* y, int B, int C, int D, int H, int W, mvsgi_stream_t stream);
int mvsgi_conv3d_rs_split(const void* x_split, const void* w_packed_rs, const float* scale, const float* shift,
                          const void* res_split, void* y, int y_is_f32, int B, int Cin, int D, int H, int W, int Cout,
                          float neg_slope, mvsgi_stream_t stream);

/* ---- polyphase ResizeConv3d (csrc/conv3d_up2poly.hip) ------------------------------------------------------
 * ResizeConv3d.forward (dsta_mvs/model/common/common_modules.py:332-355) for Cin = 32, Cout = 16, no skip input --
 * out_costs.0 of the (16, 32) regulator (cost_volume_regulator/unet_regulator.py:52-60): trilinear x2 (align_corners=False,
 * :335-341) + Conv3d(k 3, padding 1) + eval BatchNorm + LeakyReLU evaluated as 8 phase convolutions over the LOW-resolution
 * tensor (weights folded at plan time; the faces of the volume, where ATen's clamp and the conv's zero padding change the
 * folded centre taps, by separate small launches).
 *   mvsgi_conv3d_up2_poly_plan_bytes   size of the plan for a low-resolution input of D x H x W voxels (0: bad dims)
 *   mvsgi_conv3d_up2_poly_plan         HOST function, no GPU call: w_oidhw_host [16][32][3][3][3] fp32 in host memory ->
 *                                      plan_host (position independent: copy the bytes to the device unchanged)
 *   mvsgi_conv3d_up2_poly_f32          x_split: split-padded [B][D+2][H+2][W+2][32]; y: fp32 [B][2D][2H][2W][16];
 *                                      scale / shift: 16 floats each; neg_slope in [0, 1]
 */
int mvsgi_conv3d_up2_f32_out_split(const float* x, const void* w_packed, int w_layout, const float* scale, const float* shift,
                                   const float* res, void* y_split, int B, int Cin, int Dl, int Hl, int Wl, int Cout,
                                   float neg_slope, mvsgi_stream_t stream);   /* mvsgi_conv3d_up2_f32 writing a split-padded y */
/* the polyphase layer with a split-padded result [B][2D+2][2H+2][2W+2][16], and the cost head (out_costs.1: Conv3d(Cin -> 1, k 3,
 * bias), unet_regulator.py:61-68) reading that format: hi | lo fragments straight into the matrix cores, split-bf16 arithmetic.
 * scale / shift of the head are passed by value (NoOp norm: 1 and the bias); neg_slope 1 = no activation. */
int mvsgi_conv3d_up2_poly_split(const void* x_split, const void* plan_dev, const float* scale, const float* shift, void* y_split,
                                int B, int D, int H, int W, float neg_slope, mvsgi_stream_t stream);
size_t mvsgi_conv3d_head_split_packed_weight_bytes(int Cin);
int mvsgi_conv3d_head_split_pack_weights(const float* w_oidhw, void* w_packed, int Cin, mvsgi_stream_t stream);
int mvsgi_conv3d_head_split(const void* x_split, const void* w_packed, float scale, float shift, float* y, int B, int Cin,
                            int D, int H, int W, float neg_slope, mvsgi_stream_t stream);
/* the head in the fp16 split: x_split written by a kernel of the fp16 split (mvsgi_conv3d_up2_f32_out_split with w_layout |
 * MVSGI_CONV_F16); weights pre-scaled by a power of two by the caller, its inverse folded into `scale` */
int mvsgi_conv3d_head_split_pack_weights_f16(const float* w_oidhw, void* w_packed, int Cin, mvsgi_stream_t stream);
int mvsgi_conv3d_head_split_f16(const void* x_split, const void* w_packed, float scale, float shift, float* y, int B, int Cin,
                            int D, int H, int W, float neg_slope, mvsgi_stream_t stream);
size_t mvsgi_conv3d_up2_poly_plan_bytes(int D, int H, int W);
int mvsgi_conv3d_up2_poly_plan(const float* w_oidhw_host, void* plan_host, int D, int H, int W);
int mvsgi_conv3d_up2_poly_f32(const void* x_split, const void* plan_dev, const float* scale, const float* shift, float* y,
                              int B, int D, int H, int W, float neg_slope, mvsgi_stream_t stream);

/* ---- the split-padded kernels in either 16-bit split (round 5) ---------------------------------------------------
 * The *_fmt forms of the entry points above: fmt = 0 runs the bf16 split (identical to the un-suffixed function), fmt =
 * MVSGI_SPLIT_F16 the fp16 split -- every split-padded tensor the call reads or writes, its packed weights and its residual then
 * hold fp16 pairs (x = hi + lo, 11 + 11 significant bits, values clamped to +-65504), and the products run on
 * v_mfma_f32_16x16x32_f16.  Weights of an fp16 call are pre-scaled by the caller with a power of two per output channel and its
 * inverse folded into `scale` (MVSGI_CONV_F16 above); mvsgi_conv3d_s2rs_fmt, whose epilogue has no per-channel multiplier, takes
 * one power of two for the layer (`unscale`, with `scale` at packing and `shift` multiplied by 1 / unscale).  Same reference ops
 * as the un-suffixed functions (BaseConvBlk3d / ResizeConv3d forward, common/common_modules.py:107-115, 332-355;
 * SphericalSweepStdMasked.sweep, cost_volume_builder/spherical_sweep_avg.py:38-136). */
int mvsgi_act_f32_to_split_fmt(const float* x, void* y_split, int B, int C, int D, int H, int W, int fmt, mvsgi_stream_t stream);
int mvsgi_act_split_to_f32_fmt(const void* x_split, float* y, int B, int C, int D, int H, int W, int fmt, mvsgi_stream_t stream);
int mvsgi_sweep_std_nhwc_valid_split_fmt(const float* feats, const float* grids, const unsigned char* vmask, void* vol_split,
                                         int B, int N, int C, int Hi, int Wi, int D, int Ho, int Wo, int rig_batch, int fmt,
                                         mvsgi_stream_t stream);
int mvsgi_conv3d_f32_out_split_fmt(const float* x, const float* w_packed_b3, const float* scale, const float* shift,
                                   const float* res, void* y_split, int B, int Cin, int Din, int Hin, int Win, int Cout,
                                   int stride, float neg_slope, int fmt, mvsgi_stream_t stream);
int mvsgi_conv3d_rs_pack_weights_fmt(const float* w_oidhw, void* w_packed, int Cout, int Cin, int fmt, mvsgi_stream_t stream);
int mvsgi_conv3d_rs_split_fmt(const void* x_split, const void* w_packed_rs, const float* scale, const float* shift,
                              const void* res_split, void* y, int y_is_f32, int B, int Cin, int D, int H, int W, int Cout,
                              float neg_slope, int fmt, mvsgi_stream_t stream);
int mvsgi_conv3d_rs16_split_fmt(const void* x_split, const void* w_packed_rs, const float* scale, const float* shift, void* y,
                                int y_is_split, int B, int D, int H, int W, float neg_slope, int fmt, mvsgi_stream_t stream);
int mvsgi_conv3d_s2rs_pack_weights_fmt(const float* w_oidhw, const float* scale, void* w_packed, int fmt, mvsgi_stream_t stream);
int mvsgi_conv3d_s2rs_fmt(const void* x_split, const void* w_packed, const float* shift, void* y_split, int B, int D, int H, int W,
                          float neg_slope, float unscale, int fmt, mvsgi_stream_t stream);
/* y_f32p != 0 (MVSGI_SPLIT_F16 only): y is written "fp32-padded" (the padded geometry, plain fp32 records): see mvsgi_conv3d_wino32_f16 */
int mvsgi_conv3d_s2rs_out_fmt(const void* x_split, const void* w_packed, const float* shift, void* y_split, int B, int D, int H, int W,
                              float neg_slope, float unscale, int fmt, int y_f32p, mvsgi_stream_t stream);
int mvsgi_conv3d_up2_poly_plan_fmt(const float* w_oidhw_host, void* plan_host, int D, int H, int W, int fmt);
/* y_is_split: 0 = fp32 [B][2D][2H][2W][16]; 1 = split-padded [B][2D+2][2H+2][2W+2][16], main kernel by the dispatcher: with
 * MVSGI_SPLIT_F16, D == 8, H even, W a multiple of 32 and enough frames to fill the chip (mvsgi_conv3d_up2_poly_wino_pays) the
 * Winograd form (csrc/conv3d_wino_up2.hip: polyphase over (H, W) x Winograd F(2x2, 3x3) x an explicit upsample along D, 2.25 x fewer
 * matrix instructions; the same ResizeConv3d.forward, common_modules.py:332-355), else the direct register-stationary kernel;
 * 3 = split-padded, the direct kernel whatever the dispatcher would choose; 5 = split-padded, the Winograd form (rejected where it
 * does not apply) */
int mvsgi_conv3d_up2_poly_fmt(const void* x_split, const void* plan_dev, const float* scale, const float* shift, void* y,
                              int y_is_split, int B, int D, int H, int W, float neg_slope, int fmt, mvsgi_stream_t stream);
int mvsgi_conv3d_up2_poly_wino_pays(int B, int D, int H, int W);      /* 1: y_is_split = 1 in the fp16 split runs the Winograd form */

/* ---- Winograd F(2x2, 3x3) x direct-D form of the 32 -> 32 convolutions (csrc/conv3d_wino.hip) -------------------------
 * BaseConvBlk3d.forward (dsta_mvs/model/common/common_modules.py:107-115) for Cin = Cout = 32, stride 1, on split-padded
 * activations in the fp16 split (MVSGI_SPLIT_F16), D = 8 or 16, H even, W a multiple of 32: 2.25 x fewer matrix instructions than the direct
 * form.  Weights: U = G g G^T per (cout, cin, kd), pre-scaled per cout by a power of two (its inverse folded into `scale`),
 * split, as [a 4][b 4][kd 3][cout tile 2][hi | lo][64 lanes][16 B]. */
size_t mvsgi_conv3d_wino32_packed_weight_bytes(void);
int mvsgi_conv3d_wino32_applies(int Cin, int Cout, int D, int H, int W, int stride, float neg_slope);
/* w_oidhw [32][32][3][3][3] -> w_packed (mvsgi_conv3d_wino32_packed_weight_bytes()), unscale [32] (2^-k per cout: multiply the layer's scale by it) */
int mvsgi_conv3d_wino32_pack_weights(const float* w_oidhw, void* w_packed, float* unscale, mvsgi_stream_t stream);
/* y: split-padded fp16 [B][D+2][H+2][W+2][32] (y_is_f32 == 0) or plain fp32 [B][D][H][W][32]; res_split: split-padded fp16 or NULL.
 * act_f32p != 0: x, res and a padded y are "fp32-padded" instead -- the same padded geometry with plain fp32 records (32 floats per
 * voxel, zero border): the hand-over between Winograd layers, which split their operands behind the transform anyway. */
int mvsgi_conv3d_wino32_f16(const void* x_split, const void* w_packed, const float* scale, const float* shift, const void* res_split,
                            void* y, int y_is_f32, int act_f32p, int B, int D, int H, int W, float neg_slope, mvsgi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVSGI_H */
