// Kernel variants of conv3d.hip (shared with conv3d_f16.hip, which holds the fp16-split instantiations of the split kernels).
#pragma once
namespace {
enum Variant {
    V_DIRECT1, V_DIRECT4, V_HEAD,
    V_S1_N16_B256, V_S1_N32_B256, V_S1_N32_B64, V_S1_N64_B128, V_S1_N64_B64, V_S2_N32_B64, V_S2_N64_B64,
    // split-bf16 kernel: 16-wide bricks (conflict-free LDS reads); N = couts per workgroup
    B3_N16, B3_N32, B3_N48, B3_N64, B3_N64_H5, B3_N96, B3_N96_H5, B3_N128_P, B3_N128_PH5, B3_N192_PH5, B3_N32_S, B3_N64_S, B3_N16_T, B3_N32_T, B3_N16_TW, B3_N32_TB, B3_S2_N32, B3_S2_N32B, B3_S2_N64, B3_S2_N96, B3_S2_N128, B3_S2_N192,
    // 10 x 8 bricks: planes 8 mod 16 wide and a multiple of 10 high without padding (the siblings of the *_H5 / *_PH5 variants)
    B3_N64_W8, B3_N96_W8, B3_N128_PW8, B3_N192_PW8,
#ifdef MVSGI_EXPERIMENTAL
    // stride-2 bricks 2 x 2 x 16 (the W variants; round 6): a 16-voxel tile is 16 outputs of ONE row, whose stride-2 fragment reads
    // walk the 16 sixteen-byte units of a bank row with stride 10 (all even) while their pair partners, an odd number of units away,
    // take the odd ones -- against the 2 x 4 x 8 bricks' two-row tiles, which hit every bank row twice.  Measured
    // (profiles/r06_stride2_brick_shapes.txt): LDS conflict cycles -63 %, LDS-active cycles -30 %, and the layers' time -1.5 % ... +7.7 %
    // (8 % more halo to stage): the conflicts are not what bounds these layers.  Kept for the A/B (MVSGI_B3_FORCE=S2W_N96 ...), not dispatched.
    B3_S2W_N32B, B3_S2W_N64, B3_S2W_N96, B3_S2W_N128, B3_S2W_N192,
#endif
    // split-bf16 kernel with the trilinear x2 upsample fused into its producers (even bricks only)
    B3U_N16, B3U_N32, B3U_N32_M, B3U_N48, B3U_N64, B3U_N96, B3U_N32_TB,
    // Cout == 16 plane schedule (weights from mvsgi_conv3d_pack_weights_bf16x3_c16), plain and fused-upsample
    B3P_N16, B3PU_N16,
    // 32x32x16 schedule (Cout % 32 == 0, stride 1; weights from mvsgi_conv3d_pack_weights_bf16x3_v32), plain / fused upsample
    B3V_N32, B3V_N64, B3V_N64B, B3VU_N32, B3VU_N64,
    // 32-channel slices (MVSGI_CONV_BF16X3_D32; weights from mvsgi_conv3d_pack_weights_split(layout D32)): Cin % 32 == 0, stride 1,
    // the large-launch bricks only (conv3d_b3d_variants.inc)
    B3D_N64, B3D_N64_H5, B3D_N64_W8, B3D_N96, B3D_N96_H5, B3D_N96_W8, B3D_N128_P, B3D_N128_PH5, B3D_N128_PW8, B3D_N192_PH5, B3D_N192_PW8, B3D_N32_TB, B3D_N64_S,
    B3D2_N64, B3D2_N64_H5, B3D2_N64_W8, B3D2_N96, B3D2_N96_H5, B3D2_N96_W8,      // ... two-plane volumes: the depth skip
    B3D2_N32_TB, B3D2_N64_S,
    B3DU_N64, B3DU_N96, B3DU2_N64, B3DU2_N96,      // ... with the fused upsample (conv3d_b3du_variants.inc)
    V_COUNT
};
}  // namespace

namespace mvsgi {
// conv3d_f16.hip: the split kernel variants (B3*) in the fp16 split; `args` is a ConvArgs
int conv3d_launch_b3_f16(int variant, const void* args, hipStream_t st);
const char* conv3d_b3_f16_name(int variant);
}  // namespace mvsgi
