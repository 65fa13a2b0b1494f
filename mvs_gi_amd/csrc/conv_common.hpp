// Definitions shared by the 3-D (conv3d.hip) and 2-D (conv2d.hip) convolution translation units.
#pragma once
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* x;
    const float* w_oidhw;
    const f32x4* wp;
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int B, Cin, Din, Hin, Win, Cout, Do, Ho, Wo, stride;
    float neg_slope;   // act(v) = v > 0 ? v : v * neg_slope; 1.0 = identity
    int tiles_d, tiles_h, tiles_w;
    int od_off, od_cnt; // first output plane and number of depth bricks of this launch (a launch over a range of depth bricks: the
                       // border-plane skip); od_cnt = 0: the whole depth
    int total_units;   // persistent bf16x3 kernel: bricks x cout blocks
    unsigned char* y_split;    // bf16x3 kernels: when set, the output goes here in the split-padded format (conv3d_rs.hip) instead of y
    unsigned long long* dbg;   // MVSGI_STAMPS diagnostic build only
    int f16;                   // split kernels: the fp16 split (hi = fp16(x), lo = fp16(x - hi), v_mfma_*_f16) instead of the bf16 split; host-side selector
    int ys_2d;                 // y_split is the 2-D format of resblock2d_rs.hip: no border along D (= images), two pixels along H and W
    unsigned* sat;             // split kernels, fp16 split: the range report's words (csrc/api.cpp), set by the launcher
};

constexpr int kVS = 20;           // LDS floats per staged voxel: 16 channels + 4 pad


}  // namespace
