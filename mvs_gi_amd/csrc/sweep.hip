// K1: fused spherical sweep for gfx950.
//
// One thread per output voxel (b, d, ho, wo).  It evaluates the reference's two
// bilinear_grid_sample calls per candidate (dsta_mvs/model/backports/backports.py:34-86)
// for all cameras, and the masked mean/variance of
// cost_volume_builder/spherical_sweep_avg.py:92-125, without materialising any of the
// ~115 intermediate tensors per candidate, and writes the C channels of the voxel as
// contiguous 16-byte stores (channels-last volume, ready for the conv kernels).
//
// Neighbouring threads are neighbouring `wo`, whose sampling positions are neighbouring
// texels for real (smooth) grids, so the per-channel-plane gathers of a wave fall into a
// few cache lines; feats stay in the feature extractor's NCHW layout (no transpose pass).
//
// Arithmetic follows the reference operation by operation with fp contraction disabled,
// so the raw volume is bit-identical to the PyTorch CPU result.
#include "common.hpp"

namespace {

struct Bilin {
    int o00, o01, o10, o11;     // plane offsets y*W+x, or -1 when the tap is outside
    float w00, w01, w10, w11;   // weights of (x0,y0), (x0,y1), (x1,y0), (x1,y1)
};

__device__ __forceinline__ Bilin bilin_setup(float gx, float gy, int W, int H) {
#pragma clang fp contract(off)
    Bilin t;
    // backports.py:41-42 (align_corners=False)
    const float x = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
    const float y = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    const float xf = floorf(x), yf = floorf(y);
    const float x1f = xf + 1.0f, y1f = yf + 1.0f;
    // backports.py:52-55: weights from the unclamped coordinates
    t.w00 = (x1f - x) * (y1f - y);
    t.w01 = (x1f - x) * (y - yf);
    t.w10 = (x - xf) * (y1f - y);
    t.w11 = (x - xf) * (y - yf);
    // anything further out than one texel is outside anyway; clamping first keeps the
    // float->int conversion defined for huge or NaN coordinates
    const int x0 = (int)fminf(fmaxf(xf, -2.0f), (float)W + 1.0f);
    const int y0 = (int)fminf(fmaxf(yf, -2.0f), (float)H + 1.0f);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = (x0 >= 0) & (x0 < W), vx1 = (x1 >= 0) & (x1 < W);
    const bool vy0 = (y0 >= 0) & (y0 < H), vy1 = (y1 >= 0) & (y1 < H);
    t.o00 = (vx0 & vy0) ? y0 * W + x0 : -1;
    t.o01 = (vx0 & vy1) ? y1 * W + x0 : -1;
    t.o10 = (vx1 & vy0) ? y0 * W + x1 : -1;
    t.o11 = (vx1 & vy1) ? y1 * W + x1 : -1;
    return t;
}

__device__ __forceinline__ float bilin_fetch(const float* __restrict__ plane, const Bilin& t) {
#pragma clang fp contract(off)
    // zero padding: a tap outside the image reads 0 (backports.py:58-72)
    const float i00 = t.o00 >= 0 ? plane[t.o00] : 0.0f;
    const float i01 = t.o01 >= 0 ? plane[t.o01] : 0.0f;
    const float i10 = t.o10 >= 0 ? plane[t.o10] : 0.0f;
    const float i11 = t.o11 >= 0 ? plane[t.o11] : 0.0f;
    // backports.py:86: Ia*wa + Ib*wb + Ic*wc + Id*wd, left to right
    return ((i00 * t.w00 + i01 * t.w01) + i10 * t.w10) + i11 * t.w11;
}

struct SweepDims {
    int B, N, C, Hi, Wi, Hm, Wm, D, Ho, Wo;
};

template <int NCAM, int CC>
__global__ __launch_bounds__(256) void sweep_std_kernel(const float* __restrict__ feats,
                                                        const float* __restrict__ grids,
                                                        const unsigned char* __restrict__ gm_u8,
                                                        const float* __restrict__ gm_f32,
                                                        const float* __restrict__ masks,
                                                        float* __restrict__ vol, SweepDims s) {
#pragma clang fp contract(off)
    const long long total = (long long)s.B * s.D * s.Ho * s.Wo;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int wo = (int)(idx % s.Wo);
    const int ho = (int)((idx / s.Wo) % s.Ho);
    const int d = (int)((idx / ((long long)s.Wo * s.Ho)) % s.D);
    const int b = (int)(idx / ((long long)s.Wo * s.Ho * s.D));

    Bilin ft[NCAM];
    float vf[NCAM];
    float n = 0.0f;
    const long long HWi = (long long)s.Hi * s.Wi;
    const long long HWm = (long long)s.Hm * s.Wm;
#pragma unroll
    for (int cam = 0; cam < NCAM; ++cam) {
        const long long g = ((((long long)(b * NCAM + cam) * s.D + d) * s.Ho + ho) * s.Wo + wo);
        const float2 gxy = *reinterpret_cast<const float2*>(grids + g * 2);
        ft[cam] = bilin_setup(gxy.x, gxy.y, s.Wi, s.Hi);
        const Bilin mt = bilin_setup(gxy.x, gxy.y, s.Wm, s.Hm);
        const float sm = bilin_fetch(masks + (long long)(b * NCAM + cam) * HWm, mt);
        const bool gmv = gm_f32 ? (gm_f32[g] != 0.0f) : (gm_u8[g] != 0);
        const bool valid = (sm > 0.0f) && gmv;        // spherical_sweep_avg.py:92-102
        vf[cam] = valid ? 1.0f : 0.0f;
        n = n + vf[cam];                              // :106
    }
    const bool ok = n > 1.0f;                         // :108
    const float cnt = ok ? n : 1.0f;                  // :111

    float* out = vol + idx * s.C;
    for (int c = 0; c < s.C; c += CC) {
        float sv[NCAM][CC];
#pragma unroll
        for (int cam = 0; cam < NCAM; ++cam) {
            const float* plane = feats + ((long long)(b * NCAM + cam) * s.C + c) * HWi;
#pragma unroll
            for (int k = 0; k < CC; ++k) sv[cam][k] = bilin_fetch(plane + k * HWi, ft[cam]);
        }
        float r[CC];
#pragma unroll
        for (int k = 0; k < CC; ++k) {
            float sum = 0.0f;
#pragma unroll
            for (int cam = 0; cam < NCAM; ++cam) sum = sum + sv[cam][k] * vf[cam];
            const float avg = sum / cnt;              // :114
            float var = 0.0f;
#pragma unroll
            for (int cam = 0; cam < NCAM; ++cam) {
                const float t = vf[cam] != 0.0f ? sv[cam][k] : avg;   // :119
                const float df = t - avg;
                var = var + df * df;                  // :122
            }
            var = var / cnt;
            r[k] = ok ? var : 0.0f;                   // :125
        }
        if (CC == 4) {
            *reinterpret_cast<float4*>(out + c) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
#pragma unroll
            for (int k = 0; k < CC; ++k) out[c + k] = r[k];
        }
    }
}

template <int CC>
__global__ __launch_bounds__(256) void sweep_cat_kernel(const float* __restrict__ feats,
                                                        const float* __restrict__ grids,
                                                        float* __restrict__ vol, SweepDims s) {
#pragma clang fp contract(off)
    // thread = (b, d, ho, cam, wo): neighbouring lanes sample neighbouring texels of one camera
    const long long total = (long long)s.B * s.D * s.Ho * s.N * s.Wo;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int wo = (int)(idx % s.Wo);
    const int cam = (int)((idx / s.Wo) % s.N);
    const int ho = (int)((idx / ((long long)s.Wo * s.N)) % s.Ho);
    const int d = (int)((idx / ((long long)s.Wo * s.N * s.Ho)) % s.D);
    const int b = (int)(idx / ((long long)s.Wo * s.N * s.Ho * s.D));
    const long long g = ((((long long)(b * s.N + cam) * s.D + d) * s.Ho + ho) * s.Wo + wo);
    const float2 gxy = *reinterpret_cast<const float2*>(grids + g * 2);
    const Bilin ft = bilin_setup(gxy.x, gxy.y, s.Wi, s.Hi);
    const long long HWi = (long long)s.Hi * s.Wi;
    const long long vox = (((long long)b * s.D + d) * s.Ho + ho) * s.Wo + wo;
    float* out = vol + vox * ((long long)s.N * s.C) + (long long)cam * s.C;   // spherical_sweep.py:60-61
    const float* base = feats + ((long long)(b * s.N + cam) * s.C) * HWi;
    for (int c = 0; c < s.C; c += CC) {
        float r[CC];
#pragma unroll
        for (int k = 0; k < CC; ++k) r[k] = bilin_fetch(base + (long long)(c + k) * HWi, ft);
        if (CC == 4) {
            *reinterpret_cast<float4*>(out + c) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
#pragma unroll
            for (int k = 0; k < CC; ++k) out[c + k] = r[k];
        }
    }
}

int check_dims(const SweepDims& s, const char* who) {
    MVSGI_REQUIRE(s.B > 0 && s.N > 0 && s.C > 0 && s.Hi > 0 && s.Wi > 0 && s.D > 0 && s.Ho > 0 && s.Wo > 0,
                  "%s: non-positive dimension", who);
    MVSGI_REQUIRE((long long)s.Hi * s.Wi < (1ll << 31) && (long long)s.Hm * s.Wm < (1ll << 31),
                  "%s: image plane too large for 32-bit texel offsets", who);
    return 0;
}

}  // namespace

extern "C" int mvsgi_sweep_std_f32(const float* feats, const float* grids, const void* grid_masks,
                                   int grid_mask_is_f32, const float* masks, float* vol, int B, int N,
                                   int C, int Hi, int Wi, int Hm, int Wm, int D, int Ho, int Wo,
                                   mvsgi_stream_t stream) {
    SweepDims s{B, N, C, Hi, Wi, Hm, Wm, D, Ho, Wo};
    if (check_dims(s, "mvsgi_sweep_std_f32")) return 1;
    MVSGI_REQUIRE(Hm > 0 && Wm > 0, "mvsgi_sweep_std_f32: non-positive mask size");
    MVSGI_REQUIRE(feats && grids && grid_masks && masks && vol, "mvsgi_sweep_std_f32: null pointer");
    MVSGI_REQUIRE(N >= 1 && N <= 6, "mvsgi_sweep_std_f32: num_cams %d not in [1, 6]", N);
    const long long total = (long long)B * D * Ho * Wo;
    const dim3 grid((unsigned)mvsgi::cdiv(total, 256)), block(256);
    const unsigned char* g8 = grid_mask_is_f32 ? nullptr : static_cast<const unsigned char*>(grid_masks);
    const float* g32 = grid_mask_is_f32 ? static_cast<const float*>(grid_masks) : nullptr;
    hipStream_t st = mvsgi::as_stream(stream);
#define LAUNCH_STD(NC)                                                                                    \
    case NC:                                                                                              \
        if (C % 4 == 0)                                                                                   \
            hipLaunchKernelGGL((sweep_std_kernel<NC, 4>), grid, block, 0, st, feats, grids, g8, g32, masks, \
                               vol, s);                                                                   \
        else                                                                                              \
            hipLaunchKernelGGL((sweep_std_kernel<NC, 1>), grid, block, 0, st, feats, grids, g8, g32, masks, \
                               vol, s);                                                                   \
        break;
    switch (N) {
        LAUNCH_STD(1)
        LAUNCH_STD(2)
        LAUNCH_STD(3)
        LAUNCH_STD(4)
        LAUNCH_STD(5)
        LAUNCH_STD(6)
    }
#undef LAUNCH_STD
    return mvsgi::check_launch("mvsgi_sweep_std_f32");
}

extern "C" int mvsgi_sweep_cat_f32(const float* feats, const float* grids, float* vol, int B, int N, int C,
                                   int Hi, int Wi, int D, int Ho, int Wo, mvsgi_stream_t stream) {
    SweepDims s{B, N, C, Hi, Wi, 1, 1, D, Ho, Wo};
    if (check_dims(s, "mvsgi_sweep_cat_f32")) return 1;
    MVSGI_REQUIRE(feats && grids && vol, "mvsgi_sweep_cat_f32: null pointer");
    const long long total = (long long)B * D * Ho * N * Wo;
    const dim3 grid((unsigned)mvsgi::cdiv(total, 256)), block(256);
    hipStream_t st = mvsgi::as_stream(stream);
    if (C % 4 == 0)
        hipLaunchKernelGGL((sweep_cat_kernel<4>), grid, block, 0, st, feats, grids, vol, s);
    else
        hipLaunchKernelGGL((sweep_cat_kernel<1>), grid, block, 0, st, feats, grids, vol, s);
    return mvsgi::check_launch("mvsgi_sweep_cat_f32");
}
