// K1: fused spherical sweep for gfx950.
//
// Evaluates, per output voxel (b, d, ho, wo), the reference's two bilinear_grid_sample calls per
// candidate (dsta_mvs/model/backports/backports.py:34-86) for all cameras and the masked
// mean / variance of cost_volume_builder/spherical_sweep_avg.py:92-125 (or the concat of
// spherical_sweep.py:38-68), without materialising any of the ~115 intermediate tensors per
// candidate, and writes the voxel's C channels as 16-byte stores (channels-last volume, what the
// conv kernels read).  Arithmetic follows the reference operation by operation with fp
// contraction disabled, so the raw volume is bit-identical to the PyTorch CPU result.
//
// Kernels, slowest to fastest (all the same bits):
//   sweep_std_kernel / sweep_cat_kernel      feats in the extractor's NCHW planes, one thread per voxel
//                                            (any C, up to 6 cameras);
//   sweep_std_nhwc_kernel / sweep_cat_nhwc_kernel   channels-last feats: 4 lanes share a voxel, one 64-byte
//                                            texel per tap through buffer descriptors (zero padding = the
//                                            hardware range check), XCD-contiguous block order;
//   sweep_validity_kernel + sweep_std_nhwc_v_kernel   the default: the rig-constant mask half evaluated once
//                                            per rig, the per-frame kernel walks the candidates of a row
//                                            with the next candidate's grid point prefetched.
#include "common.hpp"

namespace {

#include "split_fmt.hpp"

struct Bilin {
    int o00, o01, o10, o11;     // plane offsets y*W+x, or -1 when the tap is outside
    float w00, w01, w10, w11;   // weights of (x0,y0), (x0,y1), (x1,y0), (x1,y1)
};

#ifndef MVSGI_SWEEP_SKIP_CAM
#define MVSGI_SWEEP_SKIP_CAM 1      // 0: every tap of every camera is gathered (diagnostic builds: the A/B of the wave-uniform skips)
#endif
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

// bilin_setup with the four taps as BYTE offsets into a channels-last image of C4 = 4 C bytes per texel and rowB = W C4
// bytes per row; a tap outside the image gets an offset no buffer descriptor covers (the hardware range check reads 0).
__device__ __forceinline__ Bilin bilin_setup_bytes(float gx, float gy, int W, int H, int C4, int rowB) {
#pragma clang fp contract(off)
    Bilin t;
    const float x = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
    const float y = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    const float xf = floorf(x), yf = floorf(y);
    const float x1f = xf + 1.0f, y1f = yf + 1.0f;
    t.w00 = (x1f - x) * (y1f - y);
    t.w01 = (x1f - x) * (y - yf);
    t.w10 = (x - xf) * (y1f - y);
    t.w11 = (x - xf) * (y - yf);
    const int x0 = (int)fminf(fmaxf(xf, -2.0f), (float)W + 1.0f);
    const int y0 = (int)fminf(fmaxf(yf, -2.0f), (float)H + 1.0f);
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
    const int base = __mul24(y0, rowB) + __mul24(x0, C4);        // |y0| <= H + 1, rowB < 2^23 (checked by the launcher)
    constexpr int kOutside = (int)0x80000000;
    t.o00 = (vx0 & vy0) ? base : kOutside;
    t.o01 = (vx0 & vy1) ? base + rowB : kOutside;
    t.o10 = (vx1 & vy0) ? base + C4 : kOutside;
    t.o11 = (vx1 & vy1) ? base + rowB + C4 : kOutside;
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

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// Correctly rounded x / d for a small positive integer-valued d, given inv = RN(1 / d):
// q = RN(x * inv); r = x - d*q (exact in an fma); q' = RN(q + r * inv)  (Markstein).  Identical to
// IEEE division for normal-range results; the (never observed) tiny / huge / non-finite cases
// take the hardware division so the result stays bit-identical to the reference's torch.div.
__device__ __forceinline__ float div_small(float x, float d, float inv) {
    const float ax = fabsf(x);
    const float q = x * inv;
    const float r = __builtin_fmaf(-d, q, x);
    float res = __builtin_fmaf(r, inv, q);
    if (ax == 0.0f) res = x;                                         // +-0 / d = +-0
    const bool rare = ax != 0.0f && !(ax >= 1e-30f && ax <= 1e30f);
    if (__builtin_amdgcn_ballot_w64(rare) != 0) {                    // wave-uniform: normally skipped entirely
        if (rare) res = x / d;                                       // exact hardware division
    }
    return res;
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

// ------------------------------------------------------------------------------------------
// channels-last variants: feats [B][N][Hi][Wi][C].  Four lanes share one output voxel, each owning
// channel quads {q, q+4, ...}: a tap is one 16-byte load per lane and the four lanes of a voxel
// read one contiguous 64-byte texel (C == 16), 4x fewer gather instructions than the NCHW kernel
// and 4x more waves in flight.  The per-camera set-up (bilinear taps of the feature map, the mask
// sample, validity) is computed once -- lane q of a quad takes camera q -- and shared through
// quad shuffles.  Arithmetic is unchanged (bit-identical output).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4_t ld4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }

__device__ __forceinline__ f32x4_t bilin_fetch4(const float* __restrict__ img, int C, const Bilin& t) {
#pragma clang fp contract(off)
    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
    const f32x4_t i00 = t.o00 >= 0 ? ld4(img + t.o00 * C) : z;      // o < Hi*Wi, o*C fits 32 bits (checked on host)
    const f32x4_t i01 = t.o01 >= 0 ? ld4(img + t.o01 * C) : z;
    const f32x4_t i10 = t.o10 >= 0 ? ld4(img + t.o10 * C) : z;
    const f32x4_t i11 = t.o11 >= 0 ? ld4(img + t.o11 * C) : z;
    return ((i00 * t.w00 + i01 * t.w01) + i10 * t.w10) + i11 * t.w11;
}

// Buffer-descriptor variants of the two fetches: the tap offset is a 32-bit byte offset into a
// per-image descriptor and an out-of-image tap (offset -1 -> huge unsigned) is answered with
// zeros by the hardware range check, which IS the reference's zero padding: no 64-bit address
// arithmetic and no selects per tap.
__device__ __forceinline__ f32x4_t bilin_fetch4_buf(__amdgpu_buffer_rsrc_t img, int c, int C, const Bilin& t) {
#pragma clang fp contract(off)
    const f32x4_t i00 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(img, (t.o00 * C + c) * 4, 0, 0));
    const f32x4_t i01 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(img, (t.o01 * C + c) * 4, 0, 0));
    const f32x4_t i10 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(img, (t.o10 * C + c) * 4, 0, 0));
    const f32x4_t i11 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(img, (t.o11 * C + c) * 4, 0, 0));
    return ((i00 * t.w00 + i01 * t.w01) + i10 * t.w10) + i11 * t.w11;
}

__device__ __forceinline__ float bilin_fetch_buf(__amdgpu_buffer_rsrc_t plane, const Bilin& t) {
#pragma clang fp contract(off)
    const float i00 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, t.o00 * 4, 0, 0));
    const float i01 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, t.o01 * 4, 0, 0));
    const float i10 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, t.o10 * 4, 0, 0));
    const float i11 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, t.o11 * 4, 0, 0));
    return ((i00 * t.w00 + i01 * t.w01) + i10 * t.w10) + i11 * t.w11;
}

// Blocks are dealt round-robin over the 8 XCDs (private 4 MiB L2s).  The flat block id is re-mapped
// (bijectively, cdna_hip_programming.md T1) so that every XCD owns a CONTIGUOUS run of the logical
// order (b, ho, d, w-tile): the candidates d of one output row hit neighbouring feature rows, so
// the blocks resident on an XCD at any time share a few feature-map rows of one frame in its L2.
// With the plain (x, y, z) order every XCD walked every frame's whole feature map (8x the fabric
// reads; measured FETCH_SIZE, profiles/).
__device__ __forceinline__ int sweep_xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// grid = ceil(Wo / 64) * D * Ho * B blocks (flat); block = 64 voxels x 4 lanes.  No per-lane integer
// division: (b, ho, d, w-tile) come from the (scalar) block index, wo from the thread index.
template <int NCAM>
__global__ __launch_bounds__(256) void sweep_std_nhwc_kernel(const float* __restrict__ feats,
                                                             const float* __restrict__ grids,
                                                             const unsigned char* __restrict__ gm_u8,
                                                             const float* __restrict__ gm_f32,
                                                             const float* __restrict__ masks,
                                                             float* __restrict__ vol, SweepDims s) {
#pragma clang fp contract(off)
    static_assert(NCAM <= 4, "one camera per lane of a quad");
    const int q = threadIdx.x & 3;
    const int WT = (s.Wo + 63) >> 6;
    int L = sweep_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int wt = L % WT;
    L /= WT;
    const int d = L % s.D;
    L /= s.D;
    const int ho = L % s.Ho;
    const int b = L / s.Ho;
    int wo = wt * 64 + (threadIdx.x >> 2);
    const bool live = wo < s.Wo;
    if (!live) wo = s.Wo - 1;            // keep whole quads alive for the shuffles
    const int HWm = s.Hm * s.Wm, HWi = s.Hi * s.Wi;

    // lane q sets up camera q
    Bilin mine = {};
    float myvalid = 0.f;
    if (q < NCAM) {
        const long long grow = (((long long)(b * NCAM + q) * s.D + d) * s.Ho + ho) * s.Wo;   // wave-uniform
        const float2 gxy = *reinterpret_cast<const float2*>(grids + (grow + wo) * 2);
        mine = bilin_setup(gxy.x, gxy.y, s.Wi, s.Hi);
        const Bilin mt = bilin_setup(gxy.x, gxy.y, s.Wm, s.Hm);
        const float sm = bilin_fetch(masks + (long long)(b * NCAM + q) * HWm, mt);
        const bool gmv = gm_f32 ? (gm_f32[grow + wo] != 0.0f) : (gm_u8[grow + wo] != 0);
        myvalid = ((sm > 0.0f) && gmv) ? 1.0f : 0.0f;
    }
    const int lane = threadIdx.x & 63, qbase = lane & ~3;
    __amdgpu_buffer_rsrc_t img[NCAM];                  // one descriptor per camera image (wave-uniform)
#pragma unroll
    for (int cam = 0; cam < NCAM; ++cam)
        img[cam] = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(feats + (long long)(b * NCAM + cam) * HWi * s.C), 0, HWi * s.C * 4, 0x00020000);
    Bilin ft[NCAM];
    float vf[NCAM];
    float n = 0.0f;
#pragma unroll
    for (int cam = 0; cam < NCAM; ++cam) {
        const int src = qbase + cam;
        ft[cam].o00 = __shfl(mine.o00, src);
        ft[cam].o01 = __shfl(mine.o01, src);
        ft[cam].o10 = __shfl(mine.o10, src);
        ft[cam].o11 = __shfl(mine.o11, src);
        ft[cam].w00 = __shfl(mine.w00, src);
        ft[cam].w01 = __shfl(mine.w01, src);
        ft[cam].w10 = __shfl(mine.w10, src);
        ft[cam].w11 = __shfl(mine.w11, src);
        vf[cam] = __shfl(myvalid, src);
        n = n + vf[cam];
    }
    const bool ok = n > 1.0f;
    const float cnt = ok ? n : 1.0f;
    const float inv = 1.0f / cnt;
    float* out = vol + ((((long long)b * s.D + d) * s.Ho + ho) * s.Wo + wo) * s.C;
    for (int c = q * 4; c < s.C; c += 16) {
        f32x4_t sv[NCAM];
#pragma unroll
        for (int cam = 0; cam < NCAM; ++cam) sv[cam] = bilin_fetch4_buf(img[cam], c, s.C, ft[cam]);
        f32x4_t r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float sum = 0.0f;
#pragma unroll
            for (int cam = 0; cam < NCAM; ++cam) sum = sum + sv[cam][k] * vf[cam];
            const float avg = div_small(sum, cnt, inv);
            float var = 0.0f;
#pragma unroll
            for (int cam = 0; cam < NCAM; ++cam) {
                const float t = vf[cam] != 0.0f ? sv[cam][k] : avg;
                const float df = t - avg;
                var = var + df * df;
            }
            var = div_small(var, cnt, inv);
            r[k] = ok ? var : 0.0f;
        }
        if (live) *reinterpret_cast<f32x4_t*>(out + c) = r;
    }
}

// ------------------------------------------------------------------------------------------
// Rig-constant validity.  `(bilinear_grid_sample(mask) > 0) & grid_mask` (spherical_sweep_avg.py:
// 92-102) depends only on grids / grid_masks / masks, which are constants of the camera rig
// (api/inference_class.py:40-45 builds them once): it is evaluated ONCE per rig into one byte per
// voxel (bit cam = camera cam is valid) with exactly the arithmetic of the fused kernel above, and
// the per-frame kernel below reads that byte instead of re-sampling the masks (4 scattered
// image-resolution taps per camera and voxel, and a whole dependent memory round trip).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sweep_validity_kernel(const float* __restrict__ grids,
                                                             const unsigned char* __restrict__ gm_u8,
                                                             const float* __restrict__ gm_f32,
                                                             const float* __restrict__ masks,
                                                             unsigned char* __restrict__ vmask, SweepDims s) {
#pragma clang fp contract(off)
    const long long HW = (long long)s.Ho * s.Wo;
    const long long total = (long long)s.B * s.D * HW;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long pix = idx % HW;
    const int d = (int)((idx / HW) % s.D);
    const int b = (int)(idx / (HW * s.D));
    const long long HWm = (long long)s.Hm * s.Wm;
    unsigned bits = 0;
    for (int cam = 0; cam < s.N; ++cam) {
        const long long g = ((long long)(b * s.N + cam) * s.D + d) * HW + pix;
        const float2 gxy = *reinterpret_cast<const float2*>(grids + g * 2);
        const Bilin mt = bilin_setup(gxy.x, gxy.y, s.Wm, s.Hm);
        const float sm = bilin_fetch(masks + (long long)(b * s.N + cam) * HWm, mt);
        const bool gmv = gm_f32 ? (gm_f32[g] != 0.0f) : (gm_u8[g] != 0);
        bits |= ((sm > 0.0f) && gmv) ? (1u << cam) : 0u;
    }
    vmask[idx] = (unsigned char)bits;
}

template <int CAM>
__device__ __forceinline__ int quad_bcast_i(int v) {     // lane CAM of every quad -> all four lanes (DPP, no LDS)
    return __builtin_amdgcn_mov_dpp(v, CAM * 0x55, 0xf, 0xf, true);
}
template <int CAM>
__device__ __forceinline__ float quad_bcast_f(float v) {
    return __builtin_bit_cast(float, quad_bcast_i<CAM>(__builtin_bit_cast(int, v)));
}
template <int CAM>
__device__ __forceinline__ Bilin quad_bcast(const Bilin& m) {
    Bilin t;
    t.o00 = quad_bcast_i<CAM>(m.o00);
    t.o01 = quad_bcast_i<CAM>(m.o01);
    t.o10 = quad_bcast_i<CAM>(m.o10);
    t.o11 = quad_bcast_i<CAM>(m.o11);
    t.w00 = quad_bcast_f<CAM>(m.w00);
    t.w01 = quad_bcast_f<CAM>(m.w01);
    t.w10 = quad_bcast_f<CAM>(m.w10);
    t.w11 = quad_bcast_f<CAM>(m.w11);
    return t;
}

// Per-frame kernel with the validity byte.  A block owns 64 consecutive wo of one (b, ho) row and
// walks `dchunk` candidates, fetching the NEXT candidate's grid point and validity byte before the
// 12 texel gathers of the current one, so a voxel costs one exposed memory round trip instead of
// three.  Logical block order (b, ho, d-chunk, w-tile), XCD-contiguous (see sweep_xcd_remap).
#ifndef MVSGI_SWEEP_NT
#define MVSGI_SWEEP_NT 0      // nt stores of the split-padded volume: measured neutral (post_vol reads it straight back)
#endif
#ifndef MVSGI_SWEEP_WAVES
#define MVSGI_SWEEP_WAVES 5      // waves per SIMD the register allocation aims at (experiment knob; 92 registers -> 5)
#endif
// F16: a split-padded output (vol_split) holds fp16 pairs instead of bf16 pairs (csrc/split_fmt.hpp; the variance is >= 0 and clamped
// to fp16's range)
template <int NCAM, bool C16, bool F16 = false>
__global__ __launch_bounds__(256, MVSGI_SWEEP_WAVES) void sweep_std_nhwc_v_kernel(const float* __restrict__ feats,
                                                               const float* __restrict__ grids,
                                                               const unsigned char* __restrict__ vmask,
                                                               float* __restrict__ vol, SweepDims s, int dchunk,
                                                               int nd, int rig_shared, unsigned char* __restrict__ vol_split,
                                                               unsigned* __restrict__ sat) {
#pragma clang fp contract(off)
    static_assert(NCAM <= 4, "one camera per lane of a quad");
    float satm = 0.f;          // fp16 split output: running maximum |value written| (range report, csrc/split_fmt.hpp)
    const int q = threadIdx.x & 3;
    const int WT = (s.Wo + 63) >> 6;
    int L = sweep_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int wt = L % WT;
    L /= WT;
    const int dc = L % nd;
    L /= nd;
    const int ho = L % s.Ho;
    const int b = L / s.Ho;
    int wo = wt * 64 + (threadIdx.x >> 2);
    const bool live = wo < s.Wo;
    if (!live) wo = s.Wo - 1;            // keep whole quads alive for the DPP broadcasts
    const int d0 = dc * dchunk;
    const int d1 = d0 + dchunk < s.D ? d0 + dchunk : s.D;
    const int HWi = s.Hi * s.Wi;
    const long long HW = (long long)s.Ho * s.Wo;

    __amdgpu_buffer_rsrc_t img[NCAM];                  // one descriptor per camera image (wave-uniform)
#pragma unroll
    for (int cam = 0; cam < NCAM; ++cam)
        img[cam] = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(feats + (long long)(b * NCAM + cam) * HWi * s.C), 0, HWi * s.C * 4, 0x00020000);
    // lane q walks camera q's grid (lanes beyond the rig re-read the last camera; unused)
    const int mycam = q < NCAM ? q : NCAM - 1;
    // rig_shared: one grid / validity set for the whole batch (the rig constants of api/inference_class.py:40-45 are the
    // same for every frame): all frames read frame 0's, which then stay in L2 instead of streaming B copies from HBM
    const int br = rig_shared ? 0 : b;
    const float2* gp = reinterpret_cast<const float2*>(grids) + ((long long)(br * NCAM + mycam) * s.D + d0) * HW +
                       (long long)ho * s.Wo + wo;
    const unsigned char* vp = vmask + ((long long)br * s.D + d0) * HW + (long long)ho * s.Wo + wo;
    float* out = vol + ((((long long)b * s.D + d0) * s.Ho + ho) * s.Wo + wo) * s.C;
    const long long vstep = HW * s.C;
    // vol_split (C == 16): the volume goes out in the split-padded format of conv3d_rs.hip instead -- [B][D+2][Ho+2][Wo+2]
    // records of 64 B = [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15]; lane q's 4 channels are 8 B of hi and 8 B of lo, which
    // the lane pairs of a quad trade so that every lane stores one whole 16 B piece (lanes 0..3 -> pieces 0, 2, 1, 3) and a
    // wave one contiguous KiB
    unsigned char* outs = vol_split ? vol_split + ((((long long)b * (s.D + 2) + d0 + 1) * (s.Ho + 2) + ho + 1) * (s.Wo + 2) + wo + 1) * 64 +
                                          ((q & 1) * 2 + (q >> 1)) * 16
                                    : nullptr;
    const long long sstep = (long long)(s.Ho + 2) * (s.Wo + 2) * 64;
    // one candidate: grid point -> taps (lane q = camera q, broadcast through the quad) -> 4 x NCAM
    // texel gathers -> masked variance
    const int C4 = C16 ? 64 : s.C * 4, rowB = s.Wi * C4;
    auto candidate = [&](const float2 gxy, const unsigned vm, float* __restrict__ o, unsigned char* __restrict__ os) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        // taps as BYTE offsets, worked out once by the camera's lane and broadcast through the quad
        const Bilin mine = bilin_setup_bytes(gxy.x, gxy.y, s.Wi, s.Hi, C4, rowB);
        Bilin ft[NCAM];
        ft[0] = quad_bcast<0>(mine);
        if (NCAM > 1) ft[NCAM > 1 ? 1 : 0] = quad_bcast<1>(mine);
        if (NCAM > 2) ft[NCAM > 2 ? 2 : 0] = quad_bcast<2>(mine);
        if (NCAM > 3) ft[NCAM > 3 ? 3 : 0] = quad_bcast<3>(mine);
        bool val[NCAM];
        f32x2_t VF[NCAM];
        float n = 0.0f;
#pragma unroll
        for (int cam = 0; cam < NCAM; ++cam) {
            val[cam] = ((vm >> cam) & 1u) != 0;
            const float vf = val[cam] ? 1.0f : 0.0f;
            VF[cam] = f32x2_t{vf, vf};
            n = n + vf;
        }
        const bool ok = n > 1.0f;
        const float cnt = ok ? n : 1.0f;
        // A camera that is not valid at this voxel contributes sv * 0 to the sum and is replaced by the mean in the variance
        // (spherical_sweep_avg.py:114-119), and a voxel seen by fewer than two cameras is 0 whatever was sampled (:125): those
        // texels are never needed.  Their tap offsets are sent out of the descriptor's range -- the loads return zeros and move no
        // data (same bits out for finite features: +-0 instead of +-0; a non-finite feature under an invalid camera would have
        // poisoned the reference's sum with NaN, here it is not read).  On the benchmark rig 35 % of the (voxel, camera) pairs.
#ifndef MVSGI_SWEEP_GATHER_ALL
#pragma unroll
        for (int cam = 0; cam < NCAM; ++cam) {
            const bool need = val[cam] & ok;
            ft[cam].o00 = need ? ft[cam].o00 : (int)0x80000000;
            ft[cam].o01 = need ? ft[cam].o01 : (int)0x80000000;
            ft[cam].o10 = need ? ft[cam].o10 : (int)0x80000000;
            ft[cam].o11 = need ? ft[cam].o11 : (int)0x80000000;
        }
#endif
        // RN(1 / cnt) for the camera counts there are: exactly what the division 1.0f / cnt returns
        const float inv = cnt == 2.0f ? 0.5f : cnt == 3.0f ? 0x1.555556p-2f : cnt == 4.0f ? 0.25f : 1.0f;
        const f32x2_t INV = {inv, inv}, NCNT = {-cnt, -cnt};
#pragma unroll 1
        for (int cb = q * 16; cb < (C16 ? 64 : C4); cb += 64) {      // C16: one trip, the tap offsets die with the loads
            f32x4_t tx[NCAM][4];
#ifdef MVSGI_SWEEP_ABL_NOGATHER              // diagnostic builds only: every tap reads texel 0 (L1 hits)
#pragma unroll
            for (int cam = 0; cam < NCAM; ++cam) { ft[cam].o00 &= 64; ft[cam].o01 &= 64; ft[cam].o10 &= 64; ft[cam].o11 &= 64; }
#endif
#pragma unroll
            for (int cam = 0; cam < NCAM; ++cam) {
#if MVSGI_SWEEP_SKIP_CAM
                // a camera no voxel of this wave needs (validity is spatially coherent: whole waves fall outside a camera's image
                // or mask) is not gathered at all: the kernel is bound by the texture addresser's instruction rate
                // (profiles/r04_sweep_texture_path_counters.txt), and a range-checked-away gather still costs its instruction
                if (__builtin_amdgcn_ballot_w64(val[cam] & ok) == 0) {
                    tx[cam][0] = tx[cam][1] = tx[cam][2] = tx[cam][3] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    continue;
                }
#endif
                tx[cam][0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(img[cam], ft[cam].o00 + cb, 0, 0));
                tx[cam][1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(img[cam], ft[cam].o01 + cb, 0, 0));
                tx[cam][2] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(img[cam], ft[cam].o10 + cb, 0, 0));
                tx[cam][3] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(img[cam], ft[cam].o11 + cb, 0, 0));
            }
            f32x2_t sv[NCAM][2];
#pragma unroll
            for (int cam = 0; cam < NCAM; ++cam) {
                const f32x2_t W00 = {ft[cam].w00, ft[cam].w00}, W01 = {ft[cam].w01, ft[cam].w01};
                const f32x2_t W10 = {ft[cam].w10, ft[cam].w10}, W11 = {ft[cam].w11, ft[cam].w11};
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const f32x2_t i00 = {tx[cam][0][2 * p], tx[cam][0][2 * p + 1]}, i01 = {tx[cam][1][2 * p], tx[cam][1][2 * p + 1]};
                    const f32x2_t i10 = {tx[cam][2][2 * p], tx[cam][2][2 * p + 1]}, i11 = {tx[cam][3][2 * p], tx[cam][3][2 * p + 1]};
                    sv[cam][p] = ((i00 * W00 + i01 * W01) + i10 * W10) + i11 * W11;          // backports.py:86, left to right
                }
            }
            // spherical_sweep_avg.py:106-125 on channel pairs.  The two divisions by cnt in {1, 2, 3, 4} are Markstein's
            // q = RN(x inv), r = x - cnt q (exact in the fma), RN(q + r inv): the correctly rounded quotient for every finite x
            // (cnt a power of two: q is already exact or correctly rounded; cnt = 3: r / 3 is a multiple of ulp / 3, never near
            // a rounding boundary, down to the subnormals).  Sums beyond 1e30 (and infinities) take the hardware division.
            f32x2_t sum[2], var[2];
            f32x4_t r;
            float big = 0.0f;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                sum[p] = sv[0][p] * VF[0];
#pragma unroll
                for (int cam = 1; cam < NCAM; ++cam) sum[p] = sum[p] + sv[cam][p] * VF[cam];
                const f32x2_t qa = sum[p] * INV;
                const f32x2_t avg = __builtin_elementwise_fma(__builtin_elementwise_fma(NCNT, qa, sum[p]), INV, qa);
#pragma unroll
                for (int cam = 0; cam < NCAM; ++cam) {
                    const f32x2_t t = {val[cam] ? sv[cam][p].x : avg.x, val[cam] ? sv[cam][p].y : avg.y};   // :119
                    const f32x2_t df = t - avg;
                    var[p] = cam == 0 ? df * df : var[p] + df * df;                                           // :122
                }
                const f32x2_t qv = var[p] * INV;
                const f32x2_t v = __builtin_elementwise_fma(__builtin_elementwise_fma(NCNT, qv, var[p]), INV, qv);
                r[2 * p] = ok ? v.x : 0.0f;                                                                   // :125
                r[2 * p + 1] = ok ? v.y : 0.0f;
                big = __builtin_fmaxf(__builtin_fmaxf(big, __builtin_fmaxf(__builtin_fabsf(sum[p].x), __builtin_fabsf(sum[p].y))),
                                      __builtin_fmaxf(var[p].x, var[p].y));
            }
            if (__builtin_amdgcn_ballot_w64(big > 1e30f) != 0) {                 // wave-uniform, never taken on real features
                float dv = cnt;
                asm volatile("; exact-division path" : "+v"(dv));          // (opaque: keeps the divisions inside the branch)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sm = sum[k >> 1][k & 1];
                    const float avg = sm / dv;
                    float vr = 0.0f;
#pragma unroll
                    for (int cam = 0; cam < NCAM; ++cam) {
                        const float t = val[cam] ? sv[cam][k >> 1][k & 1] : avg;
                        const float df = t - avg;
                        vr = vr + df * df;
                    }
                    vr = vr / dv;
                    r[k] = ok ? vr : 0.0f;
                }
            }
            if (os) {
                // x = hi + lo, hi = bf16(x) (RNE), lo = bf16(x - hi): the same split as the conv kernels' staging
                unsigned hi[2], lo[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if constexpr (F16) {
                        const float a_ = sf_clamp<true>(r[2 * p]), b_ = sf_clamp<true>(r[2 * p + 1]);
                        satm = sf_sat_acc(satm, a_, b_);
                        const unsigned hb = sf_cvt_pk<true>(a_, b_);
                        hi[p] = hb;
                        lo[p] = sf_cvt_pk<true>(a_ - sf_widen_lo<true>(hb), b_ - sf_widen_hi<true>(hb));
                    } else {
                        typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
                        const f32x2_t v = {r[2 * p], r[2 * p + 1]};
                        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2_t));
                        const f32x2_t hf = {__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
                        hi[p] = hb;
                        lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, b2_t));
                    }
                }
                // even lanes keep their hi and take the odd neighbour's hi; odd lanes take the even neighbour's lo
                const bool odd = (q & 1) != 0;
                const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)(odd ? hi[0] : lo[0]), 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)(odd ? hi[1] : lo[1]), 0xB1, 0xf, 0xf, true);
                const uint4 piece = odd ? make_uint4(r0, r1, lo[0], lo[1]) : make_uint4(hi[0], hi[1], r0, r1);
#ifdef MVSGI_SWEEP_ABL_NOSTORE               // diagnostic builds only
                if (live && hi[0] == 0x12345678u) {
#else
                if (live) {
#endif
#if MVSGI_SWEEP_NT
                    {
                        typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(nt_u32x4{piece.x, piece.y, piece.z, piece.w}, reinterpret_cast<nt_u32x4*>(os));
                    }
#else
                    *reinterpret_cast<uint4*>(os) = piece;
#endif
                }
            } else if (live) *reinterpret_cast<f32x4_t*>(reinterpret_cast<char*>(o) + cb) = r;
        }
    };
    // two candidates per trip with ping-pong registers (A, B): the loads of the next candidate are
    // issued before the gathers of the current one and nothing waits for them until its turn
    float2 gA = *gp;
    unsigned vA = *vp;
    for (int d = d0; d < d1; d += 2) {
        const bool hasB = d + 1 < d1;
        const long long sB = hasB ? HW : 0;
        const float2 gB = gp[sB];
        const unsigned vB = vp[sB];
        candidate(gA, vA, out, outs);
        const long long sA = d + 2 < d1 ? 2 * HW : sB;
        gA = gp[sA];
        vA = vp[sA];
        if (hasB) candidate(gB, vB, out + vstep, outs ? outs + sstep : nullptr);
        gp += 2 * HW;
        vp += 2 * HW;
        out += 2 * vstep;
        if (outs) outs += 2 * sstep;
    }
    if constexpr (F16) sf_sat_report(sat, kSatSweep, satm, kF16Max);
}

// grid = ceil(Wo / 64) * N * D * Ho * B blocks (flat), logical order (b, ho, d, cam, w-tile)
__global__ __launch_bounds__(256) void sweep_cat_nhwc_kernel(const float* __restrict__ feats,
                                                             const float* __restrict__ grids,
                                                             float* __restrict__ vol, SweepDims s) {
#pragma clang fp contract(off)
    const int q = threadIdx.x & 3;
    const int WT = (s.Wo + 63) >> 6;
    int L = sweep_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int wt = L % WT;
    L /= WT;
    const int cam = L % s.N;
    L /= s.N;
    const int d = L % s.D;
    L /= s.D;
    const int ho = L % s.Ho;
    const int b = L / s.Ho;
    const int wo = wt * 64 + (threadIdx.x >> 2);
    if (wo >= s.Wo) return;
    const long long grow = (((long long)(b * s.N + cam) * s.D + d) * s.Ho + ho) * s.Wo;
    const float2 gxy = *reinterpret_cast<const float2*>(grids + (grow + wo) * 2);
    const Bilin ft = bilin_setup(gxy.x, gxy.y, s.Wi, s.Hi);
    const int HWi = s.Hi * s.Wi;
    const long long vox = (((long long)b * s.D + d) * s.Ho + ho) * s.Wo + wo;
    float* out = vol + vox * ((long long)s.N * s.C) + (long long)cam * s.C;
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(feats + (long long)(b * s.N + cam) * HWi * s.C), 0, HWi * s.C * 4, 0x00020000);
    // (skipping the taps that lie outside the image for a whole wave, as the masked-variance kernel skips cameras, measured 2-4 %
    // SLOWER here: 1628 vs 1590 us per 16 4cam-32 frames, 1358 vs 1303 us per 64 E8 frames -- few waves of these rigs are
    // entirely outside a camera and the four wave-uniform branches cost more than they save.  Round 5: chunks of 2 / 3 / 4 consecutive
    // candidates per thread quad that INHERIT the previous candidate's texels -- the same cell 62 % of the steps on the 32-candidate
    // rig -- with the inherited requests sent outside the descriptor's range: 1679 / 1900 / 1703 us against this kernel's 1508 per 16
    // 4cam-32 frames, 1301 / 1487 / 1514 against 1201 per 64 E8 frames (profiles/r05_sweep_cat_variants.txt): a request that moves no
    // data still costs its instruction, and the instruction count is what bounds the texture path)
    for (int c = q * 4; c < s.C; c += 16)
        *reinterpret_cast<f32x4_t*>(out + c) = bilin_fetch4_buf(img, c, s.C, ft);
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

// Channels-last feature maps: feats [B][N][Hi][Wi][C], C % 4 == 0, N <= 4 (std).  Same outputs.
extern "C" int mvsgi_sweep_std_nhwc_f32(const float* feats, const float* grids, const void* grid_masks,
                                        int grid_mask_is_f32, const float* masks, float* vol, int B, int N,
                                        int C, int Hi, int Wi, int Hm, int Wm, int D, int Ho, int Wo,
                                        mvsgi_stream_t stream) {
    SweepDims s{B, N, C, Hi, Wi, Hm, Wm, D, Ho, Wo};
    if (check_dims(s, "mvsgi_sweep_std_nhwc_f32")) return 1;
    MVSGI_REQUIRE(Hm > 0 && Wm > 0, "mvsgi_sweep_std_nhwc_f32: non-positive mask size");
    MVSGI_REQUIRE(feats && grids && grid_masks && masks && vol, "mvsgi_sweep_std_nhwc_f32: null pointer");
    MVSGI_REQUIRE(N >= 1 && N <= 4, "mvsgi_sweep_std_nhwc_f32: num_cams %d not in [1, 4]", N);
    MVSGI_REQUIRE(C % 4 == 0, "mvsgi_sweep_std_nhwc_f32: C=%d must be a multiple of 4", C);
    const long long nblk = mvsgi::cdiv(Wo, 64) * Ho * D * B;
    MVSGI_REQUIRE((long long)Hi * Wi * C < (1ll << 31) && nblk < (1ll << 31),
                  "mvsgi_sweep_std_nhwc_f32: dimensions exceed the launch geometry");
    const dim3 grid((unsigned)nblk), block(256);
    const unsigned char* g8 = grid_mask_is_f32 ? nullptr : static_cast<const unsigned char*>(grid_masks);
    const float* g32 = grid_mask_is_f32 ? static_cast<const float*>(grid_masks) : nullptr;
    hipStream_t st = mvsgi::as_stream(stream);
    switch (N) {
        case 1: hipLaunchKernelGGL((sweep_std_nhwc_kernel<1>), grid, block, 0, st, feats, grids, g8, g32, masks, vol, s); break;
        case 2: hipLaunchKernelGGL((sweep_std_nhwc_kernel<2>), grid, block, 0, st, feats, grids, g8, g32, masks, vol, s); break;
        case 3: hipLaunchKernelGGL((sweep_std_nhwc_kernel<3>), grid, block, 0, st, feats, grids, g8, g32, masks, vol, s); break;
        case 4: hipLaunchKernelGGL((sweep_std_nhwc_kernel<4>), grid, block, 0, st, feats, grids, g8, g32, masks, vol, s); break;
    }
    return mvsgi::check_launch("mvsgi_sweep_std_nhwc_f32");
}

extern "C" int mvsgi_sweep_cat_nhwc_f32(const float* feats, const float* grids, float* vol, int B, int N, int C,
                                        int Hi, int Wi, int D, int Ho, int Wo, mvsgi_stream_t stream) {
    SweepDims s{B, N, C, Hi, Wi, 1, 1, D, Ho, Wo};
    if (check_dims(s, "mvsgi_sweep_cat_nhwc_f32")) return 1;
    MVSGI_REQUIRE(feats && grids && vol, "mvsgi_sweep_cat_nhwc_f32: null pointer");
    MVSGI_REQUIRE(C % 4 == 0, "mvsgi_sweep_cat_nhwc_f32: C=%d must be a multiple of 4", C);
    const long long nblk = mvsgi::cdiv(Wo, 64) * N * Ho * D * B;
    MVSGI_REQUIRE((long long)Hi * Wi * C < (1ll << 31) && nblk < (1ll << 31),
                  "mvsgi_sweep_cat_nhwc_f32: dimensions exceed the launch geometry");
    hipLaunchKernelGGL(sweep_cat_nhwc_kernel, dim3((unsigned)nblk), dim3(256), 0, mvsgi::as_stream(stream), feats,
                       grids, vol, s);
    return mvsgi::check_launch("mvsgi_sweep_cat_nhwc_f32");
}

// Rig-constant validity byte (bit cam) per voxel [B][D][Ho][Wo]; N <= 8.
extern "C" int mvsgi_sweep_validity_u8(const float* grids, const void* grid_masks, int grid_mask_is_f32,
                                       const float* masks, unsigned char* vmask, int B, int N, int Hm, int Wm, int D,
                                       int Ho, int Wo, mvsgi_stream_t stream) {
    SweepDims s{B, N, 1, 1, 1, Hm, Wm, D, Ho, Wo};
    if (check_dims(s, "mvsgi_sweep_validity_u8")) return 1;
    MVSGI_REQUIRE(Hm > 0 && Wm > 0, "mvsgi_sweep_validity_u8: non-positive mask size");
    MVSGI_REQUIRE(grids && grid_masks && masks && vmask, "mvsgi_sweep_validity_u8: null pointer");
    MVSGI_REQUIRE(N >= 1 && N <= 8, "mvsgi_sweep_validity_u8: num_cams %d not in [1, 8]", N);
    const long long total = (long long)B * D * Ho * Wo;
    MVSGI_REQUIRE(mvsgi::cdiv(total, 256) < (1ll << 31), "mvsgi_sweep_validity_u8: too many voxels");
    const unsigned char* g8 = grid_mask_is_f32 ? nullptr : static_cast<const unsigned char*>(grid_masks);
    const float* g32 = grid_mask_is_f32 ? static_cast<const float*>(grid_masks) : nullptr;
    hipLaunchKernelGGL(sweep_validity_kernel, dim3((unsigned)mvsgi::cdiv(total, 256)), dim3(256), 0,
                       mvsgi::as_stream(stream), grids, g8, g32, masks, vmask, s);
    return mvsgi::check_launch("mvsgi_sweep_validity_u8");
}

// SphericalSweepStdMasked.sweep with the validity byte of mvsgi_sweep_validity_u8 in place of
// grid_masks / masks: feats [B][N][Hi][Wi][C] -> vol [B][D][Ho][Wo][C]; identical output.
namespace {
int sweep_std_nhwc_valid_impl(const float* feats, const float* grids, const unsigned char* vmask, float* vol, int B, int N, int C,
                              int Hi, int Wi, int D, int Ho, int Wo, int rig_shared, mvsgi_stream_t stream,
                              unsigned char* vol_split = nullptr, int fmt = 0) {
    MVSGI_REQUIRE(fmt == 0 || (fmt == MVSGI_SPLIT_F16 && vol_split), "mvsgi_sweep_std_nhwc_valid_split: fmt %d not in {0, MVSGI_SPLIT_F16}", fmt);
    SweepDims s{B, N, C, Hi, Wi, 1, 1, D, Ho, Wo};
    if (check_dims(s, "mvsgi_sweep_std_nhwc_valid_f32")) return 1;
    MVSGI_REQUIRE(feats && grids && vmask && (vol || vol_split), "mvsgi_sweep_std_nhwc_valid_f32: null pointer");
    MVSGI_REQUIRE(!vol_split || C == 16, "mvsgi_sweep_std_nhwc_valid_split: the split-padded output needs C == 16 (got %d)", C);
    MVSGI_REQUIRE(N >= 1 && N <= 4, "mvsgi_sweep_std_nhwc_valid_f32: num_cams %d not in [1, 4]", N);
    MVSGI_REQUIRE(C % 4 == 0, "mvsgi_sweep_std_nhwc_valid_f32: C=%d must be a multiple of 4", C);
    // candidates per block: as many as keeps >= ~8k blocks in the launch (latency hiding across d
    // needs a few; filling 256 CUs x 4 resident blocks needs the rest)
    const long long rows = mvsgi::cdiv(Wo, 64) * Ho * B;
    // (never one chunk: with two the rows of blocks resident on an XCD span half as many feature-map rows, which then stay
    // in its 4 MiB L2 -- FETCH_SIZE 3.7 -> 0.79 GiB per 64-frame launch, profiles/r02_pmc_sweep_l2.txt)
    long long nd = rows >= 8192 ? 2 : mvsgi::cdiv(8192, rows);
    static const int nd_env = mvsgi::exp_env("MVSGI_SWEEP_ND") ? atoi(mvsgi::exp_env("MVSGI_SWEEP_ND")) : 0;      // experiments
    if (nd_env > 0) nd = nd_env;
    if (nd > D) nd = D;
    const int dchunk = (int)mvsgi::cdiv(D, nd);
    nd = mvsgi::cdiv(D, dchunk);
    const long long nblk = rows * nd;
    MVSGI_REQUIRE((long long)Hi * Wi * C < (1ll << 29) && (long long)Wi * C * 4 < (1ll << 23) && nblk < (1ll << 31),
                  "mvsgi_sweep_std_nhwc_valid_f32: dimensions exceed the launch geometry (image bytes < 2^31, row bytes < 2^23)");
    hipStream_t st = mvsgi::as_stream(stream);
    MVSGI_SAT_WORDS(sat);
    const dim3 grid((unsigned)nblk), block(256);
    if (fmt) {      // split-padded output in the fp16 split (C == 16 checked above)
        switch (N) {
            case 1: hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<1, true, true>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat); break;
            case 2: hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<2, true, true>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat); break;
            case 3: hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<3, true, true>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat); break;
            case 4: hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<4, true, true>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat); break;
        }
        return mvsgi::check_launch("mvsgi_sweep_std_nhwc_valid_split");
    }
    switch (N) {
        case 1: if (C == 16) hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<1, true>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat);
                else hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<1, false>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat);
                break;
        case 2: if (C == 16) hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<2, true>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat);
                else hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<2, false>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat);
                break;
        case 3: if (C == 16) hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<3, true>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat);
                else hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<3, false>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat);
                break;
        case 4: if (C == 16) hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<4, true>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat);
                else hipLaunchKernelGGL((sweep_std_nhwc_v_kernel<4, false>), grid, block, 0, st, feats, grids, vmask, vol, s, dchunk, (int)nd, rig_shared, vol_split, sat);
                break;
    }
    return mvsgi::check_launch("mvsgi_sweep_std_nhwc_valid_f32");
}
}  // namespace

extern "C" int mvsgi_sweep_std_nhwc_valid_f32(const float* feats, const float* grids, const unsigned char* vmask,
                                              float* vol, int B, int N, int C, int Hi, int Wi, int D, int Ho, int Wo,
                                              mvsgi_stream_t stream) {
    return sweep_std_nhwc_valid_impl(feats, grids, vmask, vol, B, N, C, Hi, Wi, D, Ho, Wo, 0, stream);
}

// The same with the volume written in the split-padded format of conv3d_rs.hip (C == 16; vol_split zero-bordered,
// [B][D+2][Ho+2][Wo+2][64 B]); rig_batch = 1: one rig for the whole batch, = B: per-frame grids / validity
extern "C" int mvsgi_sweep_std_nhwc_valid_split(const float* feats, const float* grids, const unsigned char* vmask,
                                                void* vol_split, int B, int N, int C, int Hi, int Wi, int D, int Ho, int Wo,
                                                int rig_batch, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(rig_batch == 1 || rig_batch == B, "mvsgi_sweep_std_nhwc_valid_split: rig_batch %d not in {1, B}", rig_batch);
    return sweep_std_nhwc_valid_impl(feats, grids, vmask, nullptr, B, N, C, Hi, Wi, D, Ho, Wo, rig_batch == 1 && B > 1 ? 1 : 0, stream,
                                     static_cast<unsigned char*>(vol_split));
}
extern "C" int mvsgi_sweep_std_nhwc_valid_split_fmt(const float* feats, const float* grids, const unsigned char* vmask,
                                                    void* vol_split, int B, int N, int C, int Hi, int Wi, int D, int Ho, int Wo,
                                                    int rig_batch, int fmt, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(rig_batch == 1 || rig_batch == B, "mvsgi_sweep_std_nhwc_valid_split: rig_batch %d not in {1, B}", rig_batch);
    return sweep_std_nhwc_valid_impl(feats, grids, vmask, nullptr, B, N, C, Hi, Wi, D, Ho, Wo, rig_batch == 1 && B > 1 ? 1 : 0, stream,
                                     static_cast<unsigned char*>(vol_split), fmt);
}

// The same with ONE rig for the whole batch: grids [1][N][D][Ho][Wo][2], vmask [1][D][Ho][Wo] (frame-independent rig
// constants, api/inference_class.py:40-45), feats / vol still [B]...
extern "C" int mvsgi_sweep_std_nhwc_valid_rig_f32(const float* feats, const float* grids, const unsigned char* vmask,
                                                  float* vol, int B, int N, int C, int Hi, int Wi, int D, int Ho, int Wo,
                                                  mvsgi_stream_t stream) {
    return sweep_std_nhwc_valid_impl(feats, grids, vmask, vol, B, N, C, Hi, Wi, D, Ho, Wo, 1, stream);
}
