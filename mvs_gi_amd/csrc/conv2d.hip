// 2-D convolution block for the feature extractor:
//   y = act( conv2d(x, w, pad k/2, stride) * scale + shift (+ res) )
// replaces BaseConvBlk2d.forward (dsta_mvs/model/common/common_modules.py:56-70) with eval-mode
// BatchNorm2d as the per-channel (scale, shift); images are channels-last [B][H][W][C]
// (the first layer may read the caller's NCHW images directly).
//
//  * 3x3, Cin/Cout multiples of 16: the split-bf16 MFMA kernel of conv3d_bf16x3.hpp instantiated
//    with KD = 1 (a brick is one image plane thick, 9 taps -> 5 tap pairs);
//  * anything else (the 5x5 stride-2 RGB stem, exact-fp32 mode): conv2d_direct_kernel.
#include "common.hpp"
#include "conv_common.hpp"
#ifdef MVSGI_STAMPS
#include <cstdio>
#include <cstdlib>
#endif

namespace {

#include "conv3d_bf16x3.hpp"
#include "conv3d_f32mfma.hpp"
#include "resblock2d_bf16x3.hpp"

struct Conv2dArgs {
    const float* x;
    const float* w;      // [Cout][Cin][k][k]
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int B, Cin, Hin, Win, Cout, Ho, Wo, k, stride, in_nchw;
    float neg_slope;
    unsigned char* ys;   // stem kernels: when set, the output goes here in the 2-D split-padded format (resblock2d_rs.hip) instead of y
};

// 16 channels of one pixel as a 2-D split-padded record [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15]: couts 4 q .. 4 q + 3
__device__ __forceinline__ void store_split2d_quad(unsigned char* ys, int b, int Ho, int Wo, int oy, int ox, int q, const f32x4 r) {
    u32x2 hi, lo;
    split_bf16x4(r, hi, lo);
    unsigned char* p = ys + (((long long)b * (Ho + 4) + oy + 2) * (Wo + 4) + ox + 2) * 64 + (q >> 1) * 16 + (q & 1) * 8;
    *reinterpret_cast<u32x2*>(p) = hi;
    *reinterpret_cast<u32x2*>(p + 32) = lo;
}

// thread = (pixel, CO consecutive couts); weights are wave-uniform reads
template <int CO>
__global__ __launch_bounds__(256) void conv2d_direct_kernel(Conv2dArgs a) {
    const int groups = (a.Cout + CO - 1) / CO;
    const long long total = (long long)a.B * a.Ho * a.Wo * groups;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % groups);
    const long long pix = idx / groups;
    const int ow = (int)(pix % a.Wo);
    const int oh = (int)((pix / a.Wo) % a.Ho);
    const int b = (int)(pix / ((long long)a.Wo * a.Ho));
    const int co0 = cg * CO, pad = a.k / 2;
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    for (int ky = 0; ky < a.k; ++ky) {
        const int ih = oh * a.stride - pad + ky;
        if (ih < 0 || ih >= a.Hin) continue;
        for (int kx = 0; kx < a.k; ++kx) {
            const int iw = ow * a.stride - pad + kx;
            if (iw < 0 || iw >= a.Win) continue;
            for (int ci = 0; ci < a.Cin; ++ci) {
                const float xv = a.in_nchw ? a.x[(((long long)b * a.Cin + ci) * a.Hin + ih) * a.Win + iw]
                                           : a.x[(((long long)b * a.Hin + ih) * a.Win + iw) * a.Cin + ci];
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    const int co = co0 + o;
                    if (co < a.Cout) acc[o] = fmaf(xv, a.w[(((long long)co * a.Cin + ci) * a.k + ky) * a.k + kx], acc[o]);
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        const int co = co0 + o;
        if (co >= a.Cout) break;
        float r = acc[o] * a.scale[co] + a.shift[co];
        if (a.res) r += a.res[pix * a.Cout + co];
        r = r > 0.f ? r : r * a.neg_slope;
        a.y[pix * a.Cout + co] = r;
    }
}

// RGB stem of the extractor (simple_feature_extractor.py:31-38): 5x5, stride 2, 3 -> 16 channels, NCHW
// images in, channels-last features out.  A workgroup owns a 16x16 patch of output pixels: the
// 35x35x3 input window and the 75x16 weights (transposed to [tap][cout]) sit in LDS, every thread
// accumulates the 16 output channels of one pixel (1200 FMAs against 75 + 300 LDS reads).
// U8 = true: the images arrive as the camera driver delivers them, uint8 [B][H][W][3], and are
// converted as the reference does (api/inference_class.py:104-107: .float() / 255.0) on the way into LDS.
template <bool U8>
__global__ __launch_bounds__(256) void conv2d_stem_kernel(Conv2dArgs a) {
    constexpr int T = 16, IT = 2 * T + 3, CI = 3, K = 5, CO = 16;
    __shared__ float xs[CI][IT][IT + 1];
    __shared__ __attribute__((aligned(16))) float ws[CI * K * K][CO];
    const int tid = threadIdx.x, tx = tid % T, ty = tid / T;
    const int ox0 = blockIdx.x * T, oy0 = blockIdx.y * T, b = blockIdx.z;
    for (int e = tid; e < CI * K * K * CO; e += 256) {
        const int co = e % CO, tap = e / CO;                    // tap = (ci*5 + ky)*5 + kx
        ws[tap][co] = a.w[co * (CI * K * K) + tap];
    }
    const int ix0 = ox0 * 2 - 2, iy0 = oy0 * 2 - 2;
    for (int e = tid; e < CI * IT * IT; e += 256) {
        const int x = e % IT, y = (e / IT) % IT, ci = e / (IT * IT);
        const int gx = ix0 + x, gy = iy0 + y;
        float v = 0.f;
        if (gx >= 0 && gx < a.Win && gy >= 0 && gy < a.Hin) {
            if (U8)
                v = (float)reinterpret_cast<const unsigned char*>(a.x)[(((long long)b * a.Hin + gy) * a.Win + gx) * CI + ci] / 255.0f;
            else
                v = a.x[(((long long)b * CI + ci) * a.Hin + gy) * a.Win + gx];
        }
        xs[ci][y][x] = v;
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float xv = xs[ci][2 * ty + ky][2 * tx + kx];
                const f32x4* wv = reinterpret_cast<const f32x4*>(ws[(ci * K + ky) * K + kx]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += wv[q] * xv;
            }
    const int ox = ox0 + tx, oy = oy0 + ty;
    if (ox >= a.Wo || oy >= a.Ho) return;
    float* yp = a.y + (((long long)b * a.Ho + oy) * a.Wo + ox) * CO;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 r = acc[q] * *reinterpret_cast<const f32x4*>(a.scale + 4 * q) + *reinterpret_cast<const f32x4*>(a.shift + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : r[e] * a.neg_slope;
        if (a.ys) store_split2d_quad(a.ys, b, a.Ho, a.Wo, oy, ox, q, r);
        else *reinterpret_cast<f32x4*>(yp + 4 * q) = r;
    }
}

// The same stem for camera images (uint8 [B][H][W][3]) on the matrix cores.  An 8-bit pixel is exact in bf16, so the only
// rounding left is in the weights: w / 255 is cut into THREE bf16 pieces (24 bits = the whole fp32 significand) and a k-step
// is three MFMAs whose products are exact in the fp32 accumulator -- closer to the real-number result than the reference's
// own fp32 sum.  K runs over (kernel row, byte of the row's 15-byte HWC segment): k = 16 ky + j, j = 3 kx + ch, the 16th byte
// and the sixth row carry zero weights, so a k-step of 32 is two image rows and a lane's eight operands are eight consecutive
// BYTES of one row.  A block computes 8 rows x 64 pixels from 19 staged rows (dword copies, whole dwords in or out of the image
// because 3 W is a multiple of 4); A = weights (cout x k, in registers for the whole kernel), B = pixels, so a lane ends up with
// four consecutive couts of one pixel and a wave stores a contiguous KiB.
struct StemArgs {
    const unsigned char* x;
    const bf16x8* wp;          // [3 k-steps][3 pieces][64 lanes] (stem_pack_weights_kernel)
    const float* scale;
    const float* shift;
    float* y;
    int B, Hin, Win, Ho, Wo;
    float neg_slope;
    unsigned char* ys;         // 2-D split-padded output instead of y
};

constexpr int kStemRows = 8, kStemCols = 64, kStemRowDw = 100;

__global__ void stem_pack_weights_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (kstep * 3 + piece) * 64 + lane
    if (idx >= 9 * 64) return;
    const int lane = idx & 63, piece = (idx >> 6) % 3, ks = idx / 192;
    const int co = lane & 15, g = lane >> 4;
    bf16x8 out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ky = 2 * ks + (g >> 1), j = 8 * (g & 1) + e;
        float v = 0.f;
        if (ky < 5 && j < 15) v = w[((co * 3 + j % 3) * 5 + ky) * 5 + j / 3] / 255.0f;
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        out[e] = piece == 0 ? h : piece == 1 ? m : (__bf16)r2;
    }
    wp[idx] = out;
}

__global__ __launch_bounds__(256) void conv2d_stem_u8_mfma_kernel(StemArgs a) {
    __shared__ unsigned rows[(2 * kStemRows + 3) * kStemRowDw];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ox0 = blockIdx.x * kStemCols, oy0 = blockIdx.y * kStemRows, b = blockIdx.z;
    const int rowB = a.Win * 3;
    const int bo0 = 6 * ox0 - 6;                      // first byte of the tile's window in an image row (-6 at the left edge)
    const int as = bo0 & ~3, phase = bo0 - as;        // aligned start; bo0 = 6 (64 t - 1) = 2 mod 4 -> phase 2
    const unsigned char* img = a.x + (long long)b * a.Hin * rowB;
    for (int e = tid; e < (2 * kStemRows + 3) * kStemRowDw; e += 256) {
        const int r = e / kStemRowDw, dw = e - r * kStemRowDw;
        const int gy = 2 * oy0 - 2 + r, byte = as + 4 * dw;
        unsigned v = 0u;
        if (gy >= 0 && gy < a.Hin && byte >= 0 && byte < rowB) v = *reinterpret_cast<const unsigned*>(img + (long long)gy * rowB + byte);
        rows[e] = v;
    }
    bf16x8 wA[3][3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) wA[ks][pc] = a.wp[(ks * 3 + pc) * 64 + lane];
    const int m = lane & 15, g = lane >> 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + 4 * g), sh = *reinterpret_cast<const f32x4*>(a.shift + 4 * g);
    __syncthreads();
#pragma unroll 1
    for (int t = wave; t < kStemRows * (kStemCols / 16); t += 4) {
        const int ry = t >> 2, px = (t & 3) * 16 + m;              // output row / pixel inside the tile
        const int bo = phase + 6 * px + 8 * (g & 1);               // byte of the lane's eight inside a staged row
        const int dq = bo >> 2, shb = (bo & 3) * 8;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            int ky = 2 * ks + (g >> 1);
            ky = ky < 4 ? ky : 4;                                   // (the sixth row's weights are zero: any finite data do)
            const unsigned* rp = rows + (2 * ry + ky) * kStemRowDw + dq;
            const unsigned d0 = rp[0], d1 = rp[1], d2 = rp[2];
            const unsigned lo = __builtin_amdgcn_alignbit(d1, d0, shb), hi = __builtin_amdgcn_alignbit(d2, d1, shb);
            typedef float f2_t __attribute__((ext_vector_type(2)));
            typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
            typedef unsigned u4_t __attribute__((ext_vector_type(4)));
            u4_t pk;
            pk[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{(float)(lo & 0xffu), (float)((lo >> 8) & 0xffu)}, b2_t));
            pk[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{(float)((lo >> 16) & 0xffu), (float)(lo >> 24)}, b2_t));
            pk[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{(float)(hi & 0xffu), (float)((hi >> 8) & 0xffu)}, b2_t));
            pk[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{(float)((hi >> 16) & 0xffu), (float)(hi >> 24)}, b2_t));
            const bf16x8 xb = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
            for (int pc = 2; pc >= 0; --pc) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[ks][pc], xb, acc, 0, 0, 0);   // small pieces first
        }
        const int ox = ox0 + px, oy = oy0 + ry;
        f32x4 r = acc * sc + sh;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : r[e] * a.neg_slope;
        if (a.ys) {
            // lanes g and g ^ 1 trade halves (all 64 lanes take part): g even ends up with hi / lo of channels 8 (g >> 1) .. + 7,
            // one 16-byte store per lane and a contiguous KiB per wave, as the fp32 form has
            u32x2 hi, lo;
            split_bf16x4(r, hi, lo);
            typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
            const u32x2 sa = __builtin_amdgcn_permlane16_swap(hi[0], lo[0], false, false);
            const u32x2 sb = __builtin_amdgcn_permlane16_swap(hi[1], lo[1], false, false);
            if (ox < a.Wo && oy < a.Ho)        // nt: 1.6 GB per 96 images that this launch never reads back
                __builtin_nontemporal_store(u32x4s{sa[0], sb[0], sa[1], sb[1]},
                    reinterpret_cast<u32x4s*>(a.ys + (((long long)b * (a.Ho + 4) + oy + 2) * (a.Wo + 4) + ox + 2) * 64 + (g & 1) * 32 + (g >> 1) * 16));
        } else if (ox < a.Wo && oy < a.Ho) {
            *reinterpret_cast<f32x4*>(a.y + (((long long)b * a.Ho + oy) * a.Wo + ox) * 16 + 4 * g) = r;
        }
    }
}

enum Variant2d { D2_DIRECT, D2_N16, D2_N32, D2_N64, D2_S2_N32, D2_S2_N64, D2_F32_N16, D2_F32_N32, D2_F32_S2, D2_COUNT };
const char* const kNames2d[D2_COUNT] = {
    "conv2d_direct_kernel<4>",
    "conv3d_bf16x3_kernel<1, 4, 4, 1, 1, 16, 16, 1, 1, false, false, false, false>", "conv3d_bf16x3_kernel<2, 4, 4, 1, 1, 16, 16, 1, 1, false, false, false, false>",
    "conv3d_bf16x3_kernel<2, 4, 2, 2, 1, 8, 16, 1, 1, false, false, false, false>", "conv3d_bf16x3_kernel<2, 1, 4, 1, 1, 4, 16, 2, 1, false, false, false, false>",
    "conv3d_bf16x3_kernel<2, 2, 2, 2, 1, 4, 16, 2, 1, false, false, false, false>",
    "conv3d_mfma_kernel<1, 4, 4, 1, 1, 16, 16, 1, 1>", "conv3d_mfma_kernel<2, 4, 4, 1, 1, 16, 16, 1, 1>",
    "conv3d_mfma_kernel<2, 1, 4, 1, 1, 4, 16, 2, 1>",
};

int select2d(int Cin, int Cout, int k, int stride, int impl, const void* w, const void* wp, int in_nchw) {
    const bool mfma_ok = k == 3 && Cin % 16 == 0 && Cout % 16 == 0 && !in_nchw && (stride == 1 || stride == 2);
    if (impl == MVSGI_CONV_BF16X3 && mfma_ok) {
        if (!wp) { mvsgi::fail("mvsgi_conv2d_f32: bf16x3 path needs w_packed"); return D2_COUNT; }
        const int CT = Cout / 16;
        if (stride == 2) return CT <= 2 ? D2_S2_N32 : D2_S2_N64;
        return CT == 1 ? D2_N16 : CT <= 3 ? D2_N32 : D2_N64;
    }
    if (impl == MVSGI_CONV_MFMA && mfma_ok && (Cout == 16 || Cout == 32)) {
        if (!wp) { mvsgi::fail("mvsgi_conv2d_f32: fp32 MFMA path needs w_packed"); return D2_COUNT; }
        if (stride == 2) return D2_F32_S2;
        return Cout == 16 ? D2_F32_N16 : D2_F32_N32;
    }
    if (impl != MVSGI_CONV_AUTO && impl != MVSGI_CONV_DIRECT && impl != MVSGI_CONV_BF16X3 && impl != MVSGI_CONV_MFMA) {
        mvsgi::fail("mvsgi_conv2d_f32: unknown impl %d", impl);
        return D2_COUNT;
    }
    if (!w) { mvsgi::fail("mvsgi_conv2d_f32: direct path needs w_oihw"); return D2_COUNT; }
    return D2_DIRECT;
}

}  // namespace

extern "C" size_t mvsgi_conv2d_packed_weight_bytes_bf16x3(int Cout, int Cin) {
    return (size_t)(Cin / 16) * pairs_of(1) * (size_t)(Cout / 16) * 2 * 64 * 16;
}

extern "C" int mvsgi_conv2d_pack_weights_bf16x3(const float* w_oihw, void* w_packed, int Cout, int Cin,
                                                mvsgi_stream_t stream) {
    MVSGI_REQUIRE(w_oihw && w_packed, "mvsgi_conv2d_pack_weights_bf16x3: null pointer");
    MVSGI_REQUIRE(Cout > 0 && Cin > 0 && Cout % 16 == 0 && Cin % 16 == 0,
                  "mvsgi_conv2d_pack_weights_bf16x3: Cout=%d Cin=%d must be positive multiples of 16", Cout, Cin);
    const long long total = (long long)(Cin / 16) * pairs_of(1) * (Cout / 16) * 64;
    hipLaunchKernelGGL(pack_weights_bf16x3_kernel, dim3((unsigned)mvsgi::cdiv(total, 256)), dim3(256), 0,
                       mvsgi::as_stream(stream), w_oihw, reinterpret_cast<bf16x8*>(w_packed), Cout, Cin, 9, false);
    return mvsgi::check_launch("mvsgi_conv2d_pack_weights_bf16x3");
}

// 5x5 stride-2 3 -> 16 RGB stem on uint8 images: weights / 255 in three bf16 pieces, MFMA operand order
extern "C" size_t mvsgi_conv2d_stem_packed_weight_bytes(void) { return (size_t)9 * 64 * 16; }

extern "C" int mvsgi_conv2d_stem_pack_weights(const float* w_oihw, void* w_packed, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(w_oihw && w_packed, "mvsgi_conv2d_stem_pack_weights: null pointer");
    hipLaunchKernelGGL(stem_pack_weights_kernel, dim3(3), dim3(192), 0, mvsgi::as_stream(stream), w_oihw,
                       reinterpret_cast<bf16x8*>(w_packed));
    return mvsgi::check_launch("mvsgi_conv2d_stem_pack_weights");
}

extern "C" size_t mvsgi_conv2d_packed_weight_floats(int Cout, int Cin) { return (size_t)9 * (size_t)Cout * (size_t)Cin; }

extern "C" int mvsgi_conv2d_pack_weights_f32(const float* w_oihw, float* w_packed, int Cout, int Cin, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(w_oihw && w_packed, "mvsgi_conv2d_pack_weights_f32: null pointer");
    MVSGI_REQUIRE(Cout > 0 && Cin > 0 && Cout % 16 == 0 && Cin % 16 == 0,
                  "mvsgi_conv2d_pack_weights_f32: Cout=%d Cin=%d must be positive multiples of 16", Cout, Cin);
    const long long total = (long long)(Cin / 16) * 9 * (Cout / 16) * 64;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)mvsgi::cdiv(total, 256)), dim3(256), 0,
                       mvsgi::as_stream(stream), w_oihw, reinterpret_cast<f32x4*>(w_packed), Cout, Cin, 9);
    return mvsgi::check_launch("mvsgi_conv2d_pack_weights_f32");
}

extern "C" const char* mvsgi_conv2d_variant_f32(int Cin, int Cout, int ksize, int stride, int impl, int in_nchw) {
    static const float dummy = 0.f;
    const int v = select2d(Cin, Cout, ksize, stride, impl, &dummy, &dummy, in_nchw);
    return v == D2_COUNT ? nullptr : kNames2d[v];
}

namespace {
int conv2d_run(const float* x, const float* w_oihw, const void* w_packed, const float* scale,
               const float* shift, const float* res, float* y, unsigned char* ys, int B, int Cin, int Hin, int Win,
               int Cout, int ksize, int stride, float neg_slope, int impl, int in_nchw,
               mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x && (y || ys) && scale && shift, "mvsgi_conv2d_f32: null pointer");
    MVSGI_REQUIRE(!ys || Cout == 16, "mvsgi_conv2d_f32_out_split2d: the 2-D split-padded format holds 16 channels, got Cout = %d", Cout);
    MVSGI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0, "mvsgi_conv2d_f32: bad dims");
    MVSGI_REQUIRE(ksize >= 1 && (ksize & 1) && ksize <= 7, "mvsgi_conv2d_f32: kernel size %d not odd in [1, 7]", ksize);
    MVSGI_REQUIRE(stride == 1 || stride == 2, "mvsgi_conv2d_f32: stride %d not in {1, 2}", stride);
    const int pad = ksize / 2;
    const int Ho = (Hin + 2 * pad - ksize) / stride + 1, Wo = (Win + 2 * pad - ksize) / stride + 1;
    const int v = select2d(Cin, Cout, ksize, stride, impl, w_oihw, w_packed, in_nchw);
    if (v == D2_COUNT) return 1;
    hipStream_t st = mvsgi::as_stream(stream);
    if (v == D2_DIRECT) {
        Conv2dArgs a{x, w_oihw, scale, shift, res, y, B, Cin, Hin, Win, Cout, Ho, Wo, ksize, stride, in_nchw, neg_slope, ys};
        if (ksize == 5 && stride == 2 && Cin == 3 && Cout == 16 && in_nchw && !res && B < 65536) {
            const dim3 grid((unsigned)mvsgi::cdiv(Wo, 16), (unsigned)mvsgi::cdiv(Ho, 16), (unsigned)B);
            if (in_nchw == 2 && w_packed && Win % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 3) == 0) {
                StemArgs sa{reinterpret_cast<const unsigned char*>(x), reinterpret_cast<const bf16x8*>(w_packed), scale, shift, y,
                            B, Hin, Win, Ho, Wo, neg_slope, ys};
                const dim3 gm((unsigned)mvsgi::cdiv(Wo, kStemCols), (unsigned)mvsgi::cdiv(Ho, kStemRows), (unsigned)B);
                hipLaunchKernelGGL(conv2d_stem_u8_mfma_kernel, gm, dim3(256), 0, st, sa);
                return mvsgi::check_launch("mvsgi_conv2d_f32(stem, matrix cores)");
            }
            if (in_nchw == 2)
                hipLaunchKernelGGL(conv2d_stem_kernel<true>, grid, dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL(conv2d_stem_kernel<false>, grid, dim3(256), 0, st, a);
            return mvsgi::check_launch("mvsgi_conv2d_f32(stem)");
        }
        MVSGI_REQUIRE(in_nchw != 2, "mvsgi_conv2d_f32: uint8 HWC input is implemented for the 5x5 stride-2 3->16 stem only");
        MVSGI_REQUIRE(!ys, "mvsgi_conv2d_f32_out_split2d: not available on the direct kernel (stem and split-bf16 kernels only)");
        const long long total = (long long)B * Ho * Wo * mvsgi::cdiv(Cout, 4);
        hipLaunchKernelGGL((conv2d_direct_kernel<4>), dim3((unsigned)mvsgi::cdiv(total, 256)), dim3(256), 0, st, a);
        return mvsgi::check_launch("mvsgi_conv2d_f32(direct)");
    }
    // images as a volume whose D planes do not interact: [B][H][W][C] == [1][B][H][W][C]
    ConvArgs a{};
    a.x = x;
    a.wp = reinterpret_cast<const f32x4*>(w_packed);
    a.scale = scale;
    a.shift = shift;
    a.res = res;
    a.y = ys ? reinterpret_cast<float*>(ys) : y;      // (never dereferenced when y_split is set)
    a.y_split = ys;
    a.ys_2d = ys != nullptr;
    MVSGI_REQUIRE(!ys || v <= D2_S2_N64, "mvsgi_conv2d_f32_out_split2d: split-bf16 kernels only");
    // one volume of B planes while its byte offsets fit 32 bits, else B one-plane frames (64-bit frame bases)
    const bool planes = (long long)B * Hin * Win * Cin < (1ll << 29) && (long long)B * Ho * Wo * Cout < (1ll << 31) &&
                        (!ys || (long long)(B + 2) * (Ho + 4) * (Wo + 4) * 64 < (1ll << 31));
    a.B = planes ? 1 : B;
    a.Cin = Cin;
    a.Din = planes ? B : 1;
    a.Hin = Hin;
    a.Win = Win;
    a.Cout = Cout;
    a.stride = stride;
    a.neg_slope = neg_slope;
    a.Do = a.Din;
    a.Ho = Ho;
    a.Wo = Wo;
    switch (v) {
        case D2_N16: return launch_bf16x3<1, 4, 4, 1, 1, 16, 16, 1, 1>(a, st);
        case D2_N32: return launch_bf16x3<2, 4, 4, 1, 1, 16, 16, 1, 1>(a, st);
        case D2_N64: return launch_bf16x3<2, 4, 2, 2, 1, 8, 16, 1, 1>(a, st);
        case D2_S2_N32: return launch_bf16x3<2, 1, 4, 1, 1, 4, 16, 2, 1>(a, st);
        case D2_S2_N64: return launch_bf16x3<2, 2, 2, 2, 1, 4, 16, 2, 1>(a, st);
        case D2_F32_N16: return launch_mfma<1, 4, 4, 1, 1, 16, 16, 1, 1>(a, st);
        case D2_F32_N32: return launch_mfma<2, 4, 4, 1, 1, 16, 16, 1, 1>(a, st);
        case D2_F32_S2: return launch_mfma<2, 1, 4, 1, 1, 4, 16, 2, 1>(a, st);
    }
    return mvsgi::fail("mvsgi_conv2d_f32: bad variant %d", v);
}
}  // namespace

extern "C" int mvsgi_conv2d_f32(const float* x, const float* w_oihw, const void* w_packed, const float* scale,
                                const float* shift, const float* res, float* y, int B, int Cin, int Hin, int Win,
                                int Cout, int ksize, int stride, float neg_slope, int impl, int in_nchw,
                                mvsgi_stream_t stream) {
    MVSGI_REQUIRE(y, "mvsgi_conv2d_f32: null pointer");
    return conv2d_run(x, w_oihw, w_packed, scale, shift, res, y, nullptr, B, Cin, Hin, Win, Cout, ksize, stride, neg_slope, impl, in_nchw, stream);
}

// the same layer with the output in the 2-D split-padded format of resblock2d_rs.hip ([B][Ho+4][Wo+4][64 B], Cout == 16; the caller
// zeroes the buffer once, only the interior is written): the RGB stem (either input layout) and the split-bf16 3x3 kernels.
extern "C" int mvsgi_conv2d_f32_out_split2d(const float* x, const float* w_oihw, const void* w_packed, const float* scale,
                                            const float* shift, const float* res, void* y_split, int B, int Cin, int Hin, int Win,
                                            int Cout, int ksize, int stride, float neg_slope, int impl, int in_nchw,
                                            mvsgi_stream_t stream) {
    MVSGI_REQUIRE(y_split, "mvsgi_conv2d_f32_out_split2d: null pointer");
    return conv2d_run(x, w_oihw, w_packed, scale, shift, res, nullptr, static_cast<unsigned char*>(y_split), B, Cin, Hin, Win, Cout, ksize,
                      stride, neg_slope, impl, in_nchw, stream);
}

// ResConvBlk2d.forward (common/common_modules.py:165-176) for 16 -> 16 channels, 3x3, stride 1, in one launch:
//   y = act( conv2(act(conv1(x) * scale1 + shift1)) * scale2 + shift2 + x )
// x / y [N][H][W][16] channels-last; w_packed1 / w_packed2 from mvsgi_conv2d_pack_weights_bf16x3 (Cout = Cin = 16).
extern "C" int mvsgi_resblock2d_f32(const float* x, const void* w_packed1, const float* scale1, const float* shift1,
                                    const void* w_packed2, const float* scale2, const float* shift2, float* y, int N,
                                    int H, int W, float neg_slope, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x && w_packed1 && w_packed2 && scale1 && shift1 && scale2 && shift2 && y, "mvsgi_resblock2d_f32: null pointer");
    MVSGI_REQUIRE(N > 0 && H > 0 && W > 0, "mvsgi_resblock2d_f32: non-positive dimension");
    MVSGI_REQUIRE((long long)H * W * 16 < (1ll << 31), "mvsgi_resblock2d_f32: image too large for 32-bit element offsets");
    MVSGI_REQUIRE(x != y, "mvsgi_resblock2d_f32: in-place operation is not supported (bricks read their neighbours' inputs)");
    ResBlk2dArgs a{};
    a.x = x; a.y = y;
    a.wp1 = static_cast<const f32x4*>(w_packed1); a.wp2 = static_cast<const f32x4*>(w_packed2);
    a.scale1 = scale1; a.shift1 = shift1; a.scale2 = scale2; a.shift2 = shift2;
    a.N = N; a.H = H; a.W = W; a.neg_slope = neg_slope;
    a.tiles_h = (int)mvsgi::cdiv(H, 14);
    a.tiles_w = (int)mvsgi::cdiv(W, 30);
    const long long nb = (long long)N * a.tiles_h * a.tiles_w;
    MVSGI_REQUIRE(nb < (1ll << 31), "mvsgi_resblock2d_f32: too many bricks");
    a.total_units = (int)nb;
    constexpr size_t lds_bytes = (size_t)(2 * 18 * 34 + 18 * 34) * kVSB;     // input window x 2 + conv1 result
    static mvsgi::PersistentGeom geo_cache[mvsgi::kMaxDevices] = {};
    mvsgi::PersistentGeom geo;
    if (mvsgi::persistent_geometry(resblock2d_bf16x3_kernel, 512, lds_bytes, 2, geo_cache, "mvsgi_resblock2d_f32", geo)) return 1;
    const long long resident = ((long long)geo.cus * geo.wgs_per_cu) / 8 * 8 > 0 ? ((long long)geo.cus * geo.wgs_per_cu) / 8 * 8 : 8;
    hipLaunchKernelGGL(resblock2d_bf16x3_kernel, dim3((unsigned)(nb <= resident ? nb : resident)), dim3(512), lds_bytes,
                       mvsgi::as_stream(stream), a);
    return mvsgi::check_launch("mvsgi_resblock2d_f32");
}
