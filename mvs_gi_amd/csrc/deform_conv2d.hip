// Deformable 2-D convolution with a given offset field: the compute of
// SphereConvEquirect2d.forward (dsta_mvs/model/common/common_modules.py:411-425), which calls
// torchvision.ops.deform_conv2d(input, offset, weight, bias, stride, padding, dilation, mask=None) with the
// precomputed sphere offsets of gen_offset (:427-507), followed by SphereConvBlk's norm / residual /
// activation (:509-547) fused into the epilogue.
//
// torchvision is not part of the reference checkout (and not installed here), so the operator is restated
// from its published definition (Dai et al. 2017, torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp):
//   out[n, co, ho, wo] = sum_{ci, i, j} w[co, ci, i, j] * bilinear(in[n, ci], y, x)
//   y = ho*stride_h - pad_h + i*dil_h + offset[n, 2*(i*Kw + j),     ho, wo]
//   x = wo*stride_w - pad_w + j*dil_w + offset[n, 2*(i*Kw + j) + 1, ho, wo]
//   bilinear: 0 if y <= -1 || y >= H || x <= -1 || x >= W; else the four corners floor / floor + 1 with
//   weights (1-ly)(1-lx), (1-ly)lx, ly(1-lx), ly*lx, a corner outside the image contributing 0.
//
// Layout: channels-last activations [N][H][W][C] (what the 2-D conv kernels produce and the sweep
// consumes); one 64-byte texel per corner for C = 16.  C == 16 -> 16: four lanes share an output pixel,
// lane q samples channels 4q..4q+3 and owns output channels 4q..4q+3; the sampled quads travel through DPP
// quad broadcasts, the [tap][ci][co] weights sit in LDS.  Other channel counts: one thread per (pixel, co).
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DeformArgs {
    const float* x;
    const float* offset;
    const float* wp;      // [K][Cin][Cout]
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int N, Cin, H, W, Cout, Ho, Wo, Kh, Kw, sh, sw, ph, pw, dh, dw;
    long long offset_image_stride;   // 0: one offset field shared by every image
    float neg_slope;
};

struct Corners {
    int o1, o2, o3, o4;          // element offsets (y*W + x) of the corners, -1 = contributes 0
    float w1, w2, w3, w4;
};

__device__ __forceinline__ Corners corners_of(float y, float x, int H, int W) {
#pragma clang fp contract(off)
    Corners c;
    if (y <= -1.0f || (float)H <= y || x <= -1.0f || (float)W <= x || !(y == y) || !(x == x)) {
        c.o1 = c.o2 = c.o3 = c.o4 = -1;
        c.w1 = c.w2 = c.w3 = c.w4 = 0.f;
        return c;
    }
    const int yl = (int)floorf(y), xl = (int)floorf(x);
    const int yh = yl + 1, xh = xl + 1;
    const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
    c.o1 = (yl >= 0 && xl >= 0) ? yl * W + xl : -1;
    c.o2 = (yl >= 0 && xh <= W - 1) ? yl * W + xh : -1;
    c.o3 = (yh <= H - 1 && xl >= 0) ? yh * W + xl : -1;
    c.o4 = (yh <= H - 1 && xh <= W - 1) ? yh * W + xh : -1;
    c.w1 = hy * hx;
    c.w2 = hy * lx;
    c.w3 = ly * hx;
    c.w4 = ly * lx;
    return c;
}

template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), Q * 0x55, 0xf, 0xf, true));
}

// Cin == Cout == 16; block = 64 pixels x 4 lanes
__global__ __launch_bounds__(256) void deform_conv2d_c16_kernel(DeformArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float wl[];     // [K][16][16]
    const int K = a.Kh * a.Kw;
    for (int i = threadIdx.x; i < K * 256; i += 256) wl[i] = a.wp[i];
    __syncthreads();
    const int q = threadIdx.x & 3;
    const long long HW = (long long)a.Ho * a.Wo;
    long long pix = (long long)blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool live = pix < (long long)a.N * HW;
    if (!live) pix = (long long)a.N * HW - 1;             // whole quads stay active for the DPP exchange
    const int n = (int)(pix / HW);
    const long long p = pix - (long long)n * HW;
    const int ho = (int)(p / a.Wo), wo = (int)(p - (long long)ho * a.Wo);
    const float* img = a.x + (long long)n * a.H * a.W * 16 + q * 4;
    const float* off = a.offset + (long long)n * a.offset_image_stride + p;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < K; ++t) {
        const int i = t / a.Kw, j = t - i * a.Kw;
        const float y = (float)(ho * a.sh - a.ph + i * a.dh) + off[(long long)(2 * t) * HW];
        const float x = (float)(wo * a.sw - a.pw + j * a.dw) + off[(long long)(2 * t + 1) * HW];
        const Corners c = corners_of(y, x, a.H, a.W);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 v1 = c.o1 >= 0 ? *reinterpret_cast<const f32x4*>(img + (long long)c.o1 * 16) : z;
        const f32x4 v2 = c.o2 >= 0 ? *reinterpret_cast<const f32x4*>(img + (long long)c.o2 * 16) : z;
        const f32x4 v3 = c.o3 >= 0 ? *reinterpret_cast<const f32x4*>(img + (long long)c.o3 * 16) : z;
        const f32x4 v4 = c.o4 >= 0 ? *reinterpret_cast<const f32x4*>(img + (long long)c.o4 * 16) : z;
        const f32x4 s = ((v1 * c.w1 + v2 * c.w2) + v3 * c.w3) + v4 * c.w4;
        const float* wt = wl + t * 256 + q * 4;            // [ci][co 4q..4q+3]
#define MVSGI_DC_ACC(QQ)                                                                            \
        {                                                                                           \
            const float s0 = quad_bcast<QQ>(s[0]), s1 = quad_bcast<QQ>(s[1]), s2 = quad_bcast<QQ>(s[2]), \
                        s3 = quad_bcast<QQ>(s[3]);                                                  \
            acc += *reinterpret_cast<const f32x4*>(wt + (4 * QQ + 0) * 16) * s0;                    \
            acc += *reinterpret_cast<const f32x4*>(wt + (4 * QQ + 1) * 16) * s1;                    \
            acc += *reinterpret_cast<const f32x4*>(wt + (4 * QQ + 2) * 16) * s2;                    \
            acc += *reinterpret_cast<const f32x4*>(wt + (4 * QQ + 3) * 16) * s3;                    \
        }
        MVSGI_DC_ACC(0)
        MVSGI_DC_ACC(1)
        MVSGI_DC_ACC(2)
        MVSGI_DC_ACC(3)
#undef MVSGI_DC_ACC
    }
    if (!live) return;
    f32x4 r = acc * *reinterpret_cast<const f32x4*>(a.scale + q * 4) + *reinterpret_cast<const f32x4*>(a.shift + q * 4);
    if (a.res) r += *reinterpret_cast<const f32x4*>(a.res + pix * 16 + q * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : r[e] * a.neg_slope;
    *reinterpret_cast<f32x4*>(a.y + pix * 16 + q * 4) = r;
}

// any channel counts: thread = (pixel, co)
__global__ __launch_bounds__(256) void deform_conv2d_generic_kernel(DeformArgs a) {
#pragma clang fp contract(off)
    const long long HW = (long long)a.Ho * a.Wo;
    const long long total = (long long)a.N * HW * a.Cout;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int co = (int)(idx % a.Cout);
    const long long pix = idx / a.Cout;
    const int n = (int)(pix / HW);
    const long long p = pix - (long long)n * HW;
    const int ho = (int)(p / a.Wo), wo = (int)(p - (long long)ho * a.Wo);
    const float* img = a.x + (long long)n * a.H * a.W * a.Cin;
    const float* off = a.offset + (long long)n * a.offset_image_stride + p;
    const int K = a.Kh * a.Kw;
    float acc = 0.f;
    for (int t = 0; t < K; ++t) {
        const int i = t / a.Kw, j = t - i * a.Kw;
        const float y = (float)(ho * a.sh - a.ph + i * a.dh) + off[(long long)(2 * t) * HW];
        const float x = (float)(wo * a.sw - a.pw + j * a.dw) + off[(long long)(2 * t + 1) * HW];
        const Corners c = corners_of(y, x, a.H, a.W);
        for (int ci = 0; ci < a.Cin; ++ci) {
            const float v1 = c.o1 >= 0 ? img[(long long)c.o1 * a.Cin + ci] : 0.f;
            const float v2 = c.o2 >= 0 ? img[(long long)c.o2 * a.Cin + ci] : 0.f;
            const float v3 = c.o3 >= 0 ? img[(long long)c.o3 * a.Cin + ci] : 0.f;
            const float v4 = c.o4 >= 0 ? img[(long long)c.o4 * a.Cin + ci] : 0.f;
            const float s = ((v1 * c.w1 + v2 * c.w2) + v3 * c.w3) + v4 * c.w4;
            acc += a.wp[((long long)t * a.Cin + ci) * a.Cout + co] * s;
        }
    }
    float r = acc * a.scale[co] + a.shift[co];
    if (a.res) r += a.res[pix * a.Cout + co];
    a.y[pix * a.Cout + co] = r > 0.f ? r : r * a.neg_slope;
}

// [Cout][Cin][Kh][Kw] -> [Kh*Kw][Cin][Cout]
__global__ void pack_deform_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int K) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cout * Cin * K) return;
    const int co = idx % Cout, ci = (idx / Cout) % Cin, t = idx / (Cout * Cin);
    wp[idx] = w[((long long)co * Cin + ci) * K + t];
}

}  // namespace

extern "C" int mvsgi_deform_conv2d_pack_weights_f32(const float* w_oihw, float* w_packed, int Cout, int Cin, int Kh,
                                                    int Kw, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(w_oihw && w_packed, "mvsgi_deform_conv2d_pack_weights_f32: null pointer");
    MVSGI_REQUIRE(Cout > 0 && Cin > 0 && Kh > 0 && Kw > 0 && (long long)Cout * Cin * Kh * Kw < (1ll << 30),
                  "mvsgi_deform_conv2d_pack_weights_f32: bad dims");
    const int total = Cout * Cin * Kh * Kw;
    hipLaunchKernelGGL(pack_deform_weights_kernel, dim3((unsigned)mvsgi::cdiv(total, 256)), dim3(256), 0,
                       mvsgi::as_stream(stream), w_oihw, w_packed, Cout, Cin, Kh * Kw);
    return mvsgi::check_launch("mvsgi_deform_conv2d_pack_weights_f32");
}

extern "C" int mvsgi_deform_conv2d_f32(const float* x, const float* offset, int offset_per_image, const float* w_packed,
                                       const float* scale, const float* shift, const float* res, float* y, int N,
                                       int Cin, int H, int W, int Cout, int Kh, int Kw, int stride_h, int stride_w,
                                       int pad_h, int pad_w, int dil_h, int dil_w, float neg_slope,
                                       mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x && offset && w_packed && scale && shift && y, "mvsgi_deform_conv2d_f32: null pointer");
    MVSGI_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && Kh > 0 && Kw > 0 && stride_h > 0 && stride_w > 0 &&
                      dil_h > 0 && dil_w > 0 && pad_h >= 0 && pad_w >= 0,
                  "mvsgi_deform_conv2d_f32: bad dims");
    DeformArgs a{};
    a.x = x; a.offset = offset; a.wp = w_packed; a.scale = scale; a.shift = shift; a.res = res; a.y = y;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Kh = Kh; a.Kw = Kw;
    a.sh = stride_h; a.sw = stride_w; a.ph = pad_h; a.pw = pad_w; a.dh = dil_h; a.dw = dil_w;
    a.Ho = (H + 2 * pad_h - (dil_h * (Kh - 1) + 1)) / stride_h + 1;
    a.Wo = (W + 2 * pad_w - (dil_w * (Kw - 1) + 1)) / stride_w + 1;
    MVSGI_REQUIRE(a.Ho > 0 && a.Wo > 0, "mvsgi_deform_conv2d_f32: empty output");
    MVSGI_REQUIRE((long long)H * W * Cin < (1ll << 31), "mvsgi_deform_conv2d_f32: image too large for 32-bit offsets");
    a.offset_image_stride = offset_per_image ? (long long)2 * Kh * Kw * a.Ho * a.Wo : 0;
    a.neg_slope = neg_slope;
    hipStream_t st = mvsgi::as_stream(stream);
    const long long pixels = (long long)N * a.Ho * a.Wo;
    if (Cin == 16 && Cout == 16 && Kh * Kw <= 32) {
        MVSGI_REQUIRE(mvsgi::cdiv(pixels, 64) < (1ll << 31), "mvsgi_deform_conv2d_f32: too many pixels");
        hipLaunchKernelGGL(deform_conv2d_c16_kernel, dim3((unsigned)mvsgi::cdiv(pixels, 64)), dim3(256),
                           (size_t)Kh * Kw * 256 * sizeof(float), st, a);
    } else {
        MVSGI_REQUIRE(mvsgi::cdiv(pixels * Cout, 256) < (1ll << 31), "mvsgi_deform_conv2d_f32: too many outputs");
        hipLaunchKernelGGL(deform_conv2d_generic_kernel, dim3((unsigned)mvsgi::cdiv(pixels * Cout, 256)), dim3(256), 0, st, a);
    }
    return mvsgi::check_launch("mvsgi_deform_conv2d_f32");
}
