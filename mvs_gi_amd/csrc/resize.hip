// K3: trilinear resize, channels-last fp32 (align_corners=False).
// Replaces the F.interpolate calls of ResizeConv3d.forward
// (dsta_mvs/model/common/common_modules.py:333-350): the x2 upsample in front of every
// up-convolution and, for odd pyramids, the second resize to the skip tensor's size.
// Source index rule (ATen area_pixel_compute_source_index, align_corners=False):
//   src = max(0, (dst + 0.5) * (in / out) - 0.5),  i0 = floor(src), i1 = min(i0 + 1, in - 1).
// Pure HBM-bound streaming kernel: one thread per (output voxel, 4 channels), 16-byte
// loads/stores; the 8 source voxels of neighbouring outputs overlap and hit L1/L2.
#include "common.hpp"

namespace {

struct Axis {
    int i0, i1;
    float l0, l1;
};

__device__ __forceinline__ Axis axis_setup(int dst, int in, float scale) {
    Axis a;
    float src = ((float)dst + 0.5f) * scale - 0.5f;
    src = src < 0.f ? 0.f : src;
    a.i0 = (int)src;
    if (a.i0 > in - 1) a.i0 = in - 1;
    a.i1 = a.i0 + (a.i0 < in - 1 ? 1 : 0);
    a.l1 = src - (float)a.i0;
    a.l0 = 1.0f - a.l1;
    return a;
}

template <int VEC>
__global__ __launch_bounds__(256) void resize_trilinear_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int B, int C, int Di, int Hi, int Wi, int Do, int Ho,
                                                               int Wo, float sd, float sh, float sw) {
    // grid = (ceil(Wo * C/VEC / 256), Ho, B * Do): only one 32-bit division per lane
    const int cg = C / VEC;
    const int xi = blockIdx.x * 256 + threadIdx.x;
    if (xi >= Wo * cg) return;
    const int ow = xi / cg;
    const int c = (xi - ow * cg) * VEC;
    const int oh = blockIdx.y;
    const int b = blockIdx.z / Do, od = blockIdx.z - b * Do;
    const long long vox = (((long long)b * Do + od) * Ho + oh) * Wo + ow;
    const Axis ad = axis_setup(od, Di, sd), ah = axis_setup(oh, Hi, sh), aw = axis_setup(ow, Wi, sw);
    const float* xb = x + (long long)b * Di * Hi * Wi * C + c;
    float r[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) r[e] = 0.f;
#pragma unroll
    for (int zd = 0; zd < 2; ++zd) {
        const int id = zd ? ad.i1 : ad.i0;
        const float ld = zd ? ad.l1 : ad.l0;
        float rh[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) rh[e] = 0.f;
#pragma unroll
        for (int zh = 0; zh < 2; ++zh) {
            const int ih = zh ? ah.i1 : ah.i0;
            const float lh = zh ? ah.l1 : ah.l0;
            const float* p0 = xb + (((long long)id * Hi + ih) * Wi + aw.i0) * C;
            const float* p1 = xb + (((long long)id * Hi + ih) * Wi + aw.i1) * C;
            float v0[VEC], v1[VEC];
            if (VEC == 4) {
                const float4 a0 = *reinterpret_cast<const float4*>(p0);
                const float4 a1 = *reinterpret_cast<const float4*>(p1);
                v0[0] = a0.x; v0[1] = a0.y; v0[2] = a0.z; v0[3] = a0.w;
                v1[0] = a1.x; v1[1] = a1.y; v1[2] = a1.z; v1[3] = a1.w;
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { v0[e] = p0[e]; v1[e] = p1[e]; }
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) rh[e] += lh * (aw.l0 * v0[e] + aw.l1 * v1[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) r[e] += ld * rh[e];
    }
    float* yp = y + vox * C + c;
    if (VEC == 4) {
        *reinterpret_cast<float4*>(yp) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) yp[e] = r[e];
    }
}

}  // namespace

extern "C" int mvsgi_resize_trilinear_f32(const float* x, float* y, int B, int C, int Di, int Hi, int Wi, int Do,
                                          int Ho, int Wo, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x && y, "mvsgi_resize_trilinear_f32: null pointer");
    MVSGI_REQUIRE(B > 0 && C > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0,
                  "mvsgi_resize_trilinear_f32: non-positive dimension");
    const float sd = (float)Di / (float)Do, sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
    hipStream_t st = mvsgi::as_stream(stream);
    MVSGI_REQUIRE(Ho < 65536 && (long long)B * Do < 65536 && (long long)Wo * C < (1ll << 31),
                  "mvsgi_resize_trilinear_f32: dimensions exceed the launch geometry");
    if (C % 4 == 0) {
        const dim3 grid((unsigned)mvsgi::cdiv((long long)Wo * (C / 4), 256), (unsigned)Ho, (unsigned)(B * Do));
        hipLaunchKernelGGL((resize_trilinear_kernel<4>), grid, dim3(256), 0, st, x, y, B, C, Di, Hi, Wi, Do, Ho, Wo, sd,
                           sh, sw);
    } else {
        const dim3 grid((unsigned)mvsgi::cdiv((long long)Wo * C, 256), (unsigned)Ho, (unsigned)(B * Do));
        hipLaunchKernelGGL((resize_trilinear_kernel<1>), grid, dim3(256), 0, st, x, y, B, C, Di, Hi, Wi, Do, Ho, Wo, sd,
                           sh, sw);
    }
    return mvsgi::check_launch("mvsgi_resize_trilinear_f32");
}
