// K4: fused bilinear upsample + softmax over D + expectation, fp32.
// Replaces DistanceRegressorWithFixedCandidates.forward
// (dsta_mvs/model/distance_regressor/distance_regressor.py:51-79):
//   c = costs[:, 0]; c = interpolate(c, scale_factor=s, bilinear); p = softmax(c, 1);
//   inv_dist = sum_d p_d * inv_idx_d.
// One thread per output pixel.  For D <= 32 the D blended samples are gathered ONCE into
// registers (4 taps each) and max / exp / sums / probabilities are computed from them (the
// first version walked the candidates three times: 192 instead of 64 loads and 32 instead of
// 16 exps per pixel at D = 16); larger D keeps the multi-pass walk.  The upsampled
// [B, D, sH, sW] volume and the probabilities never touch HBM unless the caller asks for
// norm_costs (training only; inference discards it, spherical_sweep_stereo.py:266).  The 4 source pixels of neighbouring lanes coincide or
// are adjacent, so every candidate plane is read once from HBM and served from L1/L2 after.
#include "common.hpp"

namespace {

struct Axis2 {
    int i0, i1;
    float l0, l1;
};

__device__ __forceinline__ Axis2 axis2(int dst, int in, int scale) {
    Axis2 a;
    if (scale == 1) {
        a.i0 = a.i1 = dst;
        a.l0 = 1.f;
        a.l1 = 0.f;
        return a;
    }
    // F.interpolate(scale_factor=s) uses 1/s as the coordinate scale
    float src = ((float)dst + 0.5f) * (1.0f / (float)scale) - 0.5f;
    src = src < 0.f ? 0.f : src;
    a.i0 = (int)src;
    if (a.i0 > in - 1) a.i0 = in - 1;
    a.i1 = a.i0 + (a.i0 < in - 1 ? 1 : 0);
    a.l1 = src - (float)a.i0;
    a.l0 = 1.0f - a.l1;
    return a;
}

__global__ __launch_bounds__(256) void softargmin_kernel(const float* __restrict__ costs,
                                                         const float* __restrict__ inv_idx,
                                                         float* __restrict__ inv_dist, float* __restrict__ norm_costs,
                                                         int B, int D, int H, int W, int scale, float post_div) {
    // grid = (ceil(OW / 256), OH, B)
    const int OH = H * scale, OW = W * scale;
    const int ox = blockIdx.x * 256 + threadIdx.x;
    if (ox >= OW) return;
    const int oy = blockIdx.y, b = blockIdx.z;
    const long long idx = ((long long)b * OH + oy) * OW + ox;
    const Axis2 ay = axis2(oy, H, scale), ax = axis2(ox, W, scale);
    const long long HW = (long long)H * W;
    const float* cb = costs + (long long)b * D * HW;
    const long long o00 = (long long)ay.i0 * W + ax.i0, o01 = (long long)ay.i0 * W + ax.i1;
    const long long o10 = (long long)ay.i1 * W + ax.i0, o11 = (long long)ay.i1 * W + ax.i1;

    auto sample = [&](int d) {
        const float* p = cb + d * HW;
        return ay.l0 * (ax.l0 * p[o00] + ax.l1 * p[o01]) + ay.l1 * (ax.l0 * p[o10] + ax.l1 * p[o11]);
    };

    if (D <= 32) {
        float v[32];
        float m = -INFINITY;
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            v[d] = d < D ? sample(d) : -INFINITY;
            m = fmaxf(m, v[d]);
        }
        float s = 0.f, t = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            if (d < D) {
                v[d] = expf(v[d] - m);
                s += v[d];
                t = fmaf(v[d], inv_idx[d], t);
            }
        }
        const float r = t / s;
        inv_dist[idx] = post_div == 1.0f ? r : r / post_div;
        if (norm_costs) {
            const long long OHW = (long long)OH * OW;
            float* np = norm_costs + (long long)b * D * OHW + (long long)oy * OW + ox;
            const float rs = 1.0f / s;
#pragma unroll
            for (int d = 0; d < 32; ++d)
                if (d < D) np[d * OHW] = v[d] * rs;
        }
        return;
    }
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, sample(d));
    float s = 0.f, t = 0.f;
    for (int d = 0; d < D; ++d) {
        const float e = expf(sample(d) - m);
        s += e;
        t = fmaf(e, inv_idx[d], t);
    }
    const float r = t / s;
    inv_dist[idx] = post_div == 1.0f ? r : r / post_div;
    if (norm_costs) {
        const long long OHW = (long long)OH * OW;
        float* np = norm_costs + (long long)b * D * OHW + (long long)oy * OW + ox;
        for (int d = 0; d < D; ++d) np[d * OHW] = expf(sample(d) - m) / s;
    }
}

}  // namespace

extern "C" int mvsgi_softargmin_div_f32(const float* costs, const float* inv_idx, float* inv_dist, float* norm_costs,
                                        int B, int D, int H, int W, int scale, float post_div, mvsgi_stream_t stream);

extern "C" int mvsgi_softargmin_f32(const float* costs, const float* inv_idx, float* inv_dist, float* norm_costs,
                                    int B, int D, int H, int W, int scale, mvsgi_stream_t stream) {
    return mvsgi_softargmin_div_f32(costs, inv_idx, inv_dist, norm_costs, B, D, H, W, scale, 1.0f, stream);
}

extern "C" int mvsgi_softargmin_div_f32(const float* costs, const float* inv_idx, float* inv_dist, float* norm_costs,
                                        int B, int D, int H, int W, int scale, float post_div, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(costs && inv_idx && inv_dist, "mvsgi_softargmin_f32: null pointer");
    MVSGI_REQUIRE(post_div != 0.0f, "mvsgi_softargmin_div_f32: post_div must be non-zero");
    MVSGI_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "mvsgi_softargmin_f32: non-positive dimension");
    MVSGI_REQUIRE(scale == 1 || scale == 2, "mvsgi_softargmin_f32: scale %d not in {1, 2}", scale);
    MVSGI_REQUIRE(H * scale < 65536 && B < 65536, "mvsgi_softargmin_f32: dimensions exceed the launch geometry");
    hipLaunchKernelGGL(softargmin_kernel, dim3((unsigned)mvsgi::cdiv(W * scale, 256), (unsigned)(H * scale), (unsigned)B), dim3(256), 0,
                       mvsgi::as_stream(stream), costs, inv_idx, inv_dist, norm_costs, B, D, H, W, scale, post_div);
    return mvsgi::check_launch("mvsgi_softargmin_f32");
}
