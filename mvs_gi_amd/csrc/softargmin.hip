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
#ifndef MVSGI_SA_NT
#define MVSGI_SA_NT 1      // norm_costs leaves with nt stores (435 MB per 64 frames nothing on the path reads back): 164 -> 157 us
#endif

#include <cstdlib>

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

// ---------------------------------------------------------------------------------------------
// scale == 2: one workgroup per (frame, low-resolution row pair k | k + 1, column tile).  Output rows 2k + 1 and 2k + 2 blend
// exactly those two rows, so the workgroup stages them for all D candidates in LDS ONCE (16-byte loads; the thread-per-pixel
// kernel above fetched every cost row pair from beyond L2 five times, PMC: 544 MB for a 105 MB input) and each thread
// finishes FOUR consecutive output pixels of one row: the four low-resolution columns c0 - 1 .. c0 + 2 they blend come from
// three LDS reads per row and candidate, and inv_dist / norm_costs leave as 16-byte stores (1 KiB per wave instruction instead
// of 256 B).  The arithmetic of a pixel is the expression of the kernel above, term for term (same taps, same weights: the
// clamped edge taps are staged replicated, where they carry the weights axis2() gives them).
// Logical unit order (b, k, x-tile) walks XCD-contiguously: the unit of row pair k + 1 finds row k + 1 in its XCD's L2.
// DMAX = 16 | 32: the blended candidates of the four pixels live in registers; DMAX = 0: any D, three passes over LDS.
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int sa_xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <int DMAX>
__global__ __launch_bounds__(DMAX == 32 ? 512 : 640) void softargmin_rows_kernel(const float* __restrict__ costs, const float* __restrict__ inv_idx,
                                                               float* __restrict__ inv_dist, float* __restrict__ norm_costs,
                                                               int B, int D, int H, int W, int xt, int xtiles, int units,
                                                               float post_div, int contiguous) {
    extern __shared__ __attribute__((aligned(16))) float sa_lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int u = contiguous ? sa_xcd_remap((int)blockIdx.x, units) : (int)blockIdx.x;
    const int xtile = u % xtiles;
    int tq = u / xtiles;
    const int k = tq % (H + 1) - 1;
    const int b = tq / (H + 1);
    const int x0 = xtile * xt;
    const int r0 = k < 0 ? 0 : k, r1 = k + 1 > H - 1 ? H - 1 : k + 1;
    const int Wp = xt + 4;                       // LDS row: index i holds column x0 + i - 2 (clamped to the image)
    const long long HW = (long long)H * W;
    const float* cb = costs + (long long)b * D * HW;
    const int xv = W - x0 < xt ? W - x0 : xt;    // columns of this tile inside the image

    // ---- stage rows r0, r1 of every candidate: LDS[(row * D + d) * Wp + i] ----
    const int nrow = 2 * D;
    if ((W & 3) == 0 && (xt & 3) == 0) {         // x0, xv multiples of 4: 16-byte loads, two 8-byte LDS stores
        const int nq = xv >> 2;
        for (int e = tid; e < nrow * nq; e += nthr) {
            const int p = e / nq, q = e - p * nq;
            const int row = p >= D, d = p - row * D;
            const f32x4 v = *reinterpret_cast<const f32x4*>(cb + d * HW + (long long)(row ? r1 : r0) * W + x0 + 4 * q);
            float* dst = sa_lds + p * Wp + 4 * q + 2;
            *reinterpret_cast<f32x2*>(dst) = f32x2{v[0], v[1]};
            *reinterpret_cast<f32x2*>(dst + 2) = f32x2{v[2], v[3]};
        }
    } else {
        for (int e = tid; e < nrow * xv; e += nthr) {
            const int p = e / xv, c = e - p * xv;
            const int row = p >= D, d = p - row * D;
            sa_lds[p * Wp + c + 2] = cb[d * HW + (long long)(row ? r1 : r0) * W + x0 + c];
        }
    }
    {   // the halo: column x0 - 1 (index 1) and columns x0 + xv .. x0 + xt (indices xv + 2 .. xt + 2), clamped = replicated
        const int npad = 2 + xt - xv;
        for (int e = tid; e < nrow * npad; e += nthr) {
            const int p = e / npad, h = e - p * npad;
            const int row = p >= D, d = p - row * D;
            const int i = h == 0 ? 1 : xv + 1 + h;
            int c = x0 + i - 2;
            c = c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
            sa_lds[p * Wp + i] = cb[d * HW + (long long)(row ? r1 : r0) * W + c];
        }
    }
    __syncthreads();

    const int half = xt >> 1;                    // threads per output row
    const int rr = tid >= half, j = tid - rr * half;
    const int oy = 2 * k + 1 + rr, c0 = x0 + 2 * j;
    const int OH = 2 * H, OW = 2 * W;
    if (tid >= 2 * half || oy < 0 || oy >= OH || c0 >= W) return;
    const Axis2 ay = axis2(oy, H, 2);
    float l0[4], l1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ox = 2 * c0 + e;
        const Axis2 ax = axis2(ox < OW ? ox : OW - 1, W, 2);
        l0[e] = ax.l0;
        l1[e] = ax.l1;
    }
    const float* la = sa_lds + 2 * j + 1;        // q0 = column c0 - 1
    const float* lb = la + D * Wp;
    // taps of output pixel e: (q0, q1), (q1, q2), (q1, q2), (q2, q3)
    auto blend4 = [&](int d, float (&o)[4]) {
        const float* pa = la + d * Wp;
        const float* pb = lb + d * Wp;
        const float a0 = pa[0], a3 = pa[3], b0 = pb[0], b3 = pb[3];
        const f32x2 am = *reinterpret_cast<const f32x2*>(pa + 1), bm = *reinterpret_cast<const f32x2*>(pb + 1);
        const float at0[4] = {a0, am[0], am[0], am[1]}, at1[4] = {am[0], am[1], am[1], a3};
        const float bt0[4] = {b0, bm[0], bm[0], bm[1]}, bt1[4] = {bm[0], bm[1], bm[1], b3};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = ay.l0 * (l0[e] * at0[e] + l1[e] * at1[e]) + ay.l1 * (l0[e] * bt0[e] + l1[e] * bt1[e]);
    };
    const long long OHW = (long long)OH * OW;
    const long long idx = ((long long)b * OH + oy) * OW + 2 * c0;
    const int nvalid = OW - 2 * c0 < 4 ? OW - 2 * c0 : 4;
    const bool vec = (OW & 3) == 0 && nvalid == 4;
    float* np = norm_costs ? norm_costs + (long long)b * D * OHW + (long long)oy * OW + 2 * c0 : nullptr;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, s[4] = {0.f, 0.f, 0.f, 0.f}, t[4] = {0.f, 0.f, 0.f, 0.f};

    if constexpr (DMAX > 0) {
        float v[DMAX][4];
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            if (d < D) {
                blend4(d, v[d]);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[d][e]);
            }
        }
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            if (d < D) {
                const float w = inv_idx[d];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[d][e] = expf(v[d][e] - m[e]);
                    s[e] += v[d][e];
                    t[e] = fmaf(v[d][e], w, t[e]);
                }
            }
        }
        if (np) {
            float rs[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) rs[e] = 1.0f / s[e];
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                if (d < D) {
                    if (vec) {
#if MVSGI_SA_NT
                        __builtin_nontemporal_store(f32x4{v[d][0] * rs[0], v[d][1] * rs[1], v[d][2] * rs[2], v[d][3] * rs[3]}, reinterpret_cast<f32x4*>(np + d * OHW));
#else
                        *reinterpret_cast<f32x4*>(np + d * OHW) = f32x4{v[d][0] * rs[0], v[d][1] * rs[1], v[d][2] * rs[2], v[d][3] * rs[3]};
#endif
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (e < nvalid) np[d * OHW + e] = v[d][e] * rs[e];
                    }
                }
            }
        }
    } else {
        float o[4];
        for (int d = 0; d < D; ++d) {
            blend4(d, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], o[e]);
        }
        for (int d = 0; d < D; ++d) {
            blend4(d, o);
            const float w = inv_idx[d];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ex = expf(o[e] - m[e]);
                s[e] += ex;
                t[e] = fmaf(ex, w, t[e]);
            }
        }
        if (np) {
            for (int d = 0; d < D; ++d) {
                blend4(d, o);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nvalid) np[d * OHW + e] = expf(o[e] - m[e]) / s[e];
            }
        }
    }
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = t[e] / s[e];
        r[e] = post_div == 1.0f ? r[e] : r[e] / post_div;
    }
    if (vec) {
        *reinterpret_cast<f32x4*>(inv_dist + idx) = f32x4{r[0], r[1], r[2], r[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid) inv_dist[idx + e] = r[e];
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
    if (scale == 2 && !mvsgi::exp_env("MVSGI_SOFTARGMIN_PIXEL")) {
        // row-pair kernel: column tiles of xt low-resolution columns (a multiple of 4), 2 * (xt / 2) threads, LDS 2 * D * (xt + 4) floats
        int xt = (int)(mvsgi::cdiv(W, 4) * 4);
        const int xt_max = (D > 16 && D <= 32) ? 512 : 640;      // = the kernels' launch bounds
        while ((xt > xt_max || (size_t)2 * D * (xt + 4) * 4 > 64 * 1024) && xt > 64) xt = (int)(mvsgi::cdiv(xt / 2, 4) * 4);
        // a frame or two: narrower column tiles until the launch has a workgroup or two per CU (one [16, 80, 320] frame: 81 workgroups
        // of 320 columns -> 324 of 80)
        while (xt > 64 && (long long)B * (H + 1) * mvsgi::cdiv(W, xt) < 2ll * mvsgi::device_cus()) xt = (int)(mvsgi::cdiv(xt / 2, 4) * 4);
        const size_t lds = (size_t)2 * D * (xt + 4) * 4;
        const long long xtiles = mvsgi::cdiv(W, xt), units = (long long)B * (H + 1) * xtiles;
        if (lds <= 160 * 1024 && units < (1ll << 31)) {
            const int threads = (int)(mvsgi::cdiv(xt, 64) * 64);          // 2 rows x xt / 2 pixels quads, whole waves
            auto kern = D <= 16 ? softargmin_rows_kernel<16> : (D <= 32 ? softargmin_rows_kernel<32> : softargmin_rows_kernel<0>);
            static bool attr_set[3][mvsgi::kMaxDevices] = {};
            int dev = 0;
            (void)hipGetDevice(&dev);
            const int ki = D <= 16 ? 0 : (D <= 32 ? 1 : 2);
            if (lds > 64 * 1024 && dev >= 0 && dev < mvsgi::kMaxDevices && !attr_set[ki][dev]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                MVSGI_REQUIRE(e == hipSuccess, "mvsgi_softargmin_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
                attr_set[ki][dev] = true;
            }
            static const int contiguous = mvsgi::exp_env("MVSGI_SOFTARGMIN_RR") ? 0 : 1;
            hipLaunchKernelGGL(kern, dim3((unsigned)units), dim3(threads), lds, mvsgi::as_stream(stream), costs, inv_idx, inv_dist,
                               norm_costs, B, D, H, W, xt, (int)xtiles, (int)units, post_div, contiguous);
            return mvsgi::check_launch("mvsgi_softargmin_f32(rows)");
        }
    }
    MVSGI_REQUIRE(H * scale < 65536 && B < 65536, "mvsgi_softargmin_f32: dimensions exceed the launch geometry");
    hipLaunchKernelGGL(softargmin_kernel, dim3((unsigned)mvsgi::cdiv(W * scale, 256), (unsigned)(H * scale), (unsigned)B), dim3(256), 0,
                       mvsgi::as_stream(stream), costs, inv_idx, inv_dist, norm_costs, B, D, H, W, scale, post_div);
    return mvsgi::check_launch("mvsgi_softargmin_f32");
}
