// K2: 3x3x3 convolution block for gfx950 -- conv3d(pad 1, stride 1|2) * scale + shift
// (+ residual) -> LeakyReLU, channels-last fp32.  Replaces BaseConvBlk3d.forward
// (dsta_mvs/model/common/common_modules.py:107-115) with eval-mode BatchNorm3d applied as
// the per-channel (scale, shift) of the epilogue.
//
// Two implementations behind one entry point:
//
//  * conv3d_mfma_kernel -- implicit GEMM on the exact-fp32 matrix instruction
//    v_mfma_f32_16x16x4_f32.  A workgroup owns a TD x TH x TW brick of output voxels and
//    a block of output channels; per 16-input-channel slice it stages the brick's halo
//    ((T-1)*stride+3 per axis) into LDS once and then runs the 27 taps as shifted LDS reads:
//         D[cout 16][voxel 16] += W[cout 16][k 4] * X[k 4][voxel 16]
//    A (weights) comes straight from HBM/L2 in a pre-packed, lane-ordered layout (one
//    coalesced 16-byte load per lane gives the 4 k-steps of a tap), B (activations) is one
//    ds_read_b128 per lane per tap (its 4 channels feed the same 4 k-steps), and the
//    accumulator tile leaves each lane holding 4 consecutive output channels of one voxel,
//    so the epilogue is a single 16-byte residual load and store per tile.
//    The MFMA result is bit-for-bit a k-ordered fmaf chain, i.e. plain fp32 arithmetic.
//
//  * conv3d_direct_kernel -- one thread per (voxel, group of CO output channels), any channel
//    counts; used for the Cout == 1 cost head (unet_regulator.py:61-68) and as the on-device
//    cross-check of the MFMA path.
#include "common.hpp"
#include "conv_common.hpp"
#ifdef MVSGI_STAMPS
#include <cstdio>
#include <cstdlib>
#endif

namespace {

// head layout (Cout == 1): [1][Cin][27] -> [27][Cin], so a tap's channels are contiguous and a
// wave-uniform 16-byte scalar load fetches four of them
__global__ void pack_head_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 27 * Cin) return;
    const int tap = idx / Cin, ci = idx % Cin;
    wp[idx] = w[ci * 27 + tap];
}

#include "conv3d_f32mfma.hpp"
#include "conv3d_bf16x3.hpp"

// ----------------------------------------------------------------------------------------
// cost head (Cout == 1, stride 1; unet_regulator.py:61-68): one thread per output voxel of a
// 4x8x8 brick; the halo brick goes through LDS exactly as in the MFMA kernel, the 27x16 weights
// of a slice are wave-uniform and arrive as scalar loads (SGPR operands of the FMAs).
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3d_head_kernel(ConvArgs a) {
    constexpr int TD = 4, TH = 8, TW = 8;
    constexpr int ITD = TD + 2, ITH = TH + 2, ITW = TW + 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tw_i = t % a.tiles_w;
    t /= a.tiles_w;
    const int th_i = t % a.tiles_h;
    t /= a.tiles_h;
    const int td_i = t % a.tiles_d;
    const int b = t / a.tiles_d;
    const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
    const int w_ = tid % TW, h_ = (tid / TW) % TH, d_ = tid / (TW * TH);
    const int base = ((d_ * ITH + h_) * ITW + w_) * kVS;
    const float* xb_base = a.x + (long long)b * a.Din * a.Hin * a.Win * a.Cin;
    const f32x4* __restrict__ wq = a.wp;   // [27][Cin/4] float4
    const int cq = a.Cin / 4;
    float acc = 0.f;
    for (int c0 = 0; c0 < a.Cin; c0 += 16) {
        __syncthreads();
        stage_slice<ITD, ITH, ITW>(lds, xb_base, a.Cin, c0, od0 - 1, oh0 - 1, ow0 - 1, a.Din, a.Hin, a.Win, tid);
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const int off = ((kd * ITH + kh) * ITW + kw) * kVS;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(&lds[base + off + q * 4]);
                const f32x4 wv = wq[tap * cq + (c0 >> 2) + q];
                acc = fmaf(xv[0], wv[0], acc);
                acc = fmaf(xv[1], wv[1], acc);
                acc = fmaf(xv[2], wv[2], acc);
                acc = fmaf(xv[3], wv[3], acc);
            }
        }
    }
    const int od = od0 + d_, oh = oh0 + h_, ow = ow0 + w_;
    if (od >= a.Do || oh >= a.Ho || ow >= a.Wo) return;
    const long long vox = (((long long)b * a.Do + od) * a.Ho + oh) * a.Wo + ow;
    float r = acc * a.scale[0] + a.shift[0];
    if (a.res) r += a.res[vox];
    r = r > 0.f ? r : r * a.neg_slope;
    a.y[vox] = r;
}

// ----------------------------------------------------------------------------------------
// direct convolution: thread = (voxel, CO consecutive couts); weights in PyTorch OIDHW order
// ----------------------------------------------------------------------------------------
template <int CO>
__global__ __launch_bounds__(256) void conv3d_direct_kernel(ConvArgs a) {
    const int groups = (a.Cout + CO - 1) / CO;
    const long long total = (long long)a.B * a.Do * a.Ho * a.Wo * groups;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % groups);
    const long long vox = idx / groups;
    const int ow = (int)(vox % a.Wo);
    const int oh = (int)((vox / a.Wo) % a.Ho);
    const int od = (int)((vox / ((long long)a.Wo * a.Ho)) % a.Do);
    const int b = (int)(vox / ((long long)a.Wo * a.Ho * a.Do));
    const int co0 = cg * CO;
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    for (int kd = 0; kd < 3; ++kd) {
        const int id = od * a.stride - 1 + kd;
        if (id < 0 || id >= a.Din) continue;
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * a.stride - 1 + kh;
            if (ih < 0 || ih >= a.Hin) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * a.stride - 1 + kw;
                if (iw < 0 || iw >= a.Win) continue;
                const int tap = (kd * 3 + kh) * 3 + kw;
                const float* xp = a.x + ((((long long)b * a.Din + id) * a.Hin + ih) * a.Win + iw) * a.Cin;
                for (int ci = 0; ci < a.Cin; ++ci) {
                    const float xv = xp[ci];
#pragma unroll
                    for (int o = 0; o < CO; ++o) {
                        const int co = co0 + o;
                        if (co < a.Cout) acc[o] = fmaf(xv, a.w_oidhw[((long long)co * a.Cin + ci) * 27 + tap], acc[o]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        const int co = co0 + o;
        if (co >= a.Cout) break;
        float r = acc[o] * a.scale[co] + a.shift[co];
        if (a.res) r += a.res[vox * a.Cout + co];
        r = r > 0.f ? r : r * a.neg_slope;
        a.y[vox * a.Cout + co] = r;
    }
}

int launch_head(ConvArgs a, hipStream_t st) {
    constexpr size_t lds_bytes = (size_t)6 * 10 * 10 * kVS * sizeof(float);
    a.tiles_d = (int)mvsgi::cdiv(a.Do, 4);
    a.tiles_h = (int)mvsgi::cdiv(a.Ho, 8);
    a.tiles_w = (int)mvsgi::cdiv(a.Wo, 8);
    const long long nt = (long long)a.B * a.tiles_d * a.tiles_h * a.tiles_w;
    MVSGI_REQUIRE(nt < (1ll << 31), "conv3d: too many tiles");
    hipLaunchKernelGGL(conv3d_head_kernel, dim3((unsigned)nt), dim3(256), lds_bytes, st, a);
    return mvsgi::check_launch("mvsgi_conv3d_f32(head)");
}

int launch_direct(const ConvArgs& a, hipStream_t st) {
    const long long vox = (long long)a.B * a.Do * a.Ho * a.Wo;
    if (a.Cout == 1) {
        hipLaunchKernelGGL((conv3d_direct_kernel<1>), dim3((unsigned)mvsgi::cdiv(vox, 256)), dim3(256), 0, st, a);
    } else {
        const long long total = vox * mvsgi::cdiv(a.Cout, 4);
        hipLaunchKernelGGL((conv3d_direct_kernel<4>), dim3((unsigned)mvsgi::cdiv(total, 256)), dim3(256), 0, st, a);
    }
    return mvsgi::check_launch("mvsgi_conv3d_f32(direct)");
}

// One table drives both the launch and the name reported to the bench/profiler, so the kernel
// named in a roofline line is the kernel that ran.  Names are the demangled kernel names as
// rocprofv3 prints them (substring match).
enum Variant {
    V_DIRECT1, V_DIRECT4, V_HEAD,
    V_S1_N16_B256, V_S1_N32_B256, V_S1_N32_B64, V_S1_N64_B128, V_S1_N64_B64, V_S2_N32_B64, V_S2_N64_B64,
    // split-bf16 kernel: 16-wide bricks (conflict-free LDS reads); N = couts per workgroup
    B3_N16, B3_N32, B3_N48, B3_N64, B3_N96, B3_N32_S, B3_N64_S, B3_S2_N32, B3_S2_N64,
    V_COUNT
};
const char* const kVariantNames[V_COUNT] = {
    "conv3d_direct_kernel<1>", "conv3d_direct_kernel<4>", "conv3d_head_kernel",
    "conv3d_mfma_kernel<1, 4, 4, 1, 4, 8, 8, 1>", "conv3d_mfma_kernel<2, 4, 4, 1, 4, 8, 8, 1>",
    "conv3d_mfma_kernel<2, 1, 4, 1, 2, 4, 8, 1>", "conv3d_mfma_kernel<2, 4, 2, 2, 2, 8, 8, 1>",
    "conv3d_mfma_kernel<2, 2, 2, 2, 2, 4, 8, 1>", "conv3d_mfma_kernel<2, 1, 4, 1, 2, 4, 8, 2>",
    "conv3d_mfma_kernel<2, 2, 2, 2, 2, 4, 8, 2>",
    "conv3d_bf16x3_kernel<1, 4, 4, 1, 4, 4, 16, 1>", "conv3d_bf16x3_kernel<2, 4, 4, 1, 4, 4, 16, 1>",
    "conv3d_bf16x3_kernel<3, 4, 4, 1, 4, 4, 16, 1>", "conv3d_bf16x3_kernel<2, 4, 2, 2, 2, 4, 16, 1>",
    "conv3d_bf16x3_kernel<3, 4, 2, 2, 2, 4, 16, 1>", "conv3d_bf16x3_kernel<2, 1, 4, 1, 1, 4, 16, 1>",
    "conv3d_bf16x3_kernel<2, 2, 2, 2, 1, 4, 16, 1>", "conv3d_bf16x3_kernel<2, 1, 4, 1, 2, 4, 8, 2>",
    "conv3d_bf16x3_kernel<2, 2, 2, 2, 2, 4, 8, 2>",
};

// returns V_COUNT when the request cannot be served (error text already set)
int select_variant(const ConvArgs& a, int impl) {
    const bool mfma_ok = (a.Cin % 16 == 0) && (a.Cout % 16 == 0);
    const bool head_ok = (a.Cout == 1) && (a.Cin % 16 == 0) && (a.stride == 1);
    if (impl == MVSGI_CONV_BF16X3 && !mfma_ok) impl = MVSGI_CONV_AUTO;   // head / odd channels: exact paths
    if (impl == MVSGI_CONV_AUTO) impl = (mfma_ok || head_ok) ? MVSGI_CONV_MFMA : MVSGI_CONV_DIRECT;
    if (impl == MVSGI_CONV_DIRECT) {
        if (!a.w_oidhw) { mvsgi::fail("mvsgi_conv3d_f32: direct path needs w_oidhw"); return V_COUNT; }
        return a.Cout == 1 ? V_DIRECT1 : V_DIRECT4;
    }
    if (impl != MVSGI_CONV_MFMA && impl != MVSGI_CONV_BF16X3) {
        mvsgi::fail("mvsgi_conv3d_f32: unknown impl %d", impl);
        return V_COUNT;
    }
    if (!a.wp) { mvsgi::fail("mvsgi_conv3d_f32: tiled path needs w_packed"); return V_COUNT; }
    if (head_ok && impl == MVSGI_CONV_MFMA) return V_HEAD;
    if (!mfma_ok) {
        mvsgi::fail("mvsgi_conv3d_f32: MFMA path needs Cin, Cout multiples of 16 (got %d, %d)", a.Cin, a.Cout);
        return V_COUNT;
    }
    const int CT = a.Cout / 16;
    const long long vox = (long long)a.B * a.Do * a.Ho * a.Wo;
    if (impl == MVSGI_CONV_BF16X3) {
        if (a.stride == 2) return CT <= 3 ? B3_S2_N32 : B3_S2_N64;
        // workgroups a 256-voxel (N <= 48) / 128-voxel (N >= 64) brick decomposition would give
        const long long big = (long long)a.B * mvsgi::cdiv(a.Do, 4) * mvsgi::cdiv(a.Ho, 4) * mvsgi::cdiv(a.Wo, 16);
        const long long mid = (long long)a.B * mvsgi::cdiv(a.Do, 2) * mvsgi::cdiv(a.Ho, 4) * mvsgi::cdiv(a.Wo, 16);
        if (CT == 1) return B3_N16;
        if (CT == 2) return big >= 384 ? B3_N32 : B3_N32_S;
        if (CT == 3) return B3_N48;
        if (CT % 6 == 0 && mid * (CT / 6) >= 384) return B3_N96;
        return mid * mvsgi::cdiv(CT, 4) >= 384 ? B3_N64 : B3_N64_S;
    }
    if (a.stride == 1) {
        if (CT == 1) return V_S1_N16_B256;
        if (CT <= 3) return vox >= 256ll * 512 ? V_S1_N32_B256 : V_S1_N32_B64;
        return vox * CT >= 128ll * 4 * 1024 ? V_S1_N64_B128 : V_S1_N64_B64;
    }
    return CT <= 3 ? V_S2_N32_B64 : V_S2_N64_B64;
}

int launch_variant(int v, const ConvArgs& a, hipStream_t st) {
    switch (v) {
        case V_DIRECT1:
        case V_DIRECT4: return launch_direct(a, st);
        case V_HEAD: return launch_head(a, st);
        case V_S1_N16_B256: return launch_mfma<1, 4, 4, 1, 4, 8, 8, 1>(a, st);
        case V_S1_N32_B256: return launch_mfma<2, 4, 4, 1, 4, 8, 8, 1>(a, st);
        case V_S1_N32_B64: return launch_mfma<2, 1, 4, 1, 2, 4, 8, 1>(a, st);
        case V_S1_N64_B128: return launch_mfma<2, 4, 2, 2, 2, 8, 8, 1>(a, st);
        case V_S1_N64_B64: return launch_mfma<2, 2, 2, 2, 2, 4, 8, 1>(a, st);
        case V_S2_N32_B64: return launch_mfma<2, 1, 4, 1, 2, 4, 8, 2>(a, st);
        case V_S2_N64_B64: return launch_mfma<2, 2, 2, 2, 2, 4, 8, 2>(a, st);
        case B3_N16: return launch_bf16x3<1, 4, 4, 1, 4, 4, 16, 1>(a, st);
        case B3_N32: return launch_bf16x3<2, 4, 4, 1, 4, 4, 16, 1>(a, st);
        case B3_N48: return launch_bf16x3<3, 4, 4, 1, 4, 4, 16, 1>(a, st);
        case B3_N64: return launch_bf16x3<2, 4, 2, 2, 2, 4, 16, 1>(a, st);
        case B3_N96: return launch_bf16x3<3, 4, 2, 2, 2, 4, 16, 1>(a, st);
        case B3_N32_S: return launch_bf16x3<2, 1, 4, 1, 1, 4, 16, 1>(a, st);
        case B3_N64_S: return launch_bf16x3<2, 2, 2, 2, 1, 4, 16, 1>(a, st);
        case B3_S2_N32: return launch_bf16x3<2, 1, 4, 1, 2, 4, 8, 2>(a, st);
        case B3_S2_N64: return launch_bf16x3<2, 2, 2, 2, 2, 4, 8, 2>(a, st);
    }
    return mvsgi::fail("mvsgi_conv3d_f32: bad variant %d", v);
}

int fill_args(ConvArgs& a, const float* x, const float* w_oidhw, const float* w_packed, const float* scale,
              const float* shift, const float* res, float* y, int B, int Cin, int Din, int Hin, int Win, int Cout,
              int stride, float neg_slope) {
    MVSGI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Din > 0 && Hin > 0 && Win > 0, "mvsgi_conv3d_f32: bad dims");
    MVSGI_REQUIRE(stride == 1 || stride == 2, "mvsgi_conv3d_f32: stride %d not in {1, 2}", stride);
    a.x = x;
    a.w_oidhw = w_oidhw;
    a.wp = reinterpret_cast<const f32x4*>(w_packed);
    a.scale = scale;
    a.shift = shift;
    a.res = res;
    a.y = y;
    a.B = B;
    a.Cin = Cin;
    a.Din = Din;
    a.Hin = Hin;
    a.Win = Win;
    a.Cout = Cout;
    a.stride = stride;
    a.neg_slope = neg_slope;
    a.Do = (Din - 1) / stride + 1;   // k=3, pad=1
    a.Ho = (Hin - 1) / stride + 1;
    a.Wo = (Win - 1) / stride + 1;
    return 0;
}

}  // namespace

extern "C" size_t mvsgi_conv3d_packed_weight_floats(int Cout, int Cin) {
    return (size_t)27 * (size_t)Cout * (size_t)Cin;
}

extern "C" size_t mvsgi_conv3d_packed_weight_bytes_bf16x3(int Cout, int Cin) {
    return (size_t)(Cin / 16) * pairs_of(3) * (size_t)(Cout / 16) * 2 * 64 * 16;
}

extern "C" int mvsgi_conv3d_pack_weights_bf16x3(const float* w_oidhw, void* w_packed, int Cout, int Cin,
                                                mvsgi_stream_t stream) {
    MVSGI_REQUIRE(w_oidhw && w_packed, "mvsgi_conv3d_pack_weights_bf16x3: null pointer");
    MVSGI_REQUIRE(Cout > 0 && Cin > 0 && Cout % 16 == 0 && Cin % 16 == 0,
                  "mvsgi_conv3d_pack_weights_bf16x3: Cout=%d Cin=%d must be positive multiples of 16", Cout, Cin);
    const long long total = (long long)(Cin / 16) * pairs_of(3) * (Cout / 16) * 64;
    hipLaunchKernelGGL(pack_weights_bf16x3_kernel, dim3((unsigned)mvsgi::cdiv(total, 256)), dim3(256), 0,
                       mvsgi::as_stream(stream), w_oidhw, reinterpret_cast<bf16x8*>(w_packed), Cout, Cin, 27);
    return mvsgi::check_launch("mvsgi_conv3d_pack_weights_bf16x3");
}

extern "C" int mvsgi_conv3d_pack_weights_f32(const float* w_oidhw, float* w_packed, int Cout, int Cin,
                                             mvsgi_stream_t stream) {
    MVSGI_REQUIRE(w_oidhw && w_packed, "mvsgi_conv3d_pack_weights_f32: null pointer");
    if (Cout == 1 && Cin > 0 && Cin % 16 == 0) {   // cost head: [27][Cin]
        hipLaunchKernelGGL(pack_head_weights_kernel, dim3((unsigned)mvsgi::cdiv(27 * Cin, 256)), dim3(256), 0,
                           mvsgi::as_stream(stream), w_oidhw, w_packed, Cin);
        return mvsgi::check_launch("mvsgi_conv3d_pack_weights_f32");
    }
    MVSGI_REQUIRE(Cout > 0 && Cin > 0 && Cout % 16 == 0 && Cin % 16 == 0,
                  "mvsgi_conv3d_pack_weights_f32: Cout=%d Cin=%d must be positive multiples of 16 (or Cout == 1)",
                  Cout, Cin);
    const long long total = (long long)(Cin / 16) * 27 * (Cout / 16) * 64;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)mvsgi::cdiv(total, 256)), dim3(256), 0,
                       mvsgi::as_stream(stream), w_oidhw, reinterpret_cast<f32x4*>(w_packed), Cout, Cin, 27);
    return mvsgi::check_launch("mvsgi_conv3d_pack_weights_f32");
}

extern "C" int mvsgi_conv3d_f32(const float* x, const float* w_oidhw, const float* w_packed, const float* scale,
                                const float* shift, const float* res, float* y, int B, int Cin, int Din, int Hin,
                                int Win, int Cout, int stride, float neg_slope, int impl, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x && y && scale && shift, "mvsgi_conv3d_f32: null pointer");
    ConvArgs a{};
    if (fill_args(a, x, w_oidhw, w_packed, scale, shift, res, y, B, Cin, Din, Hin, Win, Cout, stride, neg_slope))
        return 1;
    const int v = select_variant(a, impl);
    if (v == V_COUNT) return 1;
    return launch_variant(v, a, mvsgi::as_stream(stream));
}

extern "C" const char* mvsgi_conv3d_variant_f32(int B, int Cin, int Din, int Hin, int Win, int Cout, int stride,
                                                int impl) {
    ConvArgs a{};
    static const float dummy = 0.f;
    if (fill_args(a, &dummy, &dummy, &dummy, &dummy, &dummy, nullptr, nullptr, B, Cin, Din, Hin, Win, Cout, stride,
                  1.f))
        return nullptr;
    const int v = select_variant(a, impl);
    return v == V_COUNT ? nullptr : kVariantNames[v];
}
