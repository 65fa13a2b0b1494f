// Cost head on a SPLIT-PADDED input: out_costs.1 = Conv3d(Cin -> 1, k 3, padding 1, bias) of the regulator
// (dsta_mvs/model/cost_volume_regulator/unet_regulator.py:61-68; BaseConvBlk3d.forward, common/common_modules.py:107-115),
// split-bf16 arithmetic (hi*hi + hi*lo + lo*hi, fp32 accumulate) like every other conv of the default mode.
//
// Same decomposition as conv3d_head_kernel (conv3d.hip): Cout == 1 has no output-channel dimension, so the 27 TAPS play it:
//     P[t][u] = sum_c w[t][c] x[u][c]            a [27 -> 32] x [Cin] x [voxels] product on the matrix cores
//     out[v]  = sum_t P[t][v + offset(t)]        27 shifted copies, summed from LDS
// a workgroup owns an 8 x 32 (h, w) window of one frame and marches along D with three running outputs.  What changes:
//   * the input is the split-padded tensor the polyphase out_costs.0 writes ([B][D+2][H+2][W+2][Cin/16][hi 0-7|hi 8-15|lo 0-7|lo 8-15]):
//     ONE 16-byte load per lane is a whole B operand of v_mfma_f32_16x16x32_bf16 with K = [hi 16 channels | lo 16 channels]
//     (lane (voxel, kg) takes piece kg of the voxel's record) -- no conversion, no bounds arithmetic (zero border; the rest by
//     the buffer descriptor's range check);
//   * two MFMAs per (16-tap tile, 16-voxel tile) give all three product terms:  A1 = [w_hi | w_hi]  ->  x_hi w_hi + x_lo w_hi,
//     A2 = [w_lo | 0]  ->  x_hi w_lo;  2 x 2 x 16 cycles per 16 voxels instead of 8 x 64 cycles per 32 voxels of the exact-fp32
//     v_mfma_f32_32x32x2_f32 head, which ran at 40 % of the fp32-MFMA peak and was bound by it.
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

struct HeadArgs {
    const unsigned char* x;     // split-padded [B][D+2][H+2][W+2][Cin * 4 bytes]
    const bf16x8* wp;           // [Cin/16][2 tap tiles][A1 | A2][64 lanes]
    float* y;                   // fp32 [B][D][H][W]
    float scale, shift;
    float neg_slope;
    int B, Cin, D, H, W;
    int tiles_h, tiles_w;
};

__device__ __forceinline__ int hs_xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// [1][Cin][27] fp32 -> [Cin/16][tt 2][A1 | A2][64 lanes][8 bf16]; lane = (kg << 4) | i, tap = tt * 16 + i (taps >= 27: zero)
//   A1: w_hi[tap][16 cs + 8 (kg & 1) + j]  (kg 0, 1 meet x_hi, kg 2, 3 meet x_lo);  A2: kg < 2: w_lo[tap][16 cs + 8 kg + j], else 0
//   f16: the fp16 split (hi = fp16(w), lo = fp16(w - hi)) for an input written by a kernel of the fp16 split (conv3d_f16.hip)
__global__ void head_split_pack_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cin, bool f16) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = (Cin / 16) * 2 * 64;
    if (idx >= total) return;
    const int lane = idx & 63, tt = (idx >> 6) & 1, cs = idx >> 7;
    const int kg = lane >> 4, tap = tt * 16 + (lane & 15);
    u16x8 a1, a2;
    for (int j = 0; j < 8; ++j) {
        const float v = tap < 27 ? w[(cs * 16 + 8 * (kg & 1) + j) * 27 + tap] : 0.f;
        unsigned short h, l;
        if (f16) {
            const _Float16 hh = (_Float16)v;
            h = __builtin_bit_cast(unsigned short, hh);
            l = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)hh));
        } else {
            const __bf16 hh = (__bf16)v;
            h = __builtin_bit_cast(unsigned short, hh);
            l = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)hh));
        }
        a1[j] = h;
        a2[j] = kg < 2 ? l : (unsigned short)0;
    }
    wp[((cs * 2 + tt) * 2) * 64 + lane] = __builtin_bit_cast(bf16x8, a1);
    wp[((cs * 2 + tt) * 2 + 1) * 64 + lane] = __builtin_bit_cast(bf16x8, a2);
}

template <bool F16>
__global__ __launch_bounds__(256) void conv3d_head_split_kernel(HeadArgs a, int dchunk, int nd) {
    constexpr int TH = 8, TW = 32, ITH = TH + 2, ITW = TW + 2, PV = ITH * ITW;     // 10 x 34 halo window = 340 voxels
    constexpr int NT = (PV + 15) / 16;               // 22 voxel tiles of 16
    constexpr int TPW = (NT + 3) / 4;                // 6 per wave
    constexpr int PSTR = 360;                        // row stride of P (floats): rows 4 apart land 32 banks apart
    static_assert(NT * 16 <= PSTR, "P row too short");
    extern __shared__ __attribute__((aligned(16))) float hs_pbuf[];     // [27][PSTR]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kg = lane >> 4;
    int t = hs_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tw_i = t % a.tiles_w;
    t /= a.tiles_w;
    const int th_i = t % a.tiles_h;
    t /= a.tiles_h;
    const int dc = t % nd;
    const int b = t / nd;
    const int oh0 = th_i * TH, ow0 = tw_i * TW;
    const int od_begin = dc * dchunk;
    const int od_end = od_begin + dchunk < a.D ? od_begin + dchunk : a.D;
    const int p0 = od_begin > 0 ? od_begin - 1 : 0;
    const int p1 = od_end < a.D - 1 ? od_end : a.D - 1;          // inclusive
    const int nchunks = a.Cin / 16;
    const int U = (p1 - p0 + 1) * nchunks;
    const int Hp = a.H + 2, Wp = a.W + 2, rec = a.Cin * 4;
    const long long plane_bytes = (long long)Hp * Wp * rec;

    // this lane's voxel of each of the wave's tiles: halo voxel (ih, iw) is padded voxel (oh0 + ih, ow0 + iw); windows that stick
    // out of the padded plane (ragged H / W) are cut by the row / column test, everything else by the zero border
    unsigned goff[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const int T = wave + 4 * k, v = T * 16 + col;
        const int ih = v / ITW, iw = v - ih * ITW;
        const bool ok = T < NT && v < PV && oh0 + ih < Hp && ow0 + iw < Wp;
        goff[k] = ok ? (unsigned)(((oh0 + ih) * Wp + ow0 + iw) * rec + kg * 16) : 0xffffff00u;
    }
    const unsigned char* xb = a.x + (long long)b * (a.D + 2) * plane_bytes;
    bf16x8 wa[2][2];                                     // [tap tile][A1 | A2] of the current channel slice
    u32x4 xr[TPW];
#define HS_FETCH(P, CS)                                                                                             \
    {                                                                                                               \
        const __amdgpu_buffer_rsrc_t d_ = __builtin_amdgcn_make_buffer_rsrc(                                        \
            const_cast<unsigned char*>(xb) + (long long)((P) + 1) * plane_bytes, 0, (int)plane_bytes, 0x00020000); \
        _Pragma("unroll") for (int k = 0; k < TPW; ++k)                                                             \
            xr[k] = __builtin_amdgcn_raw_buffer_load_b128(d_, goff[k] + (unsigned)((CS) * 64), 0, 0);               \
        _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) {                                                          \
            wa[tt][0] = a.wp[(((CS) * 2 + tt) * 2) * 64 + lane];                                                    \
            wa[tt][1] = a.wp[(((CS) * 2 + tt) * 2 + 1) * 64 + lane];                                                \
        }                                                                                                           \
    }
    HS_FETCH(p0, 0)

    const int h_ = tid / TW, w_ = tid % TW;
    const int oh = oh0 + h_, ow = ow0 + w_;
    const bool inside = oh < a.H && ow < a.W;
    float run[3] = {0.f, 0.f, 0.f};                   // outputs od = p-1, p, p+1 under construction
    f32x4 acc[TPW][2];
#pragma unroll
    for (int k = 0; k < TPW; ++k) acc[k][0] = acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    int p = p0, cs = 0;
    float pend = 0.f;
    long long pend_vox = -1;
    for (int u = 0; u < U; ++u) {
        int ncs = cs + 1, np = p;
        if (ncs == nchunks) { ncs = 0; np = p + 1; }
        bf16x8 cx[TPW], cw[2][2];
#pragma unroll
        for (int k = 0; k < TPW; ++k) cx[k] = __builtin_bit_cast(bf16x8, xr[k]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) { cw[tt][0] = wa[tt][0]; cw[tt][1] = wa[tt][1]; }
        if (pend_vox >= 0) {            // a finished output is stored at the START of the next step, ahead of that step's loads
            a.y[pend_vox] = pend;
            pend_vox = -1;
        }
        {   // the next unit's operands, requested before this unit is multiplied (past the end: this unit again, unused)
            const int fp = u + 1 < U ? np : p, fcs = u + 1 < U ? ncs : cs;
            HS_FETCH(fp, fcs)
        }
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
            if (wave + 4 * k < NT) {                 // wave-uniform
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    if constexpr (F16) {
                        acc[k][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, cw[tt][1]), __builtin_bit_cast(f16x8, cx[k]), acc[k][tt], 0, 0, 0);
                        acc[k][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, cw[tt][0]), __builtin_bit_cast(f16x8, cx[k]), acc[k][tt], 0, 0, 0);
                    } else {
                        acc[k][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cw[tt][1], cx[k], acc[k][tt], 0, 0, 0);
                        acc[k][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cw[tt][0], cx[k], acc[k][tt], 0, 0, 0);
                    }
                }
            }
        }
        if (ncs == 0) {
            // P of plane p: lane (col, kg) holds taps tt * 16 + 4 kg + r of voxel T * 16 + col
#pragma unroll
            for (int k = 0; k < TPW; ++k) {
                if (wave + 4 * k < NT) {
                    const int v = (wave + 4 * k) * 16 + col;
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = tt * 16 + 4 * kg + r;
                            if (row < 27) hs_pbuf[row * PSTR + v] = acc[k][tt][r];
                        }
                    acc[k][0] = acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            __syncthreads();
#pragma unroll
            for (int k2 = 0; k2 < 9; ++k2) {
                const int vv = (h_ + k2 / 3) * ITW + (w_ + k2 % 3);
                run[0] += hs_pbuf[(18 + k2) * PSTR + vv];      // kd = 2 -> od = p - 1
                run[1] += hs_pbuf[(9 + k2) * PSTR + vv];       // kd = 1 -> od = p
                run[2] += hs_pbuf[k2 * PSTR + vv];             // kd = 0 -> od = p + 1
            }
            const int od = p - 1;
            if (od >= od_begin && inside) {
                const long long vox = (((long long)b * a.D + od) * a.H + oh) * a.W + ow;
                const float r = run[0] * a.scale + a.shift;
                pend = r > 0.f ? r : r * a.neg_slope;
                pend_vox = vox;
            }
            run[0] = run[1];
            run[1] = run[2];
            run[2] = 0.f;
            __syncthreads();                         // P consumed before the next plane overwrites it
        }
        p = np;
        cs = ncs;
    }
    if (pend_vox >= 0) a.y[pend_vox] = pend;
    if (p1 < od_end && inside) {                     // the last plane of the volume: no plane behind it
        const long long vox = (((long long)b * a.D + p1) * a.H + oh) * a.W + ow;
        const float r = run[0] * a.scale + a.shift;
        a.y[vox] = r > 0.f ? r : r * a.neg_slope;
    }
#undef HS_FETCH
}

}  // namespace

extern "C" size_t mvsgi_conv3d_head_split_packed_weight_bytes(int Cin) {
    return Cin > 0 && Cin % 16 == 0 ? (size_t)(Cin / 16) * 2 * 2 * 64 * 16 : 0;
}

namespace {
int head_pack(const float* w_oidhw, void* w_packed, int Cin, bool f16, mvsgi_stream_t stream, const char* who) {
    MVSGI_REQUIRE(w_oidhw && w_packed, "%s: null pointer", who);
    MVSGI_REQUIRE(Cin > 0 && Cin % 16 == 0, "%s: Cin=%d must be a positive multiple of 16", who, Cin);
    const int total = (Cin / 16) * 2 * 64;
    hipLaunchKernelGGL(head_split_pack_kernel, dim3((unsigned)mvsgi::cdiv(total, 256)), dim3(256), 0, mvsgi::as_stream(stream), w_oidhw,
                       static_cast<bf16x8*>(w_packed), Cin, f16);
    return mvsgi::check_launch(who);
}

int head_launch(const void* x_split, const void* w_packed, float scale, float shift, float* y, int B, int Cin, int D, int H, int W,
                float neg_slope, bool f16, mvsgi_stream_t stream, const char* who) {
    MVSGI_REQUIRE(x_split && w_packed && y, "%s: null pointer", who);
    MVSGI_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0, "%s: bad dims (Cin %% 16 == 0)", who);
    MVSGI_REQUIRE((long long)(H + 2) * (W + 2) * Cin * 4 < (1ll << 31), "%s: plane too large for 32-bit offsets", who);
    HeadArgs a{};
    a.x = static_cast<const unsigned char*>(x_split);
    a.wp = static_cast<const bf16x8*>(w_packed);
    a.y = y;
    a.scale = scale; a.shift = shift; a.neg_slope = neg_slope;
    a.B = B; a.Cin = Cin; a.D = D; a.H = H; a.W = W;
    a.tiles_h = (int)mvsgi::cdiv(H, 8);
    a.tiles_w = (int)mvsgi::cdiv(W, 32);
    // split D only when the (frame, window) count alone cannot fill the chip (each chunk re-reads its two boundary planes)
    const long long windows = (long long)B * a.tiles_h * a.tiles_w;
    long long nd = windows >= 1024 ? 1 : mvsgi::cdiv(1024, windows);
    if (nd > mvsgi::cdiv(D, 2)) nd = mvsgi::cdiv(D, 2);
    const int dchunk = (int)mvsgi::cdiv(D, nd);
    nd = mvsgi::cdiv(D, dchunk);
    const long long nt = windows * nd;
    MVSGI_REQUIRE(nt < (1ll << 31), "%s: too many tiles", who);
    constexpr size_t lds_bytes = (size_t)27 * 360 * sizeof(float);
    if (f16)
        hipLaunchKernelGGL(conv3d_head_split_kernel<true>, dim3((unsigned)nt), dim3(256), lds_bytes, mvsgi::as_stream(stream), a, dchunk, (int)nd);
    else
        hipLaunchKernelGGL(conv3d_head_split_kernel<false>, dim3((unsigned)nt), dim3(256), lds_bytes, mvsgi::as_stream(stream), a, dchunk, (int)nd);
    return mvsgi::check_launch(who);
}
}  // namespace

// w_oidhw: [1][Cin][3][3][3] fp32 (device) -> the head's fragment layout (device)
extern "C" int mvsgi_conv3d_head_split_pack_weights(const float* w_oidhw, void* w_packed, int Cin, mvsgi_stream_t stream) {
    return head_pack(w_oidhw, w_packed, Cin, false, stream, "mvsgi_conv3d_head_split_pack_weights");
}
// the same in the fp16 split (for an input written by a kernel of the fp16 split; the caller pre-scales the weights by a power of
// two and passes its inverse in `scale`, see MVSGI_CONV_F16)
extern "C" int mvsgi_conv3d_head_split_pack_weights_f16(const float* w_oidhw, void* w_packed, int Cin, mvsgi_stream_t stream) {
    return head_pack(w_oidhw, w_packed, Cin, true, stream, "mvsgi_conv3d_head_split_pack_weights_f16");
}

// y[b][d][h][w] = act(conv(x)[0] * scale + shift): x split-padded [B][D+2][H+2][W+2][Cin] (e.g. the output of
// mvsgi_conv3d_up2_poly_split), scale / shift: one float each in DEVICE memory is not needed -- they are passed by value
// (BaseConvBlk3d with NoOp norm: scale 1, shift = bias); neg_slope 1 = no activation (out_costs.1 has none).
extern "C" int mvsgi_conv3d_head_split(const void* x_split, const void* w_packed, float scale, float shift, float* y, int B, int Cin,
                                       int D, int H, int W, float neg_slope, mvsgi_stream_t stream) {
    return head_launch(x_split, w_packed, scale, shift, y, B, Cin, D, H, W, neg_slope, false, stream, "mvsgi_conv3d_head_split");
}
extern "C" int mvsgi_conv3d_head_split_f16(const void* x_split, const void* w_packed, float scale, float shift, float* y, int B, int Cin,
                                           int D, int H, int W, float neg_slope, mvsgi_stream_t stream) {
    return head_launch(x_split, w_packed, scale, shift, y, B, Cin, D, H, W, neg_slope, true, stream, "mvsgi_conv3d_head_split_f16");
}
