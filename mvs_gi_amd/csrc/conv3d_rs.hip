// K2e: "register-stationary" split-bf16 3x3x3 convolution for the narrow layers of the regulator
// (Cin, Cout <= 32: BaseConvBlk3d.forward, dsta_mvs/model/common/common_modules.py:107-115, as used by
// ResConvBlk3d :231-244 at UNet level 0), and the activation format it works on.
//
// Why a second schedule.  conv3d_bf16x3_kernel streams the pre-split weight fragments from L2 once per
// (brick, slice) and wave -- ~42 B/clk/CU of vector-memory traffic that shares the CU's memory queue with the
// staging loads -- and gives half of the workgroup's waves to splitting fp32 activations into hi|lo bf16 for
// LDS.  For the 32 -> 32 layers ALL weights are 110 KB = 432 registers x 64 lanes, which fits the register
// files of two waves (gfx950: 512 registers per lane at one wave per SIMD).  So here
//   * a workgroup is 4 waves, one per SIMD, each holding its half of the layer's weights in registers for the
//     whole launch (wave (pl, s): the 16-channel slice s, both cout tiles, all 14 tap pairs = 224 registers);
//     the main loop has no weight traffic at all;
//   * activations live in HBM ALREADY SPLIT ("split-padded" format below): the split is done once, in the
//     producing layer's epilogue, instead of once per staged halo voxel; staging is a pure copy and runs on the
//     LDS-DMA path (buffer_load ... lds: no VGPR round trip, no VALU, no producer waves);
//   * tensors carry a one-voxel zero border, so a halo brick is a fixed pattern of offsets from the brick's
//     origin: no bounds arithmetic per staged voxel (the per-lane DMA offsets are computed once per launch);
//   * the two slice-waves of a voxel set exchange half of their accumulators through LDS (32 KB per brick of
//     128 voxels against 860 KB of fragment reads) and each finishes half of the tiles: scale / shift,
//     residual, LeakyReLU, split, 16-byte stores;
//   * one s_barrier per brick, placed between tap pairs 12 and 13: the last pair's MFMAs cover the first LDS
//     reads of the next brick, the epilogue of brick u runs under the MFMAs of brick u + 1.
//
// Split-padded activation format:  [B][D+2][H+2][W+2][C/16][4][8] bf16
//   voxel record = C/16 slices x 64 B, slice = [hi(c 0-7) | hi(c 8-15) | lo(c 0-7) | lo(c 8-15)],
//   x = hi + lo, hi = bf16(x) (RNE), lo = bf16(x - hi): 16-17 significant bits, the same 4 B per element as fp32;
//   the border voxels are zero and never written (the convolution's padding).
#include "common.hpp"
#ifndef MVSGI_RS_NT0
#define MVSGI_RS_NT0 0       // cache policy of the 32 -> 32 layers' output stores (2 = nt: measured slower, the next layer reads them)
#endif
#ifndef MVSGI_RS_NT3
#define MVSGI_RS_NT3 0       // cache policy of the polyphase layer's output stores (2 = nt: measured slower, 1742 -> 1772 us)
#endif
#ifndef MVSGI_RS16_NT
#define MVSGI_RS16_NT 2      // post_vol's output stores: nt (276 -> 272 us, and the stride-2 kernel behind it 491 -> 484)
#endif

#include <cstring>
#ifdef MVSGI_RS_STAMPS
#include <cstdio>
#include <cstdlib>
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#include "split_fmt.hpp"

// ---------------------------------------------------------------------------------------------
// format conversion (module boundaries, tests): fp32 NDHWC <-> split-padded.  One thread per (voxel, 8 channels).
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ void f32_to_split_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, int B, int C, int D, int H,
                                    int W, unsigned* __restrict__ sat) {
    const long long n = (long long)B * D * H * W * (C / 8);
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < n;
    if (!live) idx = n - 1;          // whole waves reach the range report (the store below is masked)
    const int g = (int)(idx % (C / 8));          // 8-channel group
    long long v = idx / (C / 8);
    const int w = (int)(v % W);
    v /= W;
    const int h = (int)(v % H);
    v /= H;
    const int d = (int)(v % D);
    const int b = (int)(v / D);
    const float* src = x + ((((long long)b * D + d) * H + h) * W + w) * C + g * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
    u32x2 h0, l0, h1, l1;
    float satm = 0.f;
    sf_split4<F16>(a0, h0, l0, satm);
    sf_split4<F16>(a1, h1, l1, satm);
    unsigned char* dst = y + ((((long long)b * (D + 2) + d + 1) * (H + 2) + h + 1) * (W + 2) + w + 1) * (C * 4) +
                         (g >> 1) * 64 + (g & 1) * 16;
    if (live) {
        *reinterpret_cast<u32x4*>(dst) = u32x4{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<u32x4*>(dst + 32) = u32x4{l0[0], l0[1], l1[0], l1[1]};
    }
    if constexpr (F16) sf_sat_report(sat, kSatSplit, satm, kF16Max);
}

template <bool F16>
__global__ void split_to_f32_kernel(const unsigned char* __restrict__ x, float* __restrict__ y, int B, int C, int D, int H,
                                    int W) {
    const long long n = (long long)B * D * H * W * (C / 8);
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int g = (int)(idx % (C / 8));
    long long v = idx / (C / 8);
    const int w = (int)(v % W);
    v /= W;
    const int h = (int)(v % H);
    v /= H;
    const int d = (int)(v % D);
    const int b = (int)(v / D);
    const unsigned char* src = x + ((((long long)b * (D + 2) + d + 1) * (H + 2) + h + 1) * (W + 2) + w + 1) * (C * 4) +
                               (g >> 1) * 64 + (g & 1) * 16;
    const u32x4 hi = *reinterpret_cast<const u32x4*>(src), lo = *reinterpret_cast<const u32x4*>(src + 32);
    float* dst = y + ((((long long)b * D + d) * H + h) * W + w) * C + g * 8;
    *reinterpret_cast<f32x4*>(dst) = sf_join4<F16>(u32x2{hi[0], hi[1]}, u32x2{lo[0], lo[1]});
    *reinterpret_cast<f32x4*>(dst + 4) = sf_join4<F16>(u32x2{hi[2], hi[3]}, u32x2{lo[2], lo[3]});
}

// ---------------------------------------------------------------------------------------------
// the 32 -> 32 kernel
// ---------------------------------------------------------------------------------------------
struct RsArgs {
    const unsigned char* x;    // split-padded [B][D+2][H+2][W+2][32]
    unsigned char* y;          // split-padded, same geometry (stride 1)
    const unsigned char* res;  // split-padded residual or nullptr
    const bf16x8* wp;          // mvsgi_conv3d_rs_pack_weights layout: [slice][cout tile][14 pairs][hi|lo][64 lanes]
    const float* scale;
    const float* shift;
    int B, D, H, W;
    float neg_slope;           // in [0, 1]: act(v) = max(v, v * neg_slope)
    int tiles_d, tiles_h, tiles_w, total_units;
    unsigned long long* dbg;   // MVSGI_RS_STAMPS diagnostic build only
    // polyphase mode (MODE 2 below): `wp` holds 16 weight sets [pd 2][d-class 4][ph 2] of `wp_set` bf16x8 elements each;
    // walkers per (XCD, role)
    long long wp_set;
    int walkers;
    unsigned* sat;             // the range report's words (csrc/api.cpp): written when a clamp of the fp16 split engaged
};

__device__ __forceinline__ int rs_xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

namespace rs {
constexpr int TD = 2, TH = 4, TW = 16;        // brick: 128 output voxels = 2 planes x 4 rows x 16
constexpr int ITH = TH + 2, ITW = TW + 2;     // halo: 4 planes x 6 rows x 18
// LDS image of a halo brick: voxel v = d * PL + h * ROW + w in a HI region (64 B per voxel: the hi halves of the two
// 16-channel slices, 4 chunks of 16 B) followed by a LO region.  Chunk c of voxel (d, h, w) sits in slot c ^ ((w >> 2) & 3):
// the 16 voxels a ds_read_b128 lane group touches -- 8 lanes on one tap, 8 on its pair partner -- then cover the 16
// 16-byte units of a 256-B bank row exactly once when (a) both taps are in one row and differ by one voxel (in-row pairs
// kw = 0 | 1) or (b) both have kw = 2 and their rows start at the same voxel index mod 4 (ROW, PL multiples of 4).
// Because the slot depends on w only, a tap's row / plane offset and a tile's row are plain immediates of the read.
constexpr int ROW = 20, PL = ITH * ROW;       // 2 padding voxels per row
constexpr int NV = 4 * PL;                    // 480 voxels = 30 DMA pieces per region
constexpr int REGION = NV * 64;               // 30,720
constexpr int IMG = 2 * REGION;               // 61,440
constexpr int BUF1 = 65536;                   // the second image sits one address bit away
constexpr int SCR = 2 * BUF1;                 // exchange scratch: 2 x (4 waves x 4 KB), the two halves one address bit (16384) apart
constexpr int LDS_BYTES = SCR + 2 * 16384;    // 163,840 = all of the CU's LDS
static_assert((SCR & 16384) == 0 && BUF1 + IMG <= SCR, "scratch halves toggle by XOR 16384");
constexpr int NDMA = IMG / 1024;              // 60 pieces per image
constexpr int DPW = NDMA / 4;                 // 15 per wave
constexpr int kPairs = 14;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(ROW % 4 == 0 && PL % 4 == 0 && NV % 16 == 0, "image geometry");
// pair p -> its two taps as (k = kd * 3 + kh, kw); second tap of pair 13: none (-1)
__host__ __device__ constexpr int pair_k(int p, int which) { return p < 9 ? p : (2 * (p - 9) + which < 9 ? 2 * (p - 9) + which : -1); }
__host__ __device__ constexpr int pair_kw(int p, int which) { return p < 9 ? which : 2; }
}  // namespace rs

// [Cout 32][Cin 32][27] -> [slice 2][cout tile 2][14 pairs][hi|lo][64 lanes][8 bf16]
//   lane = (kg << 4) | i holds W[cout = ct*16 + i][cin = slice*16 + (kg>>1)*8 + j][tap of pair p selected by kg & 1]
__global__ void rs_pack_weights_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, bool f16) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 2 * 2 * rs::kPairs * 64) return;
    const int lane = idx & 63;
    int r = idx >> 6;
    const int p = r % rs::kPairs;
    r /= rs::kPairs;
    const int ct = r & 1, sl = r >> 1;
    const int kg = lane >> 4, co = ct * 16 + (lane & 15), ci = sl * 16 + (kg >> 1) * 8;
    const int k = rs::pair_k(p, kg & 1), kw = rs::pair_kw(p, kg & 1);
    u16x8 hi, lo;
    for (int j = 0; j < 8; ++j) {
        const float v = k >= 0 ? w[((long long)co * 32 + ci + j) * 27 + k * 3 + kw] : 0.f;
        unsigned short h_, l_;
        sf_split_weight(v, f16, h_, l_);
        hi[j] = h_;
        lo[j] = l_;
    }
    const int o = (((sl * 2 + ct) * rs::kPairs + p) * 2) * 64 + lane;
    wp[o] = __builtin_bit_cast(bf16x8, hi);
    wp[o + 64] = __builtin_bit_cast(bf16x8, lo);
}

struct RsUnit {      // coordinates of a brick (wave-uniform)
    int b, od, oh, ow;
};

// MODE 0: split-padded output (the next register-stationary layer's input).
// MODE 1: the output goes to a plain fp32 channels-last tensor [B][D][H][W][32] (the hand-over to a kernel that stages fp32).
// MODE 2: POLYPHASE form of ResizeConv3d (common_modules.py:332-355; 32 -> 16 channels: out_costs.0 of the (16, 32) regulator,
//   unet_regulator.py:52-60).  conv(trilinear_x2(x)) is, per output phase (pd, ph, pw) in {0,1}^3, an ordinary 3x3x3 convolution
//   over the LOW-resolution tensor with weights (M_d x M_h x M_w) w folded at lowering time (dropin/polyphase.py), i.e. one
//   32 -> 128 convolution + a depth-to-space shuffle, and a 32 -> 32 slice of it -- the two pw phases of one (pd, ph) -- is exactly
//   this kernel's shape: no upsampled tensor, no blend arithmetic, the same register-stationary weights, LDS-DMA staging and
//   schedule as the level-0 residual convs.  A workgroup has a ROLE (pd, ph, od): it walks the bricks (b, oh, ow) of ONE
//   d-brick od for one (pd, ph) (all roles of an XCD walk the same (b, oh, ow) range: the bricks they share come from that
//   XCD's L2), because the folded weights of a wave's plane depend on whether the plane is the first / an interior / the last
//   low-resolution plane (the upsample clamps, the conv zero-pads the UPSAMPLED grid): a wave keeps one weight set for the
//   whole launch.  The output cell (i_d, i_h, i_w), cout tile pw, goes to hi-res voxel (2 i_d + pd, 2 i_h + ph, 2 i_w + pw) of a
//   plain fp32 [B][2D][2H][2W][16] tensor = the MODE 1 addressing with other strides.  Cells on the H / W faces of the volume
//   need other centre taps along that axis: their difference arrives as a raw (pre-scale) correction that
//   csrc/conv3d_up2face.hip has written into the output voxels beforehand; the epilogue reads it back (masked to face cells)
//   where the other modes read the residual.
// F16: activations, weights and the residual in the fp16 split (csrc/split_fmt.hpp) -- the same schedule, v_mfma_f32_16x16x32_f16
template <int MODE, bool F16 = false>
__global__ __launch_bounds__(256, 1) void conv3d_rs32_kernel(RsArgs a) {
    using namespace rs;
    constexpr bool OUTF32 = MODE == 1 || MODE == 2;
    constexpr bool UP2 = MODE >= 2;                    // MODE 3: polyphase with a SPLIT-PADDED hi-res output [B][2D+2][2H+2][2W+2][64 B]
    constexpr int OB = MODE == 3 ? 1 : 0;              // border of the polyphase output tensor
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = wave & 1, s = wave >> 1;            // plane of the brick, channel slice
    const int col = lane & 15, kg = lane >> 4;
    const bool second = kg & 1;
    const int half = kg >> 1;
    const int Hp = a.H + 2, Wp = a.W + 2;
    const long long frame_bytes = (long long)(a.D + 2) * Hp * Wp * 128;
    const long long total_bytes = frame_bytes * a.B;
    // ---- polyphase role of this workgroup: blockIdx = (walker * R + role) * 8 + xcd ----
    const int Hh = 2 * a.H + 2 * OB, Wh = 2 * a.W + 2 * OB;       // rows / columns of the polyphase output tensor (border included)
    const int up_R = 4 * a.tiles_d;
    const int up_q = (int)blockIdx.x >> 3;
    const int up_role = UP2 ? up_q % up_R : 0, up_walker = UP2 ? up_q / up_R : 0;
    const int up_pd = (up_role >> 1) & 1, up_ph = up_role & 1, up_od = up_role >> 2;
    const long long oframe_bytes = UP2 ? (long long)(2 * a.D + 2 * OB) * Hh * Wh * 64 : (long long)a.D * a.H * a.W * 128;
    if constexpr (UP2) {
        const int i_d = up_od * TD + pl;
        const int cls = a.D == 1 ? 3 : (i_d == 0 ? 0 : (i_d == a.D - 1 ? 2 : 1));
        a.wp += (long long)((up_pd * 4 + cls) * 2 + up_ph) * a.wp_set;
    }

#ifdef MVSGI_RS_STAMPS
    unsigned long long t_entry_;      // kernel entry: stamp 0 of every wave (tools/rs_stamps.py)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry_)::"memory");
#endif
#ifndef MVSGI_RS_WEIGHTS_FIRST
#define MVSGI_RS_WEIGHTS_FIRST 1      // 0: the weights behind the plans and the first image's requests (an experiment, round 6: slower)
#endif
#define RS_PIN_WEIGHTS()                                                                             \
    _Pragma("unroll") for (int p = 0; p < kPairs; ++p)                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
            const bf16x8* q = a.wp + ((long long)((s * 2 + j) * kPairs + p) * 2) * 64 + lane;        \
            wh[p][j] = q[0];                                                                         \
            wl[p][j] = q[64];                                                                        \
        }                                                                                            \
    _Pragma("unroll") for (int p = 0; p < kPairs; ++p)                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) asm volatile("" : "+a"(wh[p][j]), "+a"(wl[p][j])); \
    __builtin_amdgcn_sched_barrier(0);
    // ---- weights: resident in the accumulator half of the register file for the whole launch.  At one frame a workgroup lives for
    // one or two bricks and this prologue is a fifth of it (in-kernel stamps of [8,40,160] x 1 frame, tools/rs_stamps.py 1 8 40 160,
    // cycles from kernel entry: weights resident 5.0-7.8 k, first image landed 8.8 k, phase 0 from 10.2 k, two phases of 6.9 k,
    // two drain phases of 3.5 k, exit 33.9 k).  Requesting them BEHIND the first image's LDS-DMA (-DMVSGI_RS_WEIGHTS_FIRST=0) is
    // slower: the plans (3.4 k cycles of index arithmetic) then run before any request is out instead of under the weights'
    // 56 KiB per wave -- phase 0 from 11.9 k, exit 35.5 k ----
    bf16x8 wh[kPairs][2], wl[kPairs][2];
    if constexpr (MVSGI_RS_WEIGHTS_FIRST) { RS_PIN_WEIGHTS() }

    // ---- fragment read bases (image 0, HI region); group 0 = own tiles (rows 2s, 2s+1), group 1 = the partner's ----
    const int chunk = 2 * s + half;
    int rbin[2], rb2[2][5];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int row0 = g == 0 ? 2 * s : 2 - 2 * s;
        const int w_in = col + (second ? 1 : 0), w_2 = col + 2;
        rbin[g] = (pl * PL + row0 * ROW + w_in) * 64 + ((chunk ^ ((w_in >> 2) & 3)) << 4);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int kA = 2 * q, kB = 2 * q + 1 < 9 ? 2 * q + 1 : 2 * q;
            const int k = second ? kB : kA;
            rb2[g][q] = (pl * PL + row0 * ROW + w_2 + (k / 3) * PL + (k % 3) * ROW) * 64 + ((chunk ^ ((w_2 >> 2) & 3)) << 4);
        }
    }
    // ---- DMA plan: piece i = wave + 4 m fills LDS bytes [i * 1024, +1024) of an image: 16 voxels x 4 chunks ----
    unsigned voff[DPW];
#pragma unroll
    for (int m = 0; m < DPW; ++m) {
        const int i = wave + 4 * m;
        const int region = i >= NDMA / 2 ? 1 : 0, jj = i - region * (NDMA / 2);
        const int v = 16 * jj + (lane >> 2), slot = lane & 3;
        const int d = v / PL, r = v - d * PL;
        const int h = r / ROW, w = r - h * ROW;
        const int c = slot ^ ((w >> 2) & 3);
        voff[m] = w < ITW ? (unsigned)(((d * Hp + h) * Wp + w) * 128 + (c >> 1) * 64 + region * 32 + (c & 1) * 16)
                          : 0xffffff00u;     // row padding: beyond num_records, zero-filled
    }
    // ---- output side: this lane's 16-byte piece of the own tiles' voxel records ----
    // (polyphase mode: the two cout tiles are the two pw phases of the SAME 16 output channels)
    const f32x4 esc[2] = {*reinterpret_cast<const f32x4*>(a.scale + kg * 4), *reinterpret_cast<const f32x4*>(a.scale + (UP2 ? 0 : 16) + kg * 4)};
    const f32x4 esh[2] = {*reinterpret_cast<const f32x4*>(a.shift + kg * 4), *reinterpret_cast<const f32x4*>(a.shift + (UP2 ? 0 : 16) + kg * 4)};
    unsigned voy0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
        voy0[i] = (unsigned)((((pl + 1) * Hp + 2 * s + i + 1) * Wp + col + 1) * 128 + (kg & 1) * 32 + (kg >> 1) * 16);
    unsigned voyo[2];     // the same for the OUTPUT tensor (fp32: unpadded, couts 4 kg .. 4 kg + 3 of the tile are 16 contiguous bytes)
#pragma unroll
    for (int i = 0; i < 2; ++i)
        voyo[i] = UP2 ? (unsigned)(((2 * pl * Hh + 2 * (2 * s + i)) * Wh) * 64 + col * 128 +               // cell -> hi-res voxel pair
                                   (OUTF32 ? kg * 16 : (kg & 1) * 32 + (kg >> 1) * 16))
                      : (OUTF32 ? (unsigned)(((pl * a.H + 2 * s + i) * a.W + col) * 128 + kg * 16) : voy0[i]);
    unsigned vocb[2];     // polyphase: where this lane READS its face correction (fp32 couts 4 kg .. 4 kg + 3 of the voxel's 64-byte record)
#pragma unroll
    for (int i = 0; i < 2; ++i) vocb[i] = (unsigned)(((2 * pl * Hh + 2 * (2 * s + i)) * Wh) * 64 + col * 128 + kg * 16);
    int sp_rd = SCR + (wave ^ 2) * 4096 + lane * 16, sp_wr = SCR + 16384 + wave * 4096 + lane * 16;

    int total = a.total_units;
    const int G = gridDim.x;
    int n, id0, step;
    if constexpr (UP2) {
        // per role the space (b, oh, ow) of tiles_h * tiles_w * B bricks; XCD x owns a contiguous eighth of it, shared by the
        // roles; walker j of `walkers` takes every walkers-th brick of that range
        total = a.B * a.tiles_h * a.tiles_w;
        const int x = (int)blockIdx.x & 7, q8 = total >> 3, r8 = total & 7;
        const int lo = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8, cnt = q8 + (x < r8 ? 1 : 0);
        n = cnt > up_walker ? (cnt - up_walker + a.walkers - 1) / a.walkers : 0;
        id0 = lo + up_walker;
        step = a.walkers;
        if (n == 0) return;        // the whole workgroup (uniform): nothing was started yet
    } else {
        n = (total - (int)blockIdx.x + G - 1) / G;       // bricks of this workgroup (the same for its 4 waves)
        // The walk: logical ids id0 + k * step (XCD-contiguous remap, cdna_hip_programming.md T1; G % 8 == 0 or G == total)
        id0 = rs_xcd_remap((int)blockIdx.x, total);
        step = G == total ? 0 : G >> 3;
    }
    // logical order (b, oh, od, ow), ow fastest: the bricks stacked along D share two of their four / six input planes and
    // follow each other within one XCD round (tiles_w ids apart), the bricks above / below a round or two later -- in the
    // (b, od, oh, ow) order the D neighbours were tiles_h * tiles_w ids (several L2 capacities) apart
    // (polyphase mode: od is the role's, the walk's digits are (b, oh, ow): a d-radix of 1 makes the same code do that)
    const int walk_td = UP2 ? 1 : a.tiles_d;
    const int sw = step % a.tiles_w;
    int tq = step / a.tiles_w;
    const int sd = tq % walk_td;
    tq /= walk_td;
    const int sh_ = tq % a.tiles_h, sb = tq / a.tiles_h;
    RsUnit c2{0, 0, 0, 0}, c1{0, 0, 0, 0}, c0, nx;
    {
        int t_ = id0;
        c0.ow = t_ % a.tiles_w;
        t_ /= a.tiles_w;
        c0.od = t_ % walk_td;
        t_ /= walk_td;
        c0.oh = t_ % a.tiles_h;
        c0.b = t_ / a.tiles_h;
    }
// tile coordinates of the next brick of the walk: mixed-radix add, no division (od counts within the WALK: 0 in polyphase mode)
#define RS_STEP(DST, SRC)                                                      \
    {                                                                          \
        int w_ = SRC.ow + sw, c_ = w_ >= a.tiles_w;                            \
        DST.ow = w_ - (c_ ? a.tiles_w : 0);                                    \
        int d_ = SRC.od + sd + c_;                                             \
        c_ = d_ >= walk_td;                                                    \
        DST.od = d_ - (c_ ? walk_td : 0);                                      \
        int h_ = SRC.oh + sh_ + c_;                                            \
        c_ = h_ >= a.tiles_h;                                                  \
        DST.oh = h_ - (c_ ? a.tiles_h : 0);                                    \
        DST.b = SRC.b + sb + c_;                                               \
    }
    RS_STEP(nx, c0)
// buffer descriptor whose base is the brick's origin inside PTR (a split-padded tensor of this geometry); VALID = false
// gives zero records: every access through it is dropped (loads return 0).  Straight-line scalar code on purpose: a
// branch inside the phase body splits it into basic blocks, and hipcc then permutes the accumulators between registers at
// the block boundaries (v_accvgpr_mov right in front of an asm MFMA it cannot see: an unpadded hazard, wrong sums).
#define RS_DESC(PTR, U, VALID)                                                                                   \
    ({                                                                                                           \
        const long long off_ = (long long)(U).b * frame_bytes +                                                  \
                               ((long long)(((U).od + up_od) * TD * Hp + (U).oh * TH) * Wp + (U).ow * TW) * 128; \
        const long long left_ = total_bytes - off_;                                                              \
        const int rec_ = left_ > 0x7fffff00ll ? 0x7fffff00 : (int)left_;                                         \
        const int ok_ = (int)(VALID) & (int)(left_ > 0);                                                         \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(PTR) + off_, 0, ok_ ? rec_ : 0, 0x00020000); \
    })
#define RS_DESC_OUT(U, VALID)                                                                                   \
    ({                                                                                                           \
        const long long off_ = (long long)(U).b * oframe_bytes +                                                 \
            (UP2 ? ((long long)((2 * up_od * TD + up_pd + OB) * Hh + 2 * (U).oh * TH + up_ph + OB) * Wh + 2 * (U).ow * TW + OB) * 64 \
                 : ((long long)((U).od * TD * a.H + (U).oh * TH) * a.W + (U).ow * TW) * 128);                    \
        const long long left_ = (long long)a.B * oframe_bytes - off_;                                            \
        const int rec_ = left_ > 0x7fffff00ll ? 0x7fffff00 : (int)left_;                                         \
        const int ok_ = (int)(VALID) & (int)(left_ > 0);                                                         \
        __builtin_amdgcn_make_buffer_rsrc(a.y + off_, 0, ok_ ? rec_ : 0, 0x00020000);                            \
    })
#define RS_F_SPL(...) if constexpr (!OUTF32) { __VA_ARGS__ }
#define RS_F_F32(...) if constexpr (OUTF32) { __VA_ARGS__ }
#define RS_F_UP2(...) if constexpr (UP2) { __VA_ARGS__ }
#define RS_F_NUP2(...) if constexpr (!UP2) { __VA_ARGS__ }
#define RS_DSC_Y(U, VALID) ((OUTF32 || UP2) ? RS_DESC_OUT(U, VALID) : RS_DESC(a.y, U, VALID))
// the "residual" descriptor: the residual tensor, or (polyphase mode) the OUTPUT tensor, whose face voxels hold the corrections
#define RS_DSC_R(U, VALID) (UP2 ? RS_DESC_OUT(U, VALID) : RS_DESC(a.res, U, (int)(a.res != nullptr) & (int)(VALID)))
#define RS_RES_OFF(I) (UP2 ? voc[I] : voy0[I])
#define RS_DMA(M)                                                                                                \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(dsc_x, (__attribute__((address_space(3))) void*)(lds + nxt_img + (wave + 4 * (M)) * 1024), \
                                             16, voff[M], 0, 0, 0);
// The MFMAs are inline asm so that the weight operands can be pinned to the accumulator half of the register file
// ("a"): hipcc keeps A/B operands of the builtin form in VGPRs and would shuttle the 224 weight registers through
// v_accvgpr_read every brick.  hipcc pads no hazards around asm: the schedule keeps >= 3 MFMAs between an accumulator's
// last MFMA and its first read, and nothing else reads or writes MFMA operands.
#define RS_MF(ACC, WREG, XREG)                                                                                         \
    if constexpr (F16) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "a"(WREG), "v"(XREG)); }  \
    else { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "a"(WREG), "v"(XREG)); }
// first MFMA of a brick on an accumulator: C = 0.  Declared read-write all the same ("+a"): a fresh definition would let the
// allocator move the accumulator to other registers and reconcile with v_accvgpr_mov at the loop's back edge -- directly in
// front of asm MFMAs whose hazards it cannot pad
#define RS_MF0(ACC, WREG, XREG)                                                                                        \
    if constexpr (F16) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "+a"(ACC) : "a"(WREG), "v"(XREG)); }   \
    else { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "+a"(ACC) : "a"(WREG), "v"(XREG)); }
// the split's conversions inside the generated schedules (tools/gen_rs*_schedule.py); RS_LRELU_MAX is LeakyReLU's max, RS_CLAMP the
// fp16 split's range clamp on a split output: one v_med3_f32 BEHIND the activation (round 5 folded the upper end into the
// activation's max as med3(t, t * slope, 65504), which passes t unclamped when t * slope > 65504: every t > 65504 at slope 1)
#define RS_W_LO(U) sf_widen_lo<F16>(U)
#define RS_W_HI(U) sf_widen_hi<F16>(U)
#define RS_CVT_PK(A, B) sf_cvt_pk<F16>(A, B)
#define RS_F_F16(...) if constexpr (F16) { __VA_ARGS__ }
#define RS_CLAMP(T) __builtin_amdgcn_fmed3f(T, -kF16Max, kF16Max)
#define RS_LRELU_MAX(T, U) __builtin_fmaxf(T, U)

// diagnostic builds only: MVSGI_RS_ABL bit 1 drops the fragment reads, 2 the epilogue, 4 the LDS-DMA (results are wrong)
#ifndef MVSGI_RS_ABL
#define MVSGI_RS_ABL 0
#endif
#if MVSGI_RS_ABL & 1
#define RS_F_READ(...)
#else
#define RS_F_READ(...) __VA_ARGS__
#endif
#if MVSGI_RS_ABL & 2
#define RS_F_EPI(...)
#else
#define RS_F_EPI(...) __VA_ARGS__
#endif
#if MVSGI_RS_ABL & 4
#define RS_F_DMA(...)
#else
#define RS_F_DMA(...) __VA_ARGS__
#endif
#if MVSGI_RS_ABL & 8      // no output stores (the value is kept alive)
#define RS_F_STORE(V, D, O) { u32x4 v_ = V; asm volatile("" ::"v"(v_)); }
#else
#define RS_F_STORE(V, D, O) __builtin_amdgcn_raw_buffer_store_b128(V, D, O, 0, (MODE == 3 ? MVSGI_RS_NT3 : MVSGI_RS_NT0));
#endif
#if MVSGI_RS_ABL & 16     // no residual requests
#define RS_F_RES(R, D, O)
#else
#define RS_F_RES(R, D, O) R = __builtin_amdgcn_raw_buffer_load_b128(D, O, 0, 0);
#endif
// 64 idle cycles: the last MFMAs' results must have landed before compiler-generated code reads the accumulators
#define RS_HAZARD_WAIT() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#ifdef MVSGI_RS_STAMPS   // diagnostic build (tools/rs_stamps.py): s_memtime stamps of workgroup 8, every wave
    int nst = 0;
#define STAMP()                                                                                     \
    if (a.dbg && blockIdx.x == 8 && nst < 250) {                                                    \
        unsigned long long t_;                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        if (lane == 0) a.dbg[wave * 256 + nst] = t_;                                                \
        nst++;                                                                                      \
    }
#else
#define STAMP()
#endif

#ifdef MVSGI_RS_STAMPS
    if (a.dbg && blockIdx.x == 8) { if (lane == 0) a.dbg[wave * 256] = t_entry_; nst = 1; }
    STAMP()      // plans made (MVSGI_RS_WEIGHTS_FIRST: weights resident too)
#endif
    // prologue: image 0 <- brick 0
    {
        const auto dsc_x = RS_DESC(a.x, c0, true);
        const int nxt_img = 0;
#pragma unroll
        for (int m = 0; m < DPW; ++m) RS_DMA(m)
    }
    if constexpr (!MVSGI_RS_WEIGHTS_FIRST) { RS_PIN_WEIGHTS() }
    f32x4 acc[8], keepA[4], keepB[4], snd[4];
    bf16x8 xh[2][4], xl[2][4];
    u32x4 rres[4], rresB[4], outp[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        keepA[k] = keepB[k] = snd[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        rres[k] = rresB[k] = outp[k] = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) xh[1][i] = xl[1][i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    STAMP()      // image 0 landed

    f32x4 pt0, pt1, pt2, pt3, t0, t1_, t2, t3;
    u32x2 sa0, sb0, sa1, sb1, sa2, sb2, sa3, sb3, hb0, hb1, hb2, hb3, lb0, lb1, lb2, lb3;
    f32x2v hf0, hf1, hf2, hf3;
    float rh0, rl0, u0, rh1, rl1, u1, rh2, rl2, u2, rh3, rl3, u3;
    float satm = 0.f;          // fp16 split, split output: running maximum |value written| (range report)
#define t1 t1_
    // phases: ph = 0 .. n + 1.  Phase ph = [pair 13 of brick ph - 1, accumulators handed over] [pairs 0 .. 12 of brick ph]
    // with, between the MFMAs: the staging of brick ph + 1, the epilogue of brick ph - 2, the residual request of brick
    // ph - 1.  The last two phases (no brick left to multiply) run the same stream without the pairs.
    __amdgpu_buffer_rsrc_t dsc_x, dsc_r, dsc_y;
    unsigned voy[2] = {0xffffff00u, 0xffffff00u};
    unsigned voc[2] = {0xffffff00u, 0xffffff00u};
    dsc_x = dsc_r = dsc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.x), 0, 0, 0x00020000);
// stores of the brick two phases back: voxels outside the volume (ragged sizes) are sent out of range
#define RS_VOY()                                                                                     \
    {                                                                                                \
        const int dok_ = (c2.od + up_od) * TD + pl < a.D;                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                \
            voy[i] = (dok_ & (int)(c2.oh * TH + 2 * s + i < a.H) & (int)(c2.ow * TW + col < a.W)) ? voyo[i] : 0xffffff00u; \
    }
// polyphase mode: correction offsets of brick c1 -- cells on an H or W face of the volume (and inside it), else out of range (= 0)
#define RS_VOC()                                                                                     \
    if constexpr (UP2) {                                                                             \
        const int dok_ = up_od * TD + pl < a.D;                                                      \
        const int iw_ = c1.ow * TW + col;                                                            \
        const int wf_ = (int)(iw_ == 0) | (int)(iw_ == a.W - 1);                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                              \
            const int ih_ = c1.oh * TH + 2 * s + i;                                                  \
            const int face_ = wf_ | (int)(ih_ == 0) | (int)(ih_ == a.H - 1);                         \
            voc[i] = (dok_ & face_ & (int)(ih_ < a.H) & (int)(iw_ < a.W)) ? vocb[i] : 0xffffff00u;   \
        }                                                                                            \
    }
    // One code path per loop (a main / drain diamond inside one loop made hipcc park the accumulators in VGPRs at the
    // loop header: 64 v_accvgpr moves per phase).
    int ph = 0;
    for (; ph < n; ++ph) {
        const int nxt_img = (ph & 1) ? 0 : BUF1;
        STAMP()
#include "conv3d_rs_phase_main.inc"
        STAMP()
        // all but the 4 youngest vector-memory operations (the residual requests): the next image has landed
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        STAMP()
        // (phases 0 and 1 ran the epilogue of bricks that do not exist -- on whatever the exchange scratch held; their stores are
        // masked, their "values" must not reach the range report either: a select, not a branch)
        if constexpr (F16 && !OUTF32) satm = ph < 2 ? 0.f : satm;
        __builtin_amdgcn_s_barrier();
    }
    for (; ph < n + 2; ++ph) {
        STAMP()
#include "conv3d_rs_phase_drain.inc"
        STAMP()
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        STAMP()
        if constexpr (F16 && !OUTF32) satm = ph < 2 ? 0.f : satm;
        __builtin_amdgcn_s_barrier();
    }
    STAMP()      // kernel exit
    if constexpr (F16 && !OUTF32) sf_sat_report(a.sat, kSatSplit, satm, kF16Max);
#undef RS_VOY
#undef RS_VOC
#undef RS_F_UP2
#undef RS_F_NUP2
#undef RS_DSC_Y
#undef RS_DSC_R
#undef RS_RES_OFF
#undef t1
#undef STAMP
#undef RS_PIN_WEIGHTS
#undef RS_MF
#undef RS_MF0
#undef RS_W_LO
#undef RS_W_HI
#undef RS_CVT_PK
#undef RS_F_F16
#undef RS_CLAMP
#undef RS_LRELU_MAX
#undef RS_DMA
#undef RS_DESC
#undef RS_DESC_OUT
#undef RS_F_SPL
#undef RS_F_F32
#undef RS_STEP
}


// ---------------------------------------------------------------------------------------------
// the 16 -> 16 kernel (post_vol: spherical_sweep_avg.py:30-36,165): split-padded volume in, plain fp32 out
// ---------------------------------------------------------------------------------------------
struct Rs16Args {
    const unsigned char* x;    // split-padded [B][D+2][H+2][W+2][16] (64 B per voxel)
    unsigned char* y;          // fp32 [B][D][H][W][16], or (OSPLIT) split-padded [B][D+2][H+2][W+2][16]
    const bf16x8* wp;          // mvsgi_conv3d_rs_pack_weights(16, 16): [5 in-plane pairs][3 kd][hi|lo][64 lanes]
    const float* scale;
    const float* shift;
    int B, D, H, W;
    float neg_slope;
    int tiles_d, tiles_h, tiles_w, total_units;
    unsigned* sat;             // the range report's words (csrc/api.cpp)
};

namespace rs16 {
constexpr int TD = 4, TH = 4, TW = 16;        // brick: 256 output voxels; a wave owns one h-row and its 4 planes
constexpr int ITD = TD + 2, ITH = TH + 2, ITW = TW + 2;
// LDS image: voxel v = d * PL + h * ROW + w; HI region (32 B per voxel: the two 8-channel halves) then LO region; chunk
// `half` of voxel (d, h, w) sits in slot half ^ ((w >> 3) & 1).  A ds_read_b128 lane group (8 lanes on one tap, 8 on its
// pair partner) then covers 16 distinct 16-byte units for in-row pairs (kw 0 | 1) and for the kw = 2 taps of adjacent rows
// (ROW = 24: a row is 48 units = 0 mod 16).
constexpr int ROW = 24, PL = ITH * ROW;       // 144
constexpr int NV = ITD * PL;                  // 864 voxels = 27 pieces per region
constexpr int REGION = NV * 32;               // 27,648
constexpr int IMG = 2 * REGION;               // 55,296
constexpr int BUF1 = 65536;
constexpr int NDMA = IMG / 1024;              // 54 pieces per image: 14 per wave, the last two of waves 2, 3 are dummies
constexpr int DPW = 14;
constexpr int LDS_BYTES = BUF1 + DPW * 4 * 1024;   // 122,880: the dummy pieces (zero-filled) land behind the image
static_assert(ROW % 8 == 0 && NV % 32 == 0 && BUF1 >= DPW * 4 * 1024, "image geometry");
// in-plane tap of pair pp for lane half `second`: (kh, kw) or none
__host__ __device__ constexpr int pair_kh(int pp, int second) { return pp < 3 ? pp : (pp == 3 ? second : (second ? -1 : 2)); }
__host__ __device__ constexpr int pair_kw(int pp, int second) { return pp < 3 ? second : 2; }
}  // namespace rs16

// [16][16][27] -> [5 pairs][3 kd][hi|lo][64 lanes][8 bf16]: lane (kg << 4) | i holds W[cout i][cin (kg>>1)*8 + j][kd][in-plane tap]
__global__ void rs16_pack_weights_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, bool f16) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 5 * 3 * 64) return;
    const int lane = idx & 63, r = idx >> 6;
    const int kd = r % 3, pp = r / 3;
    const int kg = lane >> 4, co = lane & 15, ci = (kg >> 1) * 8;
    const int kh = rs16::pair_kh(pp, kg & 1), kw = rs16::pair_kw(pp, kg & 1);
    u16x8 hi, lo;
    for (int j = 0; j < 8; ++j) {
        const float v = kh >= 0 ? w[((long long)co * 16 + ci + j) * 27 + kd * 9 + kh * 3 + kw] : 0.f;
        unsigned short h_, l_;
        sf_split_weight(v, f16, h_, l_);
        hi[j] = h_;
        lo[j] = l_;
    }
    const int o = ((pp * 3 + kd) * 2) * 64 + lane;
    wp[o] = __builtin_bit_cast(bf16x8, hi);
    wp[o + 64] = __builtin_bit_cast(bf16x8, lo);
}

// OSPLIT: the output goes to a split-padded tensor of the input's geometry (the hand-over to csrc/conv3d_s2rs.hip, which stages
// pre-split voxels by LDS-DMA) instead of plain fp32 channels-last.
// LOADERS (round 6, an experiment: -DMVSGI_RS16_LOADERS=1): the workgroup has four more waves, one per SIMD, that do nothing but issue
// the LDS-DMA pieces of the next brick (wave 4 + r those of compute wave r).  An issuing wave sits 100-185 cycles on every 1 KiB piece
// (with ONE wave per SIMD the 14 pieces per phase were 23 % of a phase in which the matrix pipe got nothing: 866 vs 1127 us
// per 64 frames without the DMA); the idea was that a loader wave sits there beside a compute wave whose MFMAs keep issuing (the
// kernel's 240 registers fit two waves per SIMD once the accumulators live in ordinary registers).  Measured, 16 frames of
// [16, 80, 320], same box: fp32 output 278 vs 288 us, split-padded output (the product's) 329 vs 324 us with the pieces paced
// 3 x 64 cycles apart; a burst (no pacing) and 5 x 64 are slower on both.  The stall is not hidden by a sibling wave: not kept.
#ifndef MVSGI_RS16_LOADER_SLEEP
#define MVSGI_RS16_LOADER_SLEEP 3      // x 64 cycles between a loader wave's pieces
#endif
template <bool OSPLIT, bool F16 = false, bool LOADERS = false>
__global__ __launch_bounds__(LOADERS ? 512 : 256, 1) void conv3d_rs16_kernel(Rs16Args a) {
    using namespace rs16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;                                      // = the brick's h-row of this wave (compute), or of its compute wave (loader)
    const int col = lane & 15, kg = lane >> 4;
    const bool second = kg & 1;
    const int half = kg >> 1;
    const int Hp = a.H + 2, Wp = a.W + 2;
    const long long frame_bytes = (long long)(a.D + 2) * Hp * Wp * 64;
    const long long total_bytes = frame_bytes * a.B;
    const long long oframe = (long long)a.D * a.H * a.W * 64, ototal = oframe * a.B;

    bf16x8 pwh[5][3], pwl[5][3];
#pragma unroll
    for (int pp = 0; pp < 5; ++pp)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const bf16x8* q = a.wp + ((pp * 3 + kd) * 2) * 64 + lane;
            pwh[pp][kd] = q[0];
            pwl[pp][kd] = q[64];
        }
#pragma unroll
    for (int pp = 0; pp < 5; ++pp)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) asm volatile("" : "+a"(pwh[pp][kd]), "+a"(pwl[pp][kd]));
    __builtin_amdgcn_sched_barrier(0);

    // fragment read bases (image 0, HI region, input plane 0, tap row 0): in-row pairs / the kw = 2 pair of rows 0, 1 / row 2
    const int w_in = col + (second ? 1 : 0), w_2 = col + 2;
    int b_in = (wave * ROW + w_in) * 32 + ((half ^ ((w_in >> 3) & 1)) << 4);
    int b_2a = (wave * ROW + w_2 + (second ? ROW : 0)) * 32 + ((half ^ ((w_2 >> 3) & 1)) << 4);
    int b_2b = (wave * ROW + w_2) * 32 + ((half ^ ((w_2 >> 3) & 1)) << 4);
    // DMA plan: piece i = wave + 4 m fills LDS bytes [i * 1024, +1024): 32 voxels x 2 chunks
    unsigned voff[DPW];
#pragma unroll
    for (int m = 0; m < DPW; ++m) {
        const int i = wave + 4 * m;
        const int region = i >= NDMA / 2 ? 1 : 0, jj = i - region * (NDMA / 2);
        const int v = 32 * jj + (lane >> 1), slot = lane & 1;
        const int d = v / PL, r = v - d * PL;
        const int h = r / ROW, w = r - h * ROW;
        const int c = slot ^ ((w >> 3) & 1);
        voff[m] = (i < NDMA && w < ITW) ? (unsigned)(((d * Hp + h) * Wp + w) * 64 + region * 32 + c * 16) : 0xffffff00u;
    }
    const f32x4 esc = *reinterpret_cast<const f32x4*>(a.scale + kg * 4), esh = *reinterpret_cast<const f32x4*>(a.shift + kg * 4);
    unsigned voy0[4];      // this lane's 16 B (couts 4 kg .. 4 kg + 3) of its row's voxel in output plane o
#pragma unroll
    for (int o = 0; o < 4; ++o)
        voy0[o] = OSPLIT ? (unsigned)((((o + 1) * Hp + wave + 1) * Wp + col + 1) * 64 + (kg & 1) * 32 + (kg >> 1) * 16)
                         : (unsigned)(((o * a.H + wave) * a.W + col) * 64 + kg * 16);

    const int total = a.total_units, G = gridDim.x;
    const int n = (total - (int)blockIdx.x + G - 1) / G;
    const int id0 = rs_xcd_remap((int)blockIdx.x, total);
    const int step = G == total ? 0 : G >> 3;
    // logical order (b, oh, od, ow), ow fastest: the bricks stacked along D share two of their four / six input planes and
    // follow each other within one XCD round (tiles_w ids apart), the bricks above / below a round or two later -- in the
    // (b, od, oh, ow) order the D neighbours were tiles_h * tiles_w ids (several L2 capacities) apart
    const int sw = step % a.tiles_w;
    int tq = step / a.tiles_w;
    const int sd = tq % a.tiles_d;
    tq /= a.tiles_d;
    const int sh_ = tq % a.tiles_h, sb = tq / a.tiles_h;
    RsUnit c1{0, 0, 0, 0}, c0, nx;
    {
        int t_ = id0;
        c0.ow = t_ % a.tiles_w;
        t_ /= a.tiles_w;
        c0.od = t_ % a.tiles_d;
        t_ /= a.tiles_d;
        c0.oh = t_ % a.tiles_h;
        c0.b = t_ / a.tiles_h;
    }
#define RS16_STEP(DST, SRC)                                                    \
    {                                                                          \
        int w_ = SRC.ow + sw, c_ = w_ >= a.tiles_w;                            \
        DST.ow = w_ - (c_ ? a.tiles_w : 0);                                    \
        int d_ = SRC.od + sd + c_;                                             \
        c_ = d_ >= a.tiles_d;                                                  \
        DST.od = d_ - (c_ ? a.tiles_d : 0);                                    \
        int h_ = SRC.oh + sh_ + c_;                                            \
        c_ = h_ >= a.tiles_h;                                                  \
        DST.oh = h_ - (c_ ? a.tiles_h : 0);                                    \
        DST.b = SRC.b + sb + c_;                                               \
    }
    RS16_STEP(nx, c0)
#define RS16_DESC(U, VALID)                                                                                      \
    ({                                                                                                           \
        const long long off_ = (long long)(U).b * frame_bytes +                                                  \
                               ((long long)((U).od * TD * Hp + (U).oh * TH) * Wp + (U).ow * TW) * 64;            \
        const long long left_ = total_bytes - off_;                                                              \
        const int rec_ = left_ > 0x7fffff00ll ? 0x7fffff00 : (int)left_;                                         \
        const int ok_ = (int)(VALID) & (int)(left_ > 0);                                                         \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.x) + off_, 0, ok_ ? rec_ : 0, 0x00020000); \
    })
#define RS16_DESC_OUT(U, VALID)                                                                                  \
    ({                                                                                                           \
        const long long off_ = OSPLIT ? (long long)(U).b * frame_bytes + ((long long)((U).od * TD * Hp + (U).oh * TH) * Wp + (U).ow * TW) * 64 \
                                      : (long long)(U).b * oframe + ((long long)((U).od * TD * a.H + (U).oh * TH) * a.W + (U).ow * TW) * 64; \
        const long long left_ = (OSPLIT ? total_bytes : ototal) - off_;                                          \
        const int rec_ = left_ > 0x7fffff00ll ? 0x7fffff00 : (int)left_;                                         \
        const int ok_ = (int)(VALID) & (int)(left_ > 0);                                                         \
        __builtin_amdgcn_make_buffer_rsrc(a.y + off_, 0, ok_ ? rec_ : 0, 0x00020000);                            \
    })
#define RS16_DMA(M)                                                                                              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(dsc_x, (__attribute__((address_space(3))) void*)(lds + nxt_img + (wave + 4 * (M)) * 1024), \
                                             16, voff[M], 0, 0, 0);
#define RS16_VOY()                                                                                               \
    {                                                                                                            \
        const int hok_ = c1.oh * TH + wave < a.H;                                                                \
        _Pragma("unroll") for (int o = 0; o < 4; ++o)                                                            \
            voy[o] = (hok_ & (int)(c1.od * TD + o < a.D) & (int)(c1.ow * TW + col < a.W)) ? voy0[o] : 0xffffff00u; \
    }
// (LOADERS: two waves per SIMD = 256 registers per wave, which hipcc splits 128 / 128 between the two files: the 120 weight registers
// stay in the accumulator file, the four accumulators move to ordinary registers)
#define RS_MF(ACC, WREG, XREG)                                                                                         \
    if constexpr (LOADERS) {                                                                                           \
        if constexpr (F16) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(WREG), "v"(XREG)); }  \
        else { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "a"(WREG), "v"(XREG)); }           \
    } else if constexpr (F16) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "a"(WREG), "v"(XREG)); }  \
    else { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "a"(WREG), "v"(XREG)); }
// first MFMA of a brick on an accumulator: C = 0.  Declared read-write all the same ("+a"): a fresh definition would let the
// allocator move the accumulator to other registers and reconcile with v_accvgpr_mov at the loop's back edge -- directly in
// front of asm MFMAs whose hazards it cannot pad
#define RS_MF0(ACC, WREG, XREG)                                                                                        \
    if constexpr (LOADERS) {                                                                                           \
        if constexpr (F16) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "+v"(ACC) : "a"(WREG), "v"(XREG)); }   \
        else { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "+v"(ACC) : "a"(WREG), "v"(XREG)); }            \
    } else if constexpr (F16) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "+a"(ACC) : "a"(WREG), "v"(XREG)); }   \
    else { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "+a"(ACC) : "a"(WREG), "v"(XREG)); }
// the split's conversions inside the generated schedules (tools/gen_rs*_schedule.py); RS_LRELU_MAX is LeakyReLU's max, RS_CLAMP the
// fp16 split's range clamp on a split output: one v_med3_f32 BEHIND the activation (round 5 folded the upper end into the
// activation's max as med3(t, t * slope, 65504), which passes t unclamped when t * slope > 65504: every t > 65504 at slope 1)
#define RS_W_LO(U) sf_widen_lo<F16>(U)
#define RS_W_HI(U) sf_widen_hi<F16>(U)
#define RS_CVT_PK(A, B) sf_cvt_pk<F16>(A, B)
#define RS_F_F16(...) if constexpr (F16) { __VA_ARGS__ }
#define RS_CLAMP(T) __builtin_amdgcn_fmed3f(T, -kF16Max, kF16Max)
#define RS_LRELU_MAX(T, U) __builtin_fmaxf(T, U)
#define RS_PIN_V(V) asm volatile("" : "+v"(V));
#define RS_F_STORE16(V, D, O) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, V), D, O, 0, MVSGI_RS16_NT);
#define RS16_F_SPL(...) if constexpr (OSPLIT) { __VA_ARGS__ }
#define RS16_F_F32(...) if constexpr (!OSPLIT) { __VA_ARGS__ }

    if constexpr (LOADERS) {
        if (wave8 >= 4) {
            // ---- a loader wave: brick 0 -> image 0, then during phase ph brick ph + 1 -> the other image; the same barriers as the
            // compute waves (one behind the prologue, one per phase).  The image filled in phase ph was read in phase ph - 1: every
            // compute wave passed that phase's barrier behind its last fragment read. ----
            {
                const auto dsc_x = RS16_DESC(c0, 1);
                const int nxt_img = 0;
#pragma unroll
                for (int m = 0; m < DPW; ++m) RS16_DMA(m)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int ph = 0; ph < n; ++ph) {
                const int nxt_img = (ph & 1) ? 0 : BUF1;
                const auto dsc_x = RS16_DESC(nx, (int)(ph + 1 < n));
#pragma unroll
                for (int m = 0; m < DPW; ++m) {
                    RS16_DMA(m)
                    // paced over the phase (~5300 cycles): in a burst the workgroup's 56 requests queue up in the CU's address path
                    // in front of the compute waves' fragment reads
                    __builtin_amdgcn_s_sleep(MVSGI_RS16_LOADER_SLEEP);
                    __builtin_amdgcn_sched_barrier(0);
                }
                c0 = nx;
                RS16_STEP(nx, c0)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            return;
        }
    }
#define RS16_F_OWN(...) if constexpr (!LOADERS) { __VA_ARGS__ }
    __amdgpu_buffer_rsrc_t dsc_x, dsc_y;
    u32x2 hb0, hb1, hb2, hb3, lb0, lb1, lb2, lb3, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
    f32x2v hf0, hf1, hf2, hf3;
    hb0 = hb1 = hb2 = hb3 = lb0 = lb1 = lb2 = lb3 = sa0 = sa1 = sa2 = sa3 = sb0 = sb1 = sb2 = sb3 = u32x2{0u, 0u};
    hf0 = hf1 = hf2 = hf3 = f32x2v{0.f, 0.f};
    unsigned voy[4] = {0xffffff00u, 0xffffff00u, 0xffffff00u, 0xffffff00u};
    dsc_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0, 0x00020000);
    if constexpr (!LOADERS) {   // prologue: image 0 <- brick 0
        dsc_x = RS16_DESC(c0, 1);
        const int nxt_img = 0;
#pragma unroll
        for (int m = 0; m < DPW; ++m) RS16_DMA(m)
    } else {
        dsc_x = RS16_DESC(c0, 1);
    }
    f32x4 acc[4], fin[4];
    bf16x8 xh[3][2], xl[3][2];
    float u0, u1, u2, u3;
    float satm = 0.f;          // fp16 split, split output: running maximum |value written| (range report)
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] = fin[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int q = 0; q < 2; ++q) xh[g][q] = xl[g][q] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // phase ph = [groups 13, 14 of brick ph - 1, hand-over, its epilogue] [groups 0 .. 12 of brick ph, staging of brick ph + 1]
    int ph = 0;
    for (; ph < n; ++ph) {
        const int nxt_img = (ph & 1) ? 0 : BUF1;
        if constexpr (OSPLIT) {
#include "conv3d_rs16s_phase_main.inc"
        } else {
#include "conv3d_rs16_phase_main.inc"
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    {
        if constexpr (OSPLIT) {
#include "conv3d_rs16s_phase_drain.inc"
        } else {
#include "conv3d_rs16_phase_drain.inc"
        }
    }
    if constexpr (F16 && OSPLIT) sf_sat_report(a.sat, kSatSplit, satm, kF16Max);
#undef RS16_STEP
#undef RS16_DESC
#undef RS16_DESC_OUT
#undef RS16_DMA
#undef RS16_VOY
#undef RS_MF
#undef RS_MF0
#undef RS_W_LO
#undef RS_W_HI
#undef RS_CVT_PK
#undef RS_F_F16
#undef RS_CLAMP
#undef RS_LRELU_MAX
#undef RS_PIN_V
#undef RS_F_STORE16
#undef RS16_F_SPL
#undef RS16_F_F32
#undef RS16_F_OWN
}

}  // namespace

extern "C" size_t mvsgi_act_split_bytes(int B, int C, int D, int H, int W) {
    return (size_t)B * (D + 2) * (H + 2) * (W + 2) * (size_t)C * 4;
}

// fmt: 0 = the bf16 split, MVSGI_SPLIT_F16 = the fp16 split (csrc/split_fmt.hpp)
extern "C" int mvsgi_act_f32_to_split_fmt(const float* x, void* y, int B, int C, int D, int H, int W, int fmt, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x && y, "mvsgi_act_f32_to_split: null pointer");
    MVSGI_REQUIRE(B > 0 && C > 0 && C % 16 == 0 && D > 0 && H > 0 && W > 0, "mvsgi_act_f32_to_split: bad dims (C %% 16 == 0)");
    MVSGI_REQUIRE(fmt == 0 || fmt == MVSGI_SPLIT_F16, "mvsgi_act_f32_to_split: fmt %d not in {0, MVSGI_SPLIT_F16}", fmt);
    const long long n = (long long)B * D * H * W * (C / 8);
    MVSGI_REQUIRE(mvsgi::cdiv(n, 256) < (1ll << 31), "mvsgi_act_f32_to_split: tensor too large");
    MVSGI_SAT_WORDS(sat);
    if (fmt)
        hipLaunchKernelGGL(f32_to_split_kernel<true>, dim3((unsigned)mvsgi::cdiv(n, 256)), dim3(256), 0, mvsgi::as_stream(stream), x,
                           static_cast<unsigned char*>(y), B, C, D, H, W, sat);
    else
        hipLaunchKernelGGL(f32_to_split_kernel<false>, dim3((unsigned)mvsgi::cdiv(n, 256)), dim3(256), 0, mvsgi::as_stream(stream), x,
                           static_cast<unsigned char*>(y), B, C, D, H, W, sat);
    return mvsgi::check_launch("mvsgi_act_f32_to_split");
}
extern "C" int mvsgi_act_f32_to_split(const float* x, void* y, int B, int C, int D, int H, int W, mvsgi_stream_t stream) {
    return mvsgi_act_f32_to_split_fmt(x, y, B, C, D, H, W, 0, stream);
}

extern "C" int mvsgi_act_split_to_f32_fmt(const void* x, float* y, int B, int C, int D, int H, int W, int fmt, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x && y, "mvsgi_act_split_to_f32: null pointer");
    MVSGI_REQUIRE(B > 0 && C > 0 && C % 16 == 0 && D > 0 && H > 0 && W > 0, "mvsgi_act_split_to_f32: bad dims (C %% 16 == 0)");
    MVSGI_REQUIRE(fmt == 0 || fmt == MVSGI_SPLIT_F16, "mvsgi_act_split_to_f32: fmt %d not in {0, MVSGI_SPLIT_F16}", fmt);
    const long long n = (long long)B * D * H * W * (C / 8);
    MVSGI_REQUIRE(mvsgi::cdiv(n, 256) < (1ll << 31), "mvsgi_act_split_to_f32: tensor too large");
    if (fmt)
        hipLaunchKernelGGL(split_to_f32_kernel<true>, dim3((unsigned)mvsgi::cdiv(n, 256)), dim3(256), 0, mvsgi::as_stream(stream),
                           static_cast<const unsigned char*>(x), y, B, C, D, H, W);
    else
        hipLaunchKernelGGL(split_to_f32_kernel<false>, dim3((unsigned)mvsgi::cdiv(n, 256)), dim3(256), 0, mvsgi::as_stream(stream),
                           static_cast<const unsigned char*>(x), y, B, C, D, H, W);
    return mvsgi::check_launch("mvsgi_act_split_to_f32");
}
extern "C" int mvsgi_act_split_to_f32(const void* x, float* y, int B, int C, int D, int H, int W, mvsgi_stream_t stream) {
    return mvsgi_act_split_to_f32_fmt(x, y, B, C, D, H, W, 0, stream);
}

extern "C" size_t mvsgi_conv3d_rs_packed_weight_bytes(int Cout, int Cin) {
    if (Cout == 16 && Cin == 16) return (size_t)5 * 3 * 2 * 64 * 16;
    return Cout == 32 && Cin == 32 ? (size_t)2 * 2 * rs::kPairs * 2 * 64 * 16 : 0;
}

extern "C" int mvsgi_conv3d_rs_pack_weights_fmt(const float* w_oidhw, void* w_packed, int Cout, int Cin, int fmt, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(w_oidhw && w_packed, "mvsgi_conv3d_rs_pack_weights: null pointer");
    MVSGI_REQUIRE(fmt == 0 || fmt == MVSGI_SPLIT_F16, "mvsgi_conv3d_rs_pack_weights: fmt %d not in {0, MVSGI_SPLIT_F16}", fmt);
    if (Cout == 16 && Cin == 16) {
        hipLaunchKernelGGL(rs16_pack_weights_kernel, dim3((5 * 3 * 64 + 255) / 256), dim3(256), 0, mvsgi::as_stream(stream), w_oidhw,
                           static_cast<bf16x8*>(w_packed), fmt != 0);
        return mvsgi::check_launch("mvsgi_conv3d_rs_pack_weights");
    }
    MVSGI_REQUIRE(Cout == 32 && Cin == 32, "mvsgi_conv3d_rs_pack_weights: only 32 -> 32 and 16 -> 16 channels (got %d -> %d)", Cin, Cout);
    hipLaunchKernelGGL(rs_pack_weights_kernel, dim3((2 * 2 * rs::kPairs * 64 + 255) / 256), dim3(256), 0, mvsgi::as_stream(stream),
                       w_oidhw, static_cast<bf16x8*>(w_packed), fmt != 0);
    return mvsgi::check_launch("mvsgi_conv3d_rs_pack_weights");
}
extern "C" int mvsgi_conv3d_rs_pack_weights(const float* w_oidhw, void* w_packed, int Cout, int Cin, mvsgi_stream_t stream) {
    return mvsgi_conv3d_rs_pack_weights_fmt(w_oidhw, w_packed, Cout, Cin, 0, stream);
}

extern "C" int mvsgi_conv3d_rs_split_fmt(const void* x, const void* w_packed_rs, const float* scale, const float* shift,
                                         const void* res, void* y, int y_is_f32, int B, int Cin, int D, int H, int W, int Cout,
                                         float neg_slope, int fmt, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x && y && w_packed_rs && scale && shift, "mvsgi_conv3d_rs_split: null pointer");
    MVSGI_REQUIRE(fmt == 0 || fmt == MVSGI_SPLIT_F16, "mvsgi_conv3d_rs_split: fmt %d not in {0, MVSGI_SPLIT_F16}", fmt);
    MVSGI_REQUIRE(Cin == 32 && Cout == 32, "mvsgi_conv3d_rs_split: only 32 -> 32 channels (got %d -> %d)", Cin, Cout);
    MVSGI_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "mvsgi_conv3d_rs_split: bad dims");
    MVSGI_REQUIRE(x != y, "mvsgi_conv3d_rs_split: in-place operation is not supported");
    MVSGI_REQUIRE(neg_slope >= 0.f && neg_slope <= 1.f, "mvsgi_conv3d_rs_split: neg_slope %g not in [0, 1]", (double)neg_slope);
    MVSGI_REQUIRE((long long)(D + 2) * (H + 2) * (W + 2) * 128 < (1ll << 31), "mvsgi_conv3d_rs_split: frame too large for 32-bit offsets");
    RsArgs a{};
    a.x = static_cast<const unsigned char*>(x);
    a.y = static_cast<unsigned char*>(y);
    a.res = static_cast<const unsigned char*>(res);
    a.wp = static_cast<const bf16x8*>(w_packed_rs);
    a.scale = scale;
    a.shift = shift;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.neg_slope = neg_slope;
    a.tiles_d = (int)mvsgi::cdiv(D, rs::TD);
    a.tiles_h = (int)mvsgi::cdiv(H, rs::TH);
    a.tiles_w = (int)mvsgi::cdiv(W, rs::TW);
    const long long nb = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
    MVSGI_REQUIRE(nb < (1ll << 31), "mvsgi_conv3d_rs_split: too many units");
    a.total_units = (int)nb;
    MVSGI_SAT_WORDS(sat_words_);
    a.sat = sat_words_;
#ifdef MVSGI_RS_STAMPS
    {   // MVSGI_STAMP=1: record; =2: print the stamps of the previous launch
        static unsigned long long* dbgbuf = nullptr;
        const char* e_ = getenv("MVSGI_STAMP");
        if (e_ && !dbgbuf) { (void)hipMalloc(&dbgbuf, 4 * 256 * 8); (void)hipMemset(dbgbuf, 0, 4 * 256 * 8); }
        a.dbg = e_ ? dbgbuf : nullptr;
        if (e_ && atoi(e_) == 2 && dbgbuf) {
            static unsigned long long h[4 * 256];
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, dbgbuf, sizeof(h), hipMemcpyDeviceToHost);
            for (int w = 0; w < 4; ++w) {
                fprintf(stderr, "wave %d:", w);
                for (int i = 0; i < 250; ++i) fprintf(stderr, " %lld", (long long)(h[w * 256 + i] - h[0]));
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    static mvsgi::PersistentGeom geo_cache[4][mvsgi::kMaxDevices] = {};
    mvsgi::PersistentGeom geo;
    void (*kern)(RsArgs) = fmt ? (y_is_f32 ? conv3d_rs32_kernel<1, true> : conv3d_rs32_kernel<0, true>)
                               : (y_is_f32 ? conv3d_rs32_kernel<1, false> : conv3d_rs32_kernel<0, false>);
    if (mvsgi::persistent_geometry(kern, 256, rs::LDS_BYTES, 1, geo_cache[(y_is_f32 ? 1 : 0) + (fmt ? 2 : 0)], "mvsgi_conv3d_rs_split", geo)) return 1;
    const long long resident = (long long)geo.cus / 8 * 8 > 0 ? (long long)geo.cus / 8 * 8 : 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nb <= resident ? nb : resident)), dim3(256), rs::LDS_BYTES,
                       mvsgi::as_stream(stream), a);
    return mvsgi::check_launch("mvsgi_conv3d_rs_split");
}
extern "C" int mvsgi_conv3d_rs_split(const void* x, const void* w_packed_rs, const float* scale, const float* shift,
                                     const void* res, void* y, int y_is_f32, int B, int Cin, int D, int H, int W, int Cout,
                                     float neg_slope, mvsgi_stream_t stream) {
    return mvsgi_conv3d_rs_split_fmt(x, w_packed_rs, scale, shift, res, y, y_is_f32, B, Cin, D, H, W, Cout, neg_slope, 0, stream);
}

// ---- polyphase ResizeConv3d on the 32 -> 32 kernel (MODE 2): internal entry points used by csrc/conv3d_up2poly.hip ----
namespace mvsgi {

// the host statement of rs_pack_weights_kernel: [32][32][27] fp32 -> kRs32PackedBytes in the register-stationary lane order
void rs32_pack_weights_host(const float* w, void* packed, bool f16) {
    unsigned short* out = static_cast<unsigned short*>(packed);
    for (int sl = 0; sl < 2; ++sl)
        for (int ct = 0; ct < 2; ++ct)
            for (int p = 0; p < rs::kPairs; ++p)
                for (int lane = 0; lane < 64; ++lane) {
                    const int kg = lane >> 4, co = ct * 16 + (lane & 15), ci = sl * 16 + (kg >> 1) * 8;
                    const int k = rs::pair_k(p, kg & 1), kw = rs::pair_kw(p, kg & 1);
                    const size_t o = ((((size_t)(sl * 2 + ct) * rs::kPairs + p) * 2) * 64 + lane) * 8;
                    for (int j = 0; j < 8; ++j) {
                        const float v = k >= 0 ? w[((size_t)co * 32 + ci + j) * 27 + k * 3 + kw] : 0.f;
                        unsigned short h, l;
                        sf_split_weight(v, f16, h, l);
                        out[o + j] = h;
                        out[o + 64 * 8 + j] = l;
                    }
                }
}

int rs32_up2_launch(const void* x_split, const void* w_sets, const float* scale32, const float* shift32, void* y, int y_is_split,
                    int B, int D, int H, int W, float neg_slope, bool f16, hipStream_t st) {
    MVSGI_REQUIRE((long long)(D + 2) * (H + 2) * (W + 2) * 128 < (1ll << 31) && (long long)D * H * W * 512 < (1ll << 31),
                  "mvsgi_conv3d_up2_poly_f32: frame too large for 32-bit offsets");
    RsArgs a{};
    a.x = static_cast<const unsigned char*>(x_split);
    a.y = static_cast<unsigned char*>(y);
    a.res = nullptr;
    a.wp = static_cast<const bf16x8*>(w_sets);
    a.scale = scale32;
    a.shift = shift32;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.neg_slope = neg_slope;
    a.tiles_d = (int)cdiv(D, rs::TD);
    a.tiles_h = (int)cdiv(H, rs::TH);
    a.tiles_w = (int)cdiv(W, rs::TW);
    const long long per_role = (long long)B * a.tiles_h * a.tiles_w;
    MVSGI_REQUIRE(per_role < (1ll << 31), "mvsgi_conv3d_up2_poly_f32: too many units");
    a.total_units = (int)per_role;
    MVSGI_SAT_WORDS(sat_words_);
    a.sat = sat_words_;
    a.wp_set = (long long)(kRs32PackedBytes / 16);
    MVSGI_REQUIRE((long long)(2 * D + 2) * (2 * H + 2) * (2 * W + 2) * 64 < (1ll << 31), "mvsgi_conv3d_up2_poly: output frame too large for 32-bit offsets");
    static PersistentGeom geo_cache[4][kMaxDevices] = {};
    PersistentGeom geo;
    void (*kern)(RsArgs) = f16 ? (y_is_split ? conv3d_rs32_kernel<3, true> : conv3d_rs32_kernel<2, true>)
                               : (y_is_split ? conv3d_rs32_kernel<3, false> : conv3d_rs32_kernel<2, false>);
    if (persistent_geometry(kern, 256, rs::LDS_BYTES, 1, geo_cache[(y_is_split ? 1 : 0) + (f16 ? 2 : 0)], "mvsgi_conv3d_up2_poly_f32", geo)) return 1;
    // grid = 8 XCDs x R roles x walkers; one workgroup per CU when the roles fit, never fewer than one walker per (XCD, role)
    const int R = 4 * a.tiles_d;
    long long walkers = geo.cus / (8 * R);
    const long long most = cdiv(per_role, 8);             // bricks of the largest XCD share
    if (walkers > most) walkers = most;
    if (walkers < 1) walkers = 1;
    a.walkers = (int)walkers;
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * R * walkers)), dim3(256), rs::LDS_BYTES, st, a);
    return check_launch("mvsgi_conv3d_up2_poly_f32(main)");
}

}  // namespace mvsgi

// BaseConvBlk3d.forward for Cin = Cout = 16, stride 1, no residual (post_vol, spherical_sweep_avg.py:30-36,165) on a
// split-padded input, fp32 [B][D][H][W][16] output; w_packed_rs from mvsgi_conv3d_rs_pack_weights(16, 16).
namespace {
int rs16_run(const void* x, const void* w_packed_rs, const float* scale, const float* shift, void* y, int y_is_split,
             int B, int D, int H, int W, float neg_slope, int fmt, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(fmt == 0 || fmt == MVSGI_SPLIT_F16, "mvsgi_conv3d_rs16_split: fmt %d not in {0, MVSGI_SPLIT_F16}", fmt);
    MVSGI_REQUIRE(x && y && w_packed_rs && scale && shift, "mvsgi_conv3d_rs16_split: null pointer");
    MVSGI_REQUIRE(x != y, "mvsgi_conv3d_rs16_split: in-place operation is not supported");
    MVSGI_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "mvsgi_conv3d_rs16_split: bad dims");
    MVSGI_REQUIRE(neg_slope >= 0.f && neg_slope <= 1.f, "mvsgi_conv3d_rs16_split: neg_slope %g not in [0, 1]", (double)neg_slope);
    MVSGI_REQUIRE((long long)(D + 2) * (H + 2) * (W + 2) * 64 < (1ll << 31), "mvsgi_conv3d_rs16_split: frame too large for 32-bit offsets");
    Rs16Args a{};
    a.x = static_cast<const unsigned char*>(x);
    a.y = static_cast<unsigned char*>(y);
    a.wp = static_cast<const bf16x8*>(w_packed_rs);
    a.scale = scale;
    a.shift = shift;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.neg_slope = neg_slope;
    a.tiles_d = (int)mvsgi::cdiv(D, rs16::TD);
    a.tiles_h = (int)mvsgi::cdiv(H, rs16::TH);
    a.tiles_w = (int)mvsgi::cdiv(W, rs16::TW);
    const long long nb = (long long)B * a.tiles_d * a.tiles_h * a.tiles_w;
    MVSGI_REQUIRE(nb < (1ll << 31), "mvsgi_conv3d_rs16_split: too many units");
    a.total_units = (int)nb;
    MVSGI_SAT_WORDS(sat_words_);
    a.sat = sat_words_;
    // -DMVSGI_RS16_LOADERS=1 builds the eight-wave kernel (four compute + four loader waves, LOADERS above): measured equal within
    // +-3 % (profiles/r06_post_vol_loader_waves.txt) -- the product stays on four waves
#ifndef MVSGI_RS16_LOADERS
#define MVSGI_RS16_LOADERS 0
#endif
    constexpr bool kLoaders = MVSGI_RS16_LOADERS != 0;
    static mvsgi::PersistentGeom geo_cache[4][mvsgi::kMaxDevices] = {};
    mvsgi::PersistentGeom geo;
    void (*kern)(Rs16Args) = fmt ? (y_is_split ? conv3d_rs16_kernel<true, true, kLoaders> : conv3d_rs16_kernel<false, true, kLoaders>)
                                 : (y_is_split ? conv3d_rs16_kernel<true, false, kLoaders> : conv3d_rs16_kernel<false, false, kLoaders>);
    constexpr int kThreads = kLoaders ? 512 : 256;
    if (mvsgi::persistent_geometry(kern, kThreads, rs16::LDS_BYTES, 1, geo_cache[(y_is_split ? 1 : 0) + (fmt ? 2 : 0)], "mvsgi_conv3d_rs16_split", geo))
        return 1;
    const long long resident = (long long)geo.cus / 8 * 8 > 0 ? (long long)geo.cus / 8 * 8 : 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nb <= resident ? nb : resident)), dim3(kThreads), rs16::LDS_BYTES, mvsgi::as_stream(stream), a);
    return mvsgi::check_launch("mvsgi_conv3d_rs16_split");
}
}  // namespace

extern "C" int mvsgi_conv3d_rs16_split(const void* x, const void* w_packed_rs, const float* scale, const float* shift, float* y,
                                       int B, int D, int H, int W, float neg_slope, mvsgi_stream_t stream) {
    return rs16_run(x, w_packed_rs, scale, shift, y, 0, B, D, H, W, neg_slope, 0, stream);
}

// the same layer writing a split-padded [B][D+2][H+2][W+2][16] tensor (zero-bordered by the caller, interior written)
extern "C" int mvsgi_conv3d_rs16_split_out_split(const void* x, const void* w_packed_rs, const float* scale, const float* shift, void* y_split,
                                                 int B, int D, int H, int W, float neg_slope, mvsgi_stream_t stream) {
    return rs16_run(x, w_packed_rs, scale, shift, y_split, 1, B, D, H, W, neg_slope, 0, stream);
}

// either output (y_is_split: 0 = fp32 [B][D][H][W][16], 1 = split-padded) in either split (fmt: 0 | MVSGI_SPLIT_F16: input, weights
// and a split output all in that split)
extern "C" int mvsgi_conv3d_rs16_split_fmt(const void* x, const void* w_packed_rs, const float* scale, const float* shift, void* y,
                                           int y_is_split, int B, int D, int H, int W, float neg_slope, int fmt, mvsgi_stream_t stream) {
    return rs16_run(x, w_packed_rs, scale, shift, y, y_is_split ? 1 : 0, B, D, H, W, neg_slope, fmt, stream);
}
