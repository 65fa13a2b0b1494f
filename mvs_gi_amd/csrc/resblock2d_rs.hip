// K5: the feature extractor's residual block on pre-split activations ("register-stationary", as K2e is for the regulator):
// ResConvBlk2d.forward (dsta_mvs/model/common/common_modules.py:165-176) for 16 -> 16 channels, 3x3, stride 1,
//   r = LReLU(BN1(conv1(x)));  y = LReLU(BN2(conv2(r)) + x)
// in ONE launch with r never leaving the CU -- the arithmetic of resblock2d_bf16x3.hpp, another schedule.
//
// Why a second schedule.  resblock2d_bf16x3_kernel gives half of its waves to splitting fp32 pixels into hi | lo for LDS and
// needs 147 KB of LDS (80-byte pixels, two input windows): ONE workgroup per CU, whose four consumer waves run conv1 ->
// epilogue -> barrier -> conv2 -> epilogue -> barrier back to back with nothing beside them: 35 % matrix-core time, 3.5 TB/s.
// Here
//   * activations travel between the blocks ALREADY SPLIT (format below): staging a brick's 18 x 34 input window is a pure
//     copy and runs on the LDS-DMA path (buffer_load ... lds: no VGPR round trip, no VALU, no producer waves);
//   * LDS pixels are 32 B in a HI and 32 B in a LO region (the DMA's contiguous KiB pieces), conflict-free for ds_read_b128
//     because a lane group's two 8-lane halves read DIFFERENT channel halves of the same tap (k = tap-of-pair x 16 + channel:
//     8 consecutive pixels x 2 chunks cover a 256-byte bank row once, whatever the row pitch);
//   * one window + one conv1-result image = 81 KB: TWO workgroups of four waves per CU, so one workgroup's epilogues,
//     barriers and DMA waits run under the other's MFMAs;
//   * the residual comes from the window in LDS (x = hi + lo, what conv1 multiplied), not from HBM / L2 again;
//   * the window of brick u + 1 is requested right after conv1 of brick u has released the image and lands under conv2.
// Both weight sets stay in registers for the whole launch (80 VGPRs).
//
// 2-D split-padded activation format:  [N][H + 4][W + 4][64 B], pixel record = [hi(c 0-7) | hi(c 8-15) | lo(c 0-7) | lo(c 8-15)]
// bf16, x = hi + lo, hi = bf16(x) (RNE), lo = bf16(x - hi) -- the 16-channel slice record of csrc/conv3d_rs.hip -- with a
// TWO-pixel zero border that is never written: a brick's window (two pixels of halo: two stacked 3x3 convolutions) is a fixed
// pattern of offsets from the brick's origin, always inside the tensor on the low side; past the tensor's end the buffer
// range check returns zeros.
#include "common.hpp"
#include <cstdlib>
#ifdef MVSGI_RS_STAMPS
#include <cstdio>
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4(const f32x4 x, u32x2& hi, u32x2& lo) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2v v = {x[2 * p], x[2 * p + 1]};
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        const f32x2v hf = {__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
        hi[p] = hb;
        lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, bf16x2));
    }
}
__device__ __forceinline__ f32x4 join4(const u32x2 hi, const u32x2 lo) {
    f32x4 r;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        r[2 * p] = __builtin_bit_cast(float, hi[p] << 16) + __builtin_bit_cast(float, lo[p] << 16);
        r[2 * p + 1] = __builtin_bit_cast(float, hi[p] & 0xffff0000u) + __builtin_bit_cast(float, lo[p] & 0xffff0000u);
    }
    return r;
}

// LeakyReLU for slopes in [0, 1] as mul + max (plain asm max: __builtin_fmaxf first canonicalises an MFMA result with a third op)
__device__ __forceinline__ float lrelu(const float v, const float slope) {
    const float m = v * slope;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(m));
    return r;
}

#ifndef MVSGI_RB_ST_AUX
#define MVSGI_RB_ST_AUX 2     // cache policy bits of the output stores: nt (never re-read by this launch; 9.37 -> 9.22 ms per 32 frames)
#endif
#ifndef MVSGI_RB_EPI_OVERLAP
#define MVSGI_RB_EPI_OVERLAP 1
#endif
#ifndef MVSGI_RB_DMA_SPREAD
#define MVSGI_RB_DMA_SPREAD 2     // 0: all 10 pieces behind the barrier; 1: two per step over conv2's first five steps; 2: one per step
#endif
#ifndef MVSGI_RB_LD_AUX
#define MVSGI_RB_LD_AUX 0     // cache policy bits of the window DMA (nt measured slower: the halos are re-read from L2)
#endif

namespace rb {
constexpr int PAD = 2;                        // border of the 2-D split-padded format
constexpr int TOH = 14, TOW = 30;             // output brick
constexpr int RH = TOH + 2, RW = TOW + 2;     // conv1 region 16 x 32 = 32 MFMA tiles of 16 pixels
constexpr int IH = TOH + 4, IW = TOW + 4;     // input window 18 x 34
constexpr int NPX = IH * IW;                  // 612 pixels
constexpr int PIECES = (NPX + 31) / 32;       // 20 DMA pieces of 32 pixels x 32 B per region
constexpr int REGION = PIECES * 1024;         // 20,480: HI region, then the LO region
constexpr int IMGA = 2 * REGION;              // 40,960: the window
constexpr int IMGB = REGION + NPX * 32;       // 40,064: the conv1 result (LO region at the same distance)
constexpr int LDS_BYTES = IMGA + IMGB;        // 81,024: two workgroups per CU
constexpr int DPW = 2 * PIECES / 4;           // 10 pieces per wave
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
static_assert(REGION + 7 * 2 * IW * 32 < 65536, "a wave's tiles are 16-bit immediates from its base addresses");
}  // namespace rb

// ---------------------------------------------------------------------------------------------
// format conversion (module boundaries, tests): fp32 [N][H][W][16] <-> 2-D split-padded.  One thread per (pixel, 8 channels).
// ---------------------------------------------------------------------------------------------
__global__ void f32_to_split2d_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, int N, int H, int W) {
    const long long n = (long long)N * H * W * 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int g = (int)(idx & 1);
    long long v = idx >> 1;
    const int w = (int)(v % W);
    v /= W;
    const int h = (int)(v % H);
    const int b = (int)(v / H);
    const float* src = x + (((long long)b * H + h) * W + w) * 16 + g * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
    u32x2 h0, l0, h1, l1;
    split4(a0, h0, l0);
    split4(a1, h1, l1);
    unsigned char* dst = y + (((long long)b * (H + 2 * rb::PAD) + h + rb::PAD) * (W + 2 * rb::PAD) + w + rb::PAD) * 64 + g * 16;
    *reinterpret_cast<u32x4*>(dst) = u32x4{h0[0], h0[1], h1[0], h1[1]};
    *reinterpret_cast<u32x4*>(dst + 32) = u32x4{l0[0], l0[1], l1[0], l1[1]};
}

__global__ void split2d_to_f32_kernel(const unsigned char* __restrict__ x, float* __restrict__ y, int N, int H, int W) {
    const long long n = (long long)N * H * W * 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int g = (int)(idx & 1);
    long long v = idx >> 1;
    const int w = (int)(v % W);
    v /= W;
    const int h = (int)(v % H);
    const int b = (int)(v / H);
    const unsigned char* src = x + (((long long)b * (H + 2 * rb::PAD) + h + rb::PAD) * (W + 2 * rb::PAD) + w + rb::PAD) * 64 + g * 16;
    const u32x4 hi = *reinterpret_cast<const u32x4*>(src), lo = *reinterpret_cast<const u32x4*>(src + 32);
    float* dst = y + (((long long)b * H + h) * W + w) * 16 + g * 8;
    *reinterpret_cast<f32x4*>(dst) = join4(u32x2{hi[0], hi[1]}, u32x2{lo[0], lo[1]});
    *reinterpret_cast<f32x4*>(dst + 4) = join4(u32x2{hi[2], hi[3]}, u32x2{lo[2], lo[3]});
}

// [Cout 16][Cin 16][3][3] x scale[Cout] -> [5 pairs][hi | lo][64 lanes][8 bf16]
//   lane = (kg << 4) | i holds scale[i] * W[cout = i][cin = (kg & 1) * 8 + j][tap = 2 p + (kg >> 1)]  (tap 9 of pair 4: zeros)
// The BatchNorm scale is folded into the weights here (fp32 product, then hi | lo) and the shift is the accumulators' initial
// value: the epilogues have no scale / shift arithmetic left.
__global__ void rb_pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale, bf16x8* __restrict__ wp) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= 5 * 64) return;
    const int lane = idx & 63, p = idx >> 6;
    const int kg = lane >> 4, co = lane & 15, ci = (kg & 1) * 8, tap = 2 * p + (kg >> 1);
    bf16x8 hi, lo;
    for (int j = 0; j < 8; ++j) {
        const float v = tap < 9 ? w[(co * 16 + ci + j) * 9 + tap] * scale[co] : 0.f;
        const __bf16 h = (__bf16)v;
        hi[j] = h;
        lo[j] = (__bf16)(v - (float)h);
    }
    wp[(p * 2) * 64 + lane] = hi;
    wp[(p * 2 + 1) * 64 + lane] = lo;
}

struct RbArgs {
    const unsigned char* x;    // 2-D split-padded [N][H+4][W+4][64]
    unsigned char* y;          // the same geometry, or (OUTF32) plain fp32 [N][H][W][16]
    const bf16x8* wp1;         // rb_pack_weights_kernel (scale folded in)
    const bf16x8* wp2;
    const float* shift1;
    const float* shift2;
    int N, H, W;
    int tiles_h, tiles_w, total_units;
    int patch;                 // brick order: patches of patch x patch bricks
    float neg_slope;
    unsigned long long* dbg;   // MVSGI_RS_STAMPS diagnostic build only
};

__device__ __forceinline__ int rb_xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// The epilogues are the kernel's VALU bill (two waves per SIMD share one issue port with the MFMAs), so everything that can
// be is done elsewhere: scale in the weights, shift as the accumulators' start value, the skip connection as ONE more MFMA per
// tile (A = [I | I]: hi + lo of the window's centre pixel, read as one 16-byte fragment while the window is still there),
// stores through a per-brick buffer descriptor with per-launch lane offsets and scalar row offsets, border masks only in bricks that touch the image border.
template <bool OUTF32>
__global__ __launch_bounds__(256, 2) void resblock2d_rs_kernel(RbArgs a) {
    using namespace rb;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kg = lane >> 4;
    const int Hp = a.H + 2 * PAD, Wp = a.W + 2 * PAD;
    const long long total_bytes = (long long)a.N * Hp * Wp * 64;
    const long long out_bytes = OUTF32 ? (long long)a.N * a.H * a.W * 64 : total_bytes;
    const int orow = (OUTF32 ? a.W : Wp) * 64;                       // bytes per output row
    const int total = a.total_units, G = gridDim.x;
    // bricks of this workgroup: logical ids remap(blockIdx) + k * (G / 8) -- XCD x walks a contiguous eighth of the bricks
    const int nmine = (total - (int)blockIdx.x + G - 1) / G;
    const int id0 = G == total ? (int)blockIdx.x : rb_xcd_remap((int)blockIdx.x, total);
    const int idstep = G == total ? 0 : G >> 3;

    // ---- both weight sets, resident ----
    bf16x8 w1h[5], w1l[5], w2h[5], w2l[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        w1h[p] = a.wp1[(p * 2) * 64 + lane];
        w1l[p] = a.wp1[(p * 2 + 1) * 64 + lane];
        w2h[p] = a.wp2[(p * 2) * 64 + lane];
        w2l[p] = a.wp2[(p * 2 + 1) * 64 + lane];
    }
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.shift1 + kg * 4), b2 = *reinterpret_cast<const f32x4*>(a.shift2 + kg * 4);
    // A = [I | I] over k = (hi c 0-15 | lo c 0-15): lane (m, kg) holds k = 8 kg + j, one where (8 kg + j) mod 16 == m
    bf16x8 ident;
#pragma unroll
    for (int j = 0; j < 8; ++j) ident[j] = ((kg & 1) * 8 + j == col) ? (__bf16)1.0f : (__bf16)0.0f;

    // ---- tiles of this wave: column half hf = wave & 1, region rows r0 + 2 k with r0 = wave >> 1: k = 0 .. 7 in conv1 (16 rows),
    //      k = 0 .. 6 in conv2 (the 14 stored rows: no tile is computed for rows 14, 15), every offset between them an immediate.
    //      lane (col, kg) reads chunk kg & 1 of the pixel under tap 2 p + (kg >> 1) ----
    const int hf = wave & 1, r0 = wave >> 1;
    // The conv1-result image swaps the two 16-byte chunks of every second group of four pixels (chunk c of pixel v sits at
    // c ^ ((v >> 2) & 1)): the epilogue's 8-byte writes (16 lanes x 32-byte pitch) are then 2-way instead of 4-way bank
    // conflicts, the fragment reads stay conflict-free.  A wave's tiles are 68 pixels apart, so the swap alternates with the
    // tile index: one address set for even tiles, one for odd.
    int rbp[5], rbq[5], rbqo[5];  // window / conv1-result image (even, odd tiles; its base does not fit the 16-bit immediates)
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int tA = 2 * p, tB = 2 * p + 1 < 9 ? 2 * p + 1 : 2 * p;
        const int t = (kg >> 1) ? tB : tA;
        const int v = (r0 + t / 3) * IW + 16 * hf + col + t % 3;
        rbp[p] = v * 32 + (kg & 1) * 16;
        rbq[p] = IMGA + v * 32 + (((kg & 1) ^ ((v >> 2) & 1)) << 4);
        rbqo[p] = rbq[p] ^ 16;
    }
    // skip connection: window pixel (r + 2, c + 2), 16-byte chunk kg & 1 of the HI (kg < 2) or LO region
    const int rres = ((r0 + 2) * IW + 16 * hf + col + 2) * 32 + (kg & 1) * 16 + (kg >> 1) * REGION;
    // conv1 result -> image B: region pixel (r, c), this lane's channels 4 kg .. 4 kg + 3 (8 bytes of hi, 8 of lo)
    const int wv0 = r0 * IW + 16 * hf + col;
    const int wrb = IMGA + wv0 * 32 + ((((kg >> 1) ^ ((wv0 >> 2) & 1)) << 4) | ((kg & 1) << 3));
    const int wrbo = wrb ^ 16;
    constexpr int TSTEP = 2 * IW * 32;          // bytes between a wave's consecutive tiles (two rows)
    // ---- DMA plan: piece q = wave + 4 m fills LDS bytes [q * 1024, +1024) of the window: 32 pixels x 2 chunks of one region ----
    unsigned voff[DPW];
#pragma unroll
    for (int m = 0; m < DPW; ++m) {
        const int q = wave + 4 * m;
        const int region = q >= PIECES ? 1 : 0, j = q - region * PIECES;
        const int v = 32 * j + (lane >> 1), chunk = lane & 1;
        const int wy = v / IW, wx = v - wy * IW;
        voff[m] = v < NPX ? (unsigned)((wy * Wp + wx) * 64 + region * 32 + chunk * 16) : 0xffffff00u;   // beyond num_records: zeros
    }
    // ---- output: lane offset from the brick's first output pixel, tile k = + 2 k * orow (scalar) ----
    //      split: lanes kg and kg ^ 1 trade halves, kg even stores hi / lo of channels 8 (kg >> 1) .. + 7 (16 bytes)
    const unsigned vst = (unsigned)(r0 * orow + (16 * hf + col) * 64 + (OUTF32 ? kg * 16 : (kg & 1) * 32 + (kg >> 1) * 16));

    // brick order: an image is cut into PATCHES of patch x patch bricks, ids run patch by patch (ragged patches at the right /
    // bottom edges are smaller): the 64 workgroups of an XCD, walking 64 consecutive ids at a time, then work on one compact
    // patch whose inner halos (2 x 4 of 18 rows, 2 x 4 of 34 columns per brick) come from that XCD's L2 instead of HBM again
    // -- in row-major order a brick's vertical neighbours are a round away and the measured fetch was 1.40 x the tensor
    // (1.24 x in patches of 8 x 8).
#define RB_DECODE(ID, N_, OH, OW)                                                        \
    {                                                                                    \
        const int per_img_ = a.tiles_h * a.tiles_w;                                      \
        const int t_ = (ID);                                                             \
        N_ = t_ / per_img_;                                                              \
        const int r_ = t_ - N_ * per_img_;                                               \
        const int rowblk_ = a.patch * a.tiles_w;                                         \
        const int R_ = r_ / rowblk_, r2_ = r_ - R_ * rowblk_;                            \
        const int hR_ = a.tiles_h - a.patch * R_ < a.patch ? a.tiles_h - a.patch * R_ : a.patch; \
        const int colblk_ = hR_ * a.patch;                                               \
        const int C_ = r2_ / colblk_, r3_ = r2_ - C_ * colblk_;                          \
        const int wC_ = a.tiles_w - a.patch * C_ < a.patch ? a.tiles_w - a.patch * C_ : a.patch; \
        const int py_ = r3_ / wC_, px_ = r3_ - py_ * wC_;                                \
        OH = (a.patch * R_ + py_) * TOH;                                                 \
        OW = (a.patch * C_ + px_) * TOW;                                                 \
    }
    // window of brick (n, oh0, ow0): origin = padded pixel (oh0, ow0) = image pixel (oh0 - 2, ow0 - 2)
#define RB_DESC(N_, OH, OW)                                                                                      \
    ({                                                                                                           \
        const long long off_ = (((long long)(N_) * Hp + (OH)) * Wp + (OW)) * 64;                                 \
        const long long left_ = total_bytes - off_;                                                              \
        const int rec_ = left_ > 0x7fffff00ll ? 0x7fffff00 : (int)left_;                                         \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.x) + off_, 0, rec_, 0x00020000);          \
    })
#define RB_PIECE(DSC, M)                                                                                         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(DSC, (__attribute__((address_space(3))) void*)(lds + (wave + 4 * (M)) * 1024), \
                                             16, voff[M], 0, 0, MVSGI_RB_LD_AUX);
#define RB_STAGE(N_, OH, OW)                                                                                     \
    {                                                                                                            \
        const auto dsc_ = RB_DESC(N_, OH, OW);                                                                   \
        _Pragma("unroll") for (int m = 0; m < DPW; ++m) RB_PIECE(dsc_, m)                                        \
    }
// fragments of tiles 4 G .. 4 G + NT - 1 under tap pair P
#define RB_READ(RB, RBO, G, NT, P, BUFI)                                                                         \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < (NT); ++i) {                                                       \
            xh[BUFI][i] = *reinterpret_cast<const bf16x8*>(lds + ((i & 1) ? RBO[P] : RB[P]) + (4 * (G) + i) * TSTEP);          \
            xl[BUFI][i] = *reinterpret_cast<const bf16x8*>(lds + ((i & 1) ? RBO[P] : RB[P]) + (4 * (G) + i) * TSTEP + REGION); \
        }                                                                                                        \
    }
// one convolution = 10 steps (2 groups of tiles x 5 tap pairs) of up to 4 tiles x 3 products, fragments requested one step
// ahead; NT1 = tiles of the second group (4: conv1's 8 tiles, 3: conv2's 7)
#define RB_NOHOOK(ST)
#define RB_CONV(RB, RBO, WH, WL, NT1, HOOK)                                                                           \
    {                                                                                                            \
        bf16x8 xh[2][4], xl[2][4];                                                                               \
        RB_READ(RB, RBO, 0, 4, 0, 0)                                                                               \
        _Pragma("unroll") for (int st = 0; st < 10; ++st) {                                                      \
            const int g = st / 5, p = st % 5, nt = g ? (NT1) : 4;                                                \
            HOOK(st)                                                                                             \
            if (st + 1 < 10) RB_READ(RB, RBO, (st + 1) / 5, ((st + 1) / 5 ? (NT1) : 4), (st + 1) % 5, (st + 1) & 1)   \
            _Pragma("unroll") for (int i = 0; i < nt; ++i)                                                       \
                acc[4 * g + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WL[p], xh[st & 1][i], acc[4 * g + i], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < nt; ++i)                                                       \
                acc[4 * g + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WH[p], xl[st & 1][i], acc[4 * g + i], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < nt; ++i)                                                       \
                acc[4 * g + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WH[p], xh[st & 1][i], acc[4 * g + i], 0, 0, 0); \
        }                                                                                                        \
    }

#ifdef MVSGI_RS_STAMPS   // diagnostic build (tools/rb_stamps.py): s_memtime stamps of workgroup 8, every wave
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
    int n_, oh0, ow0;
    RB_DECODE(id0, n_, oh0, ow0)
    RB_STAGE(n_, oh0, ow0)
    f32x4 acc[8];
    for (int u = 0; u < nmine; ++u) {
        int nn, noh, now;                          // the next brick (clamped: its request is skipped past the end)
        RB_DECODE(id0 + (u + 1 < nmine ? u + 1 : u) * idstep, nn, noh, now)
        // does the conv1 region (image rows oh0 - 1 .. oh0 + 14, columns ow0 - 1 .. ow0 + 30) leave the image?
        const bool edge = oh0 == 0 || ow0 == 0 || oh0 + RH - 1 > a.H || ow0 + RW - 1 > a.W;
        STAMP()
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        STAMP()
        asm volatile("s_barrier" ::: "memory");   // window of brick u landed (every wave's pieces); conv1-result image free
        // ---- phase A: conv1 on the 16 x 32 region, result -> image B (zero outside the image: conv2's padding) ----
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = b1;
        // the epilogue of a tile whose accumulation is complete (group 0: tiles 0 .. 3, after step 4) is written INTO the steps of
        // group 1 (MVSGI_RB_EPI_OVERLAP): its ~25 vector instructions then sit between that step's MFMAs instead of behind the
        // whole convolution
        bf16x8 xres[7];                            // skip connection: window pixel (r + 2, c + 2) as an MFMA operand
#define RB_EPI_A(K)                                                                                              \
        {                                                                                                        \
            f32x4 v = acc[K];                                                                                    \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e], a.neg_slope);                       \
            u32x2 hi, lo;                                                                                        \
            split4(v, hi, lo);                                                                                   \
            if (edge) {                                                                                          \
                const int gh = oh0 - 1 + r0 + 2 * (K), gw = ow0 - 1 + 16 * hf + col;                             \
                if (!(gh >= 0 && gh < a.H && gw >= 0 && gw < a.W)) hi = lo = u32x2{0u, 0u};                      \
            }                                                                                                    \
            *reinterpret_cast<u32x2*>(lds + (((K) & 1) ? wrbo : wrb) + (K) * TSTEP) = hi;                        \
            *reinterpret_cast<u32x2*>(lds + (((K) & 1) ? wrbo : wrb) + (K) * TSTEP + REGION) = lo;               \
            if ((K) < 7) xres[(K) < 7 ? (K) : 0] = *reinterpret_cast<const bf16x8*>(lds + rres + (K) * TSTEP);   \
        }
#if MVSGI_RB_EPI_OVERLAP
#define RB_HOOK_A(ST) if ((ST) >= 5 && (ST) < 9) RB_EPI_A((ST) - 5)
#else
#define RB_HOOK_A(ST)
#endif
        STAMP()
        RB_CONV(rbp, rbp, w1h, w1l, 4, RB_HOOK_A)
        STAMP()
#pragma unroll
        for (int k = MVSGI_RB_EPI_OVERLAP ? 4 : 0; k < 8; ++k) RB_EPI_A(k)
#undef RB_HOOK_A
#undef RB_EPI_A
        STAMP()
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");             // conv1 result complete; every wave is done with the window
        // the window of brick u + 1 lands under conv2.  Its 10 pieces per wave go out one per step of conv2 (MVSGI_RB_DMA_SPREAD):
        // issued in one burst behind the barrier, the workgroup's 40 requests queue up in the CU's address path and every wave
        // sits ~2000 cycles in that segment (tools/rb_stamps.py): 8.95 ms per 32 frames in a burst, 8.70 two per step, 8.66 one
        const bool more = u + 1 < nmine;
        const auto dsc_n = RB_DESC(nn, noh, now);
        // output descriptor: base = the brick's first output pixel; lanes of columns >= 30 (or beyond the image) are masked
        // (an out-of-range voffset is no substitute: the scalar row offset is added before the range check and wraps),
        // rows beyond the image are skipped (wave-uniform)
        const long long ooff_ = OUTF32 ? (((long long)n_ * a.H + oh0) * a.W + ow0) * 64
                                       : (((long long)n_ * Hp + oh0 + PAD) * Wp + ow0 + PAD) * 64;
        const long long oleft_ = out_bytes - ooff_;
        const auto dsc_o = __builtin_amdgcn_make_buffer_rsrc(a.y + ooff_, 0, oleft_ > 0x7fffff00ll ? 0x7fffff00 : (int)oleft_, 0x00020000);
        const bool okc = 16 * hf + col < TOW && ow0 + 16 * hf + col < a.W;
#define RB_EPI_B(K)                                                                                              \
        {                                                                                                        \
            f32x4 v = acc[K];                                                                                    \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e], a.neg_slope);                       \
            u32x4 o;                                                                                             \
            if constexpr (OUTF32) {                                                                              \
                o = __builtin_bit_cast(u32x4, v);                                                                \
            } else {                                                                                             \
                u32x2 hi, lo;                                                                                    \
                split4(v, hi, lo);                                                                               \
                const u32x2 sa = __builtin_amdgcn_permlane16_swap(hi[0], lo[0], false, false);                   \
                const u32x2 sb = __builtin_amdgcn_permlane16_swap(hi[1], lo[1], false, false);                   \
                o = u32x4{sa[0], sb[0], sa[1], sb[1]};                                                           \
            }                                                                                                    \
            /* masked lanes / rows get an offset beyond num_records (no scalar offset here: it would be added before the   */ \
            /* range check and wrap): no exec-mask region, so the store can sit between the MFMAs                          */ \
            const unsigned so_ = (okc && oh0 + r0 + 2 * (K) < a.H) ? vst + (unsigned)(2 * (K) * orow) : 0xffffff00u;      \
            __builtin_amdgcn_raw_buffer_store_b128(o, dsc_o, so_, 0, MVSGI_RB_ST_AUX);                           \
        }
#if !MVSGI_RB_DMA_SPREAD
        if (more) { _Pragma("unroll") for (int m = 0; m < DPW; ++m) RB_PIECE(dsc_n, m) }
#define RB_DMAHOOK(ST)
#elif MVSGI_RB_DMA_SPREAD == 2
#define RB_DMAHOOK(ST)                                                                                           \
            {                                                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                if (more) { RB_PIECE(dsc_n, (ST)) }                                                              \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }
#else
#define RB_DMAHOOK(ST)                                                                                           \
            if ((ST) < DPW / 2) {                                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                if (more) { RB_PIECE(dsc_n, 2 * (ST)) RB_PIECE(dsc_n, 2 * (ST) + 1) }                            \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }
#endif
#if MVSGI_RB_EPI_OVERLAP
#define RB_HOOK_B(ST) RB_DMAHOOK(ST) if ((ST) >= 5 && (ST) < 9) RB_EPI_B((ST) - 5)
#else
#define RB_HOOK_B(ST) RB_DMAHOOK(ST)
#endif
        // ---- phase B: conv2 on the 14 x 30 brick: 7 tiles per wave (columns 30, 31 of the right half are not stored) ----
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ident, xres[k], b2, 0, 0, 0);
        STAMP()
        RB_CONV(rbq, rbqo, w2h, w2l, 3, RB_HOOK_B)
        STAMP()
#pragma unroll
        for (int k = MVSGI_RB_EPI_OVERLAP ? 4 : 0; k < 7; ++k) RB_EPI_B(k)
#undef RB_HOOK_B
#undef RB_DMAHOOK
#undef RB_EPI_B
        n_ = nn; oh0 = noh; ow0 = now;
    }
#undef RB_DECODE
#undef RB_STAGE
#undef RB_DESC
#undef RB_PIECE
#undef RB_NOHOOK
#undef RB_READ
#undef RB_CONV
}

template <bool OUTF32>
int rb_launch(const RbArgs& a, hipStream_t st, const char* what) {
    static mvsgi::PersistentGeom geo_cache[mvsgi::kMaxDevices] = {};
    mvsgi::PersistentGeom geo;
    if (mvsgi::persistent_geometry(resblock2d_rs_kernel<OUTF32>, 256, rb::LDS_BYTES, 2, geo_cache, what, geo)) return 1;
    long long resident = ((long long)geo.cus * geo.wgs_per_cu) / 8 * 8;
    if (resident < 8) resident = 8;
    const long long nb = a.total_units;
#ifdef MVSGI_RS_STAMPS
    RbArgs a2 = a;
    {   // MVSGI_STAMP=1: record; =2: print the stamps of the previous launch
        static unsigned long long* dbgbuf = nullptr;
        const char* e_ = getenv("MVSGI_STAMP");
        if (e_ && !dbgbuf) { (void)hipMalloc(&dbgbuf, 4 * 256 * 8); (void)hipMemset(dbgbuf, 0, 4 * 256 * 8); }
        a2.dbg = e_ ? dbgbuf : nullptr;
        if (e_ && atoi(e_) == 2 && dbgbuf) {
            static unsigned long long h[4 * 256];
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, dbgbuf, sizeof(h), hipMemcpyDeviceToHost);
            for (int w = 0; w < 4; ++w) {
                fprintf(stderr, "rbwave %d:", w);
                for (int i = 0; i < 250; ++i) fprintf(stderr, " %lld", (long long)(h[w * 256 + i] - h[0]));
                fprintf(stderr, "\n");
            }
        }
    }
    hipLaunchKernelGGL(resblock2d_rs_kernel<OUTF32>, dim3((unsigned)(nb <= resident ? nb : resident)), dim3(256), rb::LDS_BYTES, st, a2);
#else
    hipLaunchKernelGGL(resblock2d_rs_kernel<OUTF32>, dim3((unsigned)(nb <= resident ? nb : resident)), dim3(256), rb::LDS_BYTES, st, a);
#endif
    return mvsgi::check_launch(what);
}

// ---------------------------------------------------------------------------------------------
// The extractor's stride-2 layer between its two runs of residual blocks (BaseConvBlk2d 16 -> 16, 3x3, stride 2, padding 1 +
// BatchNorm + LeakyReLU; simple_feature_extractor.py:60-66) on the same format: the 3-D kernel of csrc/conv3d_s2rs.hip in two
// dimensions.  A brick = 8 output rows x 16 columns, its window 17 x 33 input pixels, de-interleaved into even and odd columns by
// the DMA's per-lane source addresses (a tile's 16 outputs then read unit-stride pixels for every tap); double-buffered windows
// (72 KB: two workgroups per CU), one DMA piece in front of each MFMA group; wave w owns output rows 2 w, 2 w + 1; weights in
// rb_pack_weights_kernel's order (scale folded in), shift = the accumulators' start value.  The layer is HBM-bound (reads the
// full-resolution tensor once): with it the last block of the first run hands on split activations instead of fp32.
// ---------------------------------------------------------------------------------------------
namespace c2s {
constexpr int TH = 8, TW = 16;                 // output brick
constexpr int IHt = 2 * TH + 1, IW = 2 * TW + 1;   // window 17 x 33
constexpr int NPX = IHt * IW;                  // 561
constexpr int PIECES = (NPX + 31) / 32;        // 18 per region
constexpr int REGION = PIECES * 1024;
constexpr int IMG = 2 * REGION;                // 36,864
constexpr int LDS_BYTES = 2 * IMG;
constexpr int DPW = 2 * PIECES / 4;            // 9 pieces per wave
static_assert(2 * PIECES % 4 == 0 && 2 * LDS_BYTES <= 160 * 1024, "geometry");
}  // namespace c2s

struct C2sArgs {
    const unsigned char* x;    // [N][H+4][W+4][64]
    unsigned char* y;          // [N][Ho+4][Wo+4][64]
    const bf16x8* wp;
    const float* shift;
    int N, H, W, Ho, Wo;
    int tiles_h, tiles_w, total_units;
    float neg_slope;
};

__device__ __forceinline__ void c2s_dma_piece(const __amdgpu_buffer_rsrc_t dsc, unsigned char* lds_dst, const unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(dsc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void conv2d_s2rs_kernel(C2sArgs a) {
    using namespace c2s;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kg = lane >> 4;
    const int Hp = a.H + 2 * rb::PAD, Wp = a.W + 2 * rb::PAD, Hop = a.Ho + 2 * rb::PAD, Wop = a.Wo + 2 * rb::PAD;
    const long long total_bytes = (long long)a.N * Hp * Wp * 64, ototal_bytes = (long long)a.N * Hop * Wop * 64;
    const int total = a.total_units, G = gridDim.x;
    const int nmine = (total - (int)blockIdx.x + G - 1) / G;
    const int id0 = G == total ? (int)blockIdx.x : rb_xcd_remap((int)blockIdx.x, total);
    const int idstep = G == total ? 0 : G >> 3;

    bf16x8 wh[5], wl[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        wh[p] = a.wp[(p * 2) * 64 + lane];
        wl[p] = a.wp[(p * 2 + 1) * 64 + lane];
    }
    const f32x4 bsh = *reinterpret_cast<const f32x4*>(a.shift + kg * 4);
    // tile i of this wave = output row 2 wave + i; lane (col, kg) reads chunk kg & 1 of window pixel (2 row + kh, 2 col + kw) under
    // tap 2 p + (kg >> 1); a window row holds its 17 even columns, then its 16 odd ones
    int rbp[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int tA = 2 * p, tB = 2 * p + 1 < 9 ? 2 * p + 1 : 2 * p;
        const int t = (kg >> 1) ? tB : tA;
        const int kh = t / 3, kw = t % 3;
        rbp[p] = ((4 * wave + kh) * IW + (kw & 1) * (TW + 1) + col + (kw >> 1)) * 32 + (kg & 1) * 16;
    }
    unsigned voff[DPW];
#pragma unroll
    for (int m = 0; m < DPW; ++m) {
        const int q = wave + 4 * m;
        const int region = q >= PIECES ? 1 : 0, j = q - region * PIECES;
        const int v = 32 * j + (lane >> 1), chunk = lane & 1;
        const int row = v / IW, e = v - row * IW;
        const int c = e <= TW ? 2 * e : 2 * (e - TW - 1) + 1;
        voff[m] = v < NPX ? (unsigned)((row * Wp + c) * 64 + region * 32 + chunk * 16) : 0xffffff00u;
    }
    const unsigned vst = (unsigned)((2 * wave * Wop + col) * 64 + (kg & 1) * 32 + (kg >> 1) * 16);

#define C2S_DECODE(ID, N_, OH, OW)                               \
    {                                                            \
        int t_ = (ID);                                           \
        OW = (t_ % a.tiles_w) * TW;                              \
        t_ /= a.tiles_w;                                         \
        OH = (t_ % a.tiles_h) * TH;                              \
        N_ = t_ / a.tiles_h;                                     \
    }
    // window of brick (n, oh0, ow0): origin = padded input pixel (2 oh0 + 1, 2 ow0 + 1) = image pixel (2 oh0 - 1, 2 ow0 - 1)
#define C2S_DESC(N_, OH, OW)                                                                                     \
    ({                                                                                                           \
        const long long off_ = (((long long)(N_) * Hp + 2 * (OH) + 1) * Wp + 2 * (OW) + 1) * 64;                 \
        const long long left_ = total_bytes - off_;                                                              \
        const int rec_ = left_ > 0x7fffff00ll ? 0x7fffff00 : (int)left_;                                         \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.x) + off_, 0, rec_, 0x00020000);          \
    })
    int n_, oh0, ow0;
    C2S_DECODE(id0, n_, oh0, ow0)
    {
        const auto d0_ = C2S_DESC(n_, oh0, ow0);
#pragma unroll
        for (int m = 0; m < DPW; ++m) c2s_dma_piece(d0_, lds + (wave + 4 * m) * 1024, voff[m]);
    }
    for (int u = 0; u < nmine; ++u) {
        int nn, noh, now;
        C2S_DECODE(id0 + (u + 1 < nmine ? u + 1 : u) * idstep, nn, noh, now)
        const int img = (u & 1) * IMG;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");    // window u landed; every wave is done with window u - 1
        const bool more = u + 1 < nmine;
        const auto dsc_n = C2S_DESC(nn, noh, now);
        f32x4 acc[2], acc1[2], acc2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc[i] = bsh;
            acc1[i] = acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        bf16x8 xh[2][2], xl[2][2];
#define C2S_READ(P, BUFI)                                                                                        \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
            xh[BUFI][i] = *reinterpret_cast<const bf16x8*>(lds + img + rbp[P] + (2 * i * IW) * 32);              \
            xl[BUFI][i] = *reinterpret_cast<const bf16x8*>(lds + img + rbp[P] + (2 * i * IW) * 32 + REGION);     \
        }                                                                                                        \
    }
// the next window's pieces: one in front of each group of MFMAs (three groups per tap pair), never a burst
#define C2S_DMA(K)                                                                                               \
    if ((K) < DPW) {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (more) c2s_dma_piece(dsc_n, lds + (IMG - img) + (wave + 4 * (K)) * 1024, voff[K]);                    \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    }
        C2S_READ(0, 0)
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            if (p + 1 < 5) C2S_READ(p + 1, (p + 1) & 1)
            C2S_DMA(2 * p)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[p], xh[p & 1][i], acc1[i], 0, 0, 0);
            C2S_DMA(2 * p + 1)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[p], xl[p & 1][i], acc2[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[p], xh[p & 1][i], acc[i], 0, 0, 0);
        }
#undef C2S_READ
#undef C2S_DMA
        {
            const long long off_ = (((long long)n_ * Hop + oh0 + rb::PAD) * Wop + ow0 + rb::PAD) * 64;
            const long long left_ = ototal_bytes - off_;
            const int rec_ = left_ > 0x7fffff00ll ? 0x7fffff00 : (int)left_;
            const auto dsc_ = __builtin_amdgcn_make_buffer_rsrc(a.y + off_, 0, rec_, 0x00020000);
            const bool okc = ow0 + col < a.Wo;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 v = acc[i] + (acc1[i] + acc2[i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e], a.neg_slope);
                u32x2 hi, lo;
                split4(v, hi, lo);
                const u32x2 sa = __builtin_amdgcn_permlane16_swap(hi[0], lo[0], false, false);
                const u32x2 sb = __builtin_amdgcn_permlane16_swap(hi[1], lo[1], false, false);
                if (okc && oh0 + 2 * wave + i < a.Ho)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{sa[0], sb[0], sa[1], sb[1]}, dsc_, vst, i * Wop * 64, MVSGI_RB_ST_AUX);
            }
        }
        n_ = nn; oh0 = noh; ow0 = now;
    }
#undef C2S_DECODE
#undef C2S_DESC
}

}  // namespace

// BaseConvBlk2d 16 -> 16, 3x3, stride 2, padding 1 (+ scale folded into w_packed by mvsgi_resblock2d_split_pack_weights, + shift,
// LeakyReLU) on 2-D split-padded activations: x_split [N][H+4][W+4][64 B] -> y_split [N][Ho+4][Wo+4][64 B], Ho = (H - 1) / 2 + 1.
extern "C" int mvsgi_conv2d_s2_split(const void* x_split, const void* w_packed, const float* shift, void* y_split, int N, int H, int W,
                                     float neg_slope, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x_split && w_packed && shift && y_split, "mvsgi_conv2d_s2_split: null pointer");
    MVSGI_REQUIRE(N > 0 && H > 0 && W > 0, "mvsgi_conv2d_s2_split: non-positive dimension");
    MVSGI_REQUIRE(neg_slope >= 0.f && neg_slope <= 1.f, "mvsgi_conv2d_s2_split: negative slope %g outside [0, 1]", (double)neg_slope);
    MVSGI_REQUIRE((long long)(c2s::IHt + 1) * (W + 2 * rb::PAD) * 64 < 0x7fffff00ll, "mvsgi_conv2d_s2_split: rows too long for 32-bit window offsets");
    C2sArgs a{};
    a.x = static_cast<const unsigned char*>(x_split);
    a.y = static_cast<unsigned char*>(y_split);
    a.wp = static_cast<const bf16x8*>(w_packed);
    a.shift = shift;
    a.N = N; a.H = H; a.W = W;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.neg_slope = neg_slope;
    a.tiles_h = (int)mvsgi::cdiv(a.Ho, c2s::TH);
    a.tiles_w = (int)mvsgi::cdiv(a.Wo, c2s::TW);
    const long long nb = (long long)N * a.tiles_h * a.tiles_w;
    MVSGI_REQUIRE(nb < (1ll << 31), "mvsgi_conv2d_s2_split: too many bricks");
    a.total_units = (int)nb;
    static mvsgi::PersistentGeom geo_cache[mvsgi::kMaxDevices] = {};
    mvsgi::PersistentGeom geo;
    if (mvsgi::persistent_geometry(conv2d_s2rs_kernel, 256, c2s::LDS_BYTES, 2, geo_cache, "mvsgi_conv2d_s2_split", geo)) return 1;
    long long resident = ((long long)geo.cus * geo.wgs_per_cu) / 8 * 8;
    if (resident < 8) resident = 8;
    hipLaunchKernelGGL(conv2d_s2rs_kernel, dim3((unsigned)(nb <= resident ? nb : resident)), dim3(256), c2s::LDS_BYTES,
                       mvsgi::as_stream(stream), a);
    return mvsgi::check_launch("mvsgi_conv2d_s2_split");
}

extern "C" size_t mvsgi_split2d_bytes(int N, int H, int W) {
    return (size_t)N * (size_t)(H + 2 * rb::PAD) * (size_t)(W + 2 * rb::PAD) * 64;
}

// fp32 [N][H][W][16] -> 2-D split-padded (interior only: the caller zeroes the buffer once, the border is never written)
extern "C" int mvsgi_f32_to_split2d(const float* x, void* y_split, int N, int H, int W, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x && y_split, "mvsgi_f32_to_split2d: null pointer");
    MVSGI_REQUIRE(N > 0 && H > 0 && W > 0, "mvsgi_f32_to_split2d: non-positive dimension");
    const long long n = (long long)N * H * W * 2;
    MVSGI_REQUIRE(mvsgi::cdiv(n, 256) < (1ll << 31), "mvsgi_f32_to_split2d: tensor too large");
    hipLaunchKernelGGL(f32_to_split2d_kernel, dim3((unsigned)mvsgi::cdiv(n, 256)), dim3(256), 0, mvsgi::as_stream(stream), x,
                       static_cast<unsigned char*>(y_split), N, H, W);
    return mvsgi::check_launch("mvsgi_f32_to_split2d");
}

extern "C" int mvsgi_split2d_to_f32(const void* x_split, float* y, int N, int H, int W, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x_split && y, "mvsgi_split2d_to_f32: null pointer");
    MVSGI_REQUIRE(N > 0 && H > 0 && W > 0, "mvsgi_split2d_to_f32: non-positive dimension");
    const long long n = (long long)N * H * W * 2;
    MVSGI_REQUIRE(mvsgi::cdiv(n, 256) < (1ll << 31), "mvsgi_split2d_to_f32: tensor too large");
    hipLaunchKernelGGL(split2d_to_f32_kernel, dim3((unsigned)mvsgi::cdiv(n, 256)), dim3(256), 0, mvsgi::as_stream(stream),
                       static_cast<const unsigned char*>(x_split), y, N, H, W);
    return mvsgi::check_launch("mvsgi_split2d_to_f32");
}

extern "C" size_t mvsgi_resblock2d_split_packed_weight_bytes(void) { return (size_t)5 * 2 * 64 * 16; }

extern "C" int mvsgi_resblock2d_split_pack_weights(const float* w_oihw, const float* scale, void* w_packed, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(w_oihw && scale && w_packed, "mvsgi_resblock2d_split_pack_weights: null pointer");
    hipLaunchKernelGGL(rb_pack_weights_kernel, dim3(5), dim3(64), 0, mvsgi::as_stream(stream), w_oihw, scale, static_cast<bf16x8*>(w_packed));
    return mvsgi::check_launch("mvsgi_resblock2d_split_pack_weights");
}

// ResConvBlk2d.forward (common/common_modules.py:165-176), 16 -> 16 channels, 3x3, stride 1, on 2-D split-padded activations:
//   y = act( conv2(act(conv1(x) * scale1 + shift1)) * scale2 + shift2 + x )
// (scale1 / scale2 are folded into w_packed1 / w_packed2 by mvsgi_resblock2d_split_pack_weights.)
// x_split [N][H+4][W+4][64 B]; y the same format (y_is_split, border untouched) or plain fp32 [N][H][W][16].
extern "C" int mvsgi_resblock2d_split(const void* x_split, const void* w_packed1, const float* shift1,
                                      const void* w_packed2, const float* shift2, void* y, int y_is_split,
                                      int N, int H, int W, float neg_slope, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x_split && w_packed1 && w_packed2 && shift1 && shift2 && y, "mvsgi_resblock2d_split: null pointer");
    MVSGI_REQUIRE(N > 0 && H > 0 && W > 0, "mvsgi_resblock2d_split: non-positive dimension");
    MVSGI_REQUIRE(neg_slope >= 0.f && neg_slope <= 1.f, "mvsgi_resblock2d_split: negative slope %g outside [0, 1]", (double)neg_slope);
    MVSGI_REQUIRE((long long)(rb::IH + 1) * (W + 2 * rb::PAD) * 64 < 0x7fffff00ll, "mvsgi_resblock2d_split: rows too long for 32-bit window offsets");
    MVSGI_REQUIRE(y_is_split || ((reinterpret_cast<uintptr_t>(y) & 15) == 0), "mvsgi_resblock2d_split: fp32 output must be 16-byte aligned");
    MVSGI_REQUIRE(x_split != y, "mvsgi_resblock2d_split: in-place operation is not supported (bricks read their neighbours' inputs)");
    RbArgs a{};
    a.x = static_cast<const unsigned char*>(x_split);
    a.y = static_cast<unsigned char*>(y);
    a.wp1 = static_cast<const bf16x8*>(w_packed1);
    a.wp2 = static_cast<const bf16x8*>(w_packed2);
    a.shift1 = shift1; a.shift2 = shift2;
    a.N = N; a.H = H; a.W = W; a.neg_slope = neg_slope;
    a.tiles_h = (int)mvsgi::cdiv(H, rb::TOH);
    a.tiles_w = (int)mvsgi::cdiv(W, rb::TOW);
    const long long nb = (long long)N * a.tiles_h * a.tiles_w;
    MVSGI_REQUIRE(nb < (1ll << 31), "mvsgi_resblock2d_split: too many bricks");
    a.total_units = (int)nb;
    static const int patch_env = [] { const char* e = mvsgi::exp_env("MVSGI_RB_PATCH"); return e ? atoi(e) : 0; }();   // experiments
    a.patch = patch_env > 0 ? patch_env : 8;
    hipStream_t st = mvsgi::as_stream(stream);
    return y_is_split ? rb_launch<false>(a, st, "mvsgi_resblock2d_split") : rb_launch<true>(a, st, "mvsgi_resblock2d_split(fp32 out)");
}
