// K2g: the regulator's first stride-2 convolution (UNetDownBlk.first of level 0: BaseConvBlk3d 16 -> 32 channels, 3x3x3,
// stride 2, padding 1 + BatchNorm + LeakyReLU; dsta_mvs/model/cost_volume_regulator/unet_regulator.py:286-303,
// common/common_modules.py:107-115) on pre-split activations.
//
// Why its own kernel.  The layer reads the full-resolution 16-channel volume once (1.68 GB per 64 frames of G16V) and writes
// an eighth of the voxels: 3 flop per byte, HBM-bound by a wide margin (all its MFMAs are 130 us of matrix-core time).  The
// streaming kernel (conv3d_bf16x3.hpp) spends its time in producer waves that fetch fp32 voxels into registers, split them and
// write them to LDS: 707 us, 2.96 TB/s.  Here the input arrives ALREADY SPLIT (post_vol's epilogue, csrc/conv3d_rs.hip, writes
// the split-padded format), so staging is a pure copy on the LDS-DMA path with a whole brick in flight per CU at all times:
//   * a workgroup = 4 waves, persistent, double-buffered windows: the DMA of brick u + 1 is issued before the MFMAs of brick u;
//   * the window of a brick of TH x 16 outputs of one output plane is 3 x (2 TH + 1) x 33 input voxels; the DMA's per-lane source
//     addresses de-interleave it on the way into LDS -- even columns, then odd columns of a row -- so that the 16 outputs of a
//     tile read unit-stride voxels for every tap (kw = 0 | 2: even columns ow, ow + 1; kw = 1: odd column ow);
//   * LDS voxels are 32 B in a HI and 32 B in a LO region; a ds_read_b128 lane group reads 8 voxels' channels 0-7 and 8 OTHER
//     voxels' channels 8-15 of the same tap (k = tap-of-pair x 16 + channel): conflict-free at any pitch;
//   * wave w owns cout tile w & 1 of the brick's output rows 2 (w >> 1), + 1 with that tile's weights (14 tap pairs x hi | lo =
//     112 registers) resident for the whole launch; BatchNorm's scale is folded into the weights, its shift is the
//     accumulators' start value;
//   * the epilogue writes the 32-channel split-padded format the register-stationary 32 -> 32 kernel reads (conv3d_rs.hip).
#include "common.hpp"

#include <cstdlib>

#ifndef MVSGI_S2RS_ST_AUX
#define MVSGI_S2RS_ST_AUX 0     // cache policy bits of the output stores (2 = nt: measured neutral to slower)
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

// LeakyReLU for slopes in [0, 1] as mul + max (plain asm max: __builtin_fmaxf first canonicalises an MFMA result with a third op)
__device__ __forceinline__ float lrelu(const float v, const float slope) {
    const float m = v * slope;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(m));
    return r;
}

// one LDS-DMA piece: 64 lanes x 16 B from per-lane offsets of a buffer into 1 KiB of LDS.  (A plain function on purpose: in the
// kernel TEMPLATE below the builtin's operands would be value-dependent, and hipcc's host pass then drops the kernel's stub
// without a diagnostic.)
__device__ __forceinline__ void s2_dma_piece(const __amdgpu_buffer_rsrc_t dsc, unsigned char* lds_dst, const unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(dsc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t s2_desc(const unsigned char* base, const long long off, const long long total) {
    const long long left = total - off;
    const int rec = left > 0x7fffff00ll ? 0x7fffff00 : (int)left;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base) + off, 0, rec, 0x00020000);
}
__device__ __forceinline__ void s2_store16(const u32x4 v, const __amdgpu_buffer_rsrc_t dsc, const unsigned voff, const int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, dsc, voff, soff, MVSGI_S2RS_ST_AUX);
}
template <bool F16>
__device__ __forceinline__ u32x4 s2_pack_split(const f32x4 v, float& satm) {
    // hi | lo of four channels; lanes kg and kg ^ 1 trade halves: kg even ends up with hi / lo of channels 8 (kg >> 1) .. + 7
    u32x2 hi, lo;
    sf_split4<F16>(v, hi, lo, satm);
    const u32x2 sa = __builtin_amdgcn_permlane16_swap(hi[0], lo[0], false, false);
    const u32x2 sb = __builtin_amdgcn_permlane16_swap(hi[1], lo[1], false, false);
    return u32x4{sa[0], sb[0], sa[1], sb[1]};
}
template <bool F16>
__device__ __forceinline__ f32x4 s2_mfma(const bf16x8 a, const bf16x8 b, const f32x4 c) {
    return sf_mfma16<F16>(a, b, c);
}

#ifndef MVSGI_S2RS_DMA_SPREAD
#define MVSGI_S2RS_DMA_SPREAD 1
#endif

namespace s2 {
constexpr int TW = 16;                         // outputs per tile (one row)
constexpr int IW = 2 * TW + 1;                 // 33 input columns: 17 even, then 16 odd, per LDS row
constexpr int kPairs = 14;                     // 27 taps in pairs (the 28th slot: zero weights)
template <int TH>
struct Geo {
    static constexpr int IHt = 2 * TH + 1;                 // input rows of a brick
    static constexpr int NPX = 3 * IHt * IW;               // voxels of a window
    static constexpr int PIECES = (NPX + 31) / 32;         // DMA pieces of 32 voxels x 32 B per region
    static constexpr int DPW = 2 * PIECES / 4;             // pieces per wave (HI and LO regions)
    static constexpr int REGION = PIECES * 1024;           // HI region, the LO region right behind
    static constexpr int IMG = 2 * REGION;
    static_assert(2 * PIECES % 4 == 0, "whole pieces per wave");
};
}  // namespace s2

// [Cout 32][Cin 16][27] x scale[Cout] -> [cout tile 2][14 pairs][hi | lo][64 lanes][8 bf16]
//   lane = (kg << 4) | i holds scale[co] * W[co = 16 ct + i][cin = (kg & 1) * 8 + j][tap = 2 p + (kg >> 1)]  (tap 27: zeros)
__global__ void s2rs_pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale, bf16x8* __restrict__ wp, bool f16) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= 2 * s2::kPairs * 64) return;
    const int lane = idx & 63, r = idx >> 6;
    const int p = r % s2::kPairs, ct = r / s2::kPairs;
    const int kg = lane >> 4, co = ct * 16 + (lane & 15), ci = (kg & 1) * 8, tap = 2 * p + (kg >> 1);
    u16x8 hi, lo;
    for (int j = 0; j < 8; ++j) {
        const float v = tap < 27 ? w[((long long)co * 16 + ci + j) * 27 + tap] * scale[co] : 0.f;
        unsigned short h_, l_;
        sf_split_weight(v, f16, h_, l_);
        hi[j] = h_;
        lo[j] = l_;
    }
    wp[((ct * s2::kPairs + p) * 2) * 64 + lane] = __builtin_bit_cast(bf16x8, hi);
    wp[((ct * s2::kPairs + p) * 2 + 1) * 64 + lane] = __builtin_bit_cast(bf16x8, lo);
}

struct S2Args {
    const unsigned char* x;    // split-padded [B][D+2][H+2][W+2][64 B] (16 channels)
    unsigned char* y;          // split-padded [B][Do+2][Ho+2][Wo+2][128 B] (32 channels)
    const bf16x8* wp;
    const float* shift;        // [32]
    int B, D, H, W, Do, Ho, Wo;
    int tiles_h, tiles_w, total_units;
    float neg_slope;
    float unscale;             // fp16 split: the packed weights and `shift` carry a power of two 1 / unscale (so that the weights' lo parts
                               // are normal fp16 numbers); the accumulators are multiplied by it in front of the activation
    unsigned* sat;             // the range report's words (csrc/api.cpp)
};

__device__ __forceinline__ int s2_xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// NBUF = 2: one workgroup per CU (TH = 4), the window of brick u + 1 in flight under brick u.  NBUF = 1: ONE window per workgroup and
// two workgroups per CU -- the next window is requested when the workgroup is done with the current one, and the partner workgroup's
// request is in flight meanwhile (~1.6 windows in flight per CU instead of 1).  Measured the same within 2 % (565 vs 577 us per 64
// frames): the kernel is not short of requests in flight; NBUF = 2 is the default, MVSGI_S2RS_NBUF=1 selects the other.
// O32P: the output is "fp32-padded" -- the split-padded geometry with plain fp32 records -- for a Winograd-form level 0 behind it
// (csrc/conv3d_wino.hip: its layers split their operands behind the transform, a pre-split input buys them nothing)
template <int TH, int NBUF, bool F16 = false, bool O32P = false>
__global__ __launch_bounds__(256, (TH == 4 && NBUF == 2) ? 1 : 2) void conv3d_s2rs_kernel(S2Args a) {
    using namespace s2;
    using G = Geo<TH>;
    constexpr int IHt = G::IHt, NPX = G::NPX, PIECES = G::PIECES, DPW = G::DPW, REGION = G::REGION, IMG = G::IMG;
    constexpr int TPW = TH / 2;                       // tiles (output rows) per wave: waves 2 q, 2 q + 1 share rows q TPW ..
    static_assert(TH % 2 == 0, "two waves per pair of cout tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave & 1, t0 = (wave >> 1) * TPW;
    const int col = lane & 15, kg = lane >> 4;
    const int Hp = a.H + 2, Wp = a.W + 2, Hop = a.Ho + 2, Wop = a.Wo + 2;
    const long long frame_bytes = (long long)(a.D + 2) * Hp * Wp * 64, total_bytes = frame_bytes * a.B;
    const long long oframe_bytes = (long long)(a.Do + 2) * Hop * Wop * 128, ototal_bytes = oframe_bytes * a.B;
    const int total = a.total_units, Gd = gridDim.x;
    const int nmine = (total - (int)blockIdx.x + Gd - 1) / Gd;
    const int id0 = Gd == total ? (int)blockIdx.x : s2_xcd_remap((int)blockIdx.x, total);
    const int idstep = Gd == total ? 0 : Gd >> 3;

    // ---- this wave's weights (cout tile ct), resident ----
    bf16x8 wh[kPairs], wl[kPairs];
#pragma unroll
    for (int p = 0; p < kPairs; ++p) {
        wh[p] = a.wp[((ct * kPairs + p) * 2) * 64 + lane];
        wl[p] = a.wp[((ct * kPairs + p) * 2 + 1) * 64 + lane];
    }
    const f32x4 bsh = *reinterpret_cast<const f32x4*>(a.shift + ct * 16 + kg * 4);

    // ---- fragment read addresses (window 0, HI region): tile i of this wave = output row t0 + i, outputs ow = col;
    //      lane (col, kg) reads chunk kg & 1 of voxel (plane kd, row 2 (t0 + i) + kh, column 2 col + kw) under tap 2 p + (kg >> 1) ----
    int rbp[kPairs];
#pragma unroll
    for (int p = 0; p < kPairs; ++p) {
        const int tA = 2 * p, tB = 2 * p + 1 < 27 ? 2 * p + 1 : 2 * p;
        const int t = (kg >> 1) ? tB : tA;
        const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
        rbp[p] = (((kd * IHt + 2 * t0 + kh) * IW) + (kw & 1) * (TW + 1) + col + (kw >> 1)) * 32 + (kg & 1) * 16;
    }
    // ---- DMA plan: piece q = wave + 4 m fills LDS bytes [q * 1024, +1024) of a window: 32 voxels x 2 chunks of one region;
    //      LDS voxel index = (plane * IHt + row) * 33 + (column odd ? 17 + column / 2 : column / 2) ----
    unsigned voff[DPW];
#pragma unroll
    for (int m = 0; m < DPW; ++m) {
        const int q = wave + 4 * m;
        const int region = q >= PIECES ? 1 : 0, j = q - region * PIECES;
        const int v = 32 * j + (lane >> 1), chunk = lane & 1;
        const int pr = v / IW, e = v - pr * IW;               // (plane * IHt + row), slot in the row
        const int pl = pr / IHt, row = pr - pl * IHt;
        const int c = e <= TW ? 2 * e : 2 * (e - TW - 1) + 1;
        voff[m] = v < NPX ? (unsigned)(((pl * Hp + row) * Wp + c) * 64 + region * 32 + chunk * 16) : 0xffffff00u;
    }
    // ---- output: this lane's 16 bytes of voxel (row t0 + i, ow = col) of slice ct: lanes kg and kg ^ 1 trade halves ----
    unsigned vst = O32P ? (unsigned)((t0 * Wop + col) * 128 + ct * 64 + kg * 16)
                        : (unsigned)((t0 * Wop + col) * 128 + ct * 64 + (kg & 1) * 32 + (kg >> 1) * 16);

    // brick order (b, oh, od, ow), ow fastest: the bricks stacked along D share one of their three input planes and follow each
    // other within one XCD round
#define S2_DECODE(ID, B_, OD, OH, OW)                            \
    {                                                            \
        int t_ = (ID);                                           \
        OW = (t_ % a.tiles_w) * TW;                              \
        t_ /= a.tiles_w;                                         \
        OD = t_ % a.Do;                                          \
        t_ /= a.Do;                                              \
        OH = (t_ % a.tiles_h) * TH;                              \
        B_ = t_ / a.tiles_h;                                     \
    }
    // window of brick (b, od, oh0, ow0): origin = padded input voxel (2 od, 2 oh0, 2 ow0)
#define S2_STAGE(IMGOFF, B_, OD, OH, OW)                                                                         \
    {                                                                                                            \
        const auto dsc_ = s2_desc(a.x, (long long)(B_) * frame_bytes + (((long long)(2 * (OD)) * Hp + 2 * (OH)) * Wp + 2 * (OW)) * 64, total_bytes); \
        _Pragma("unroll") for (int m = 0; m < DPW; ++m) s2_dma_piece(dsc_, lds + (IMGOFF) + (wave + 4 * m) * 1024, voff[m]); \
    }
    int b_, od, oh0, ow0;
    S2_DECODE(id0, b_, od, oh0, ow0)
    S2_STAGE(0, b_, od, oh0, ow0)
    float satm = 0.f;          // fp16 split: running maximum |value written| (range report)
    for (int u = 0; u < nmine; ++u) {
        int nb, nod, noh, now;
        S2_DECODE(id0 + (u + 1 < nmine ? u + 1 : u) * idstep, nb, nod, noh, now)
        const int img = NBUF == 2 ? (u & 1) * IMG : 0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");    // window u landed; every wave is done with window u - 1
        // NBUF == 2: the window of brick u + 1 lands under this brick's MFMAs and the next wait.  Its pieces go out one per tap
        // pair (DPW == kPairs for TH == 4) instead of in a burst behind the barrier, where the workgroup's 56 requests queue up in
        // the CU's address path in front of the MFMAs (MVSGI_S2RS_DMA_SPREAD=0 restores the burst: 559 vs 486 us per 64 frames).
        // Four different issue points inside a pair for the four waves instead of one: 501 vs 493 us -- not kept
        const bool more = u + 1 < nmine;
        const auto dsc_n = s2_desc(a.x, (long long)nb * frame_bytes + (((long long)(2 * nod) * Hp + 2 * noh) * Wp + 2 * now) * 64, total_bytes);
        constexpr bool SPREAD = NBUF == 2 && MVSGI_S2RS_DMA_SPREAD && DPW <= kPairs;
        if constexpr (NBUF == 2 && !SPREAD)
            if (more) { _Pragma("unroll") for (int m = 0; m < DPW; ++m) s2_dma_piece(dsc_n, lds + (IMG - img) + (wave + 4 * m) * 1024, voff[m]); }
        // three accumulators per tile, one per product term: a tile's consecutive MFMAs then never wait for each other (with one
        // accumulator per tile a wave has two dependent chains and the matrix pipe idles half of the time)
        f32x4 acc[TPW], acc1[TPW], acc2[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            acc[i] = bsh;
            acc1[i] = acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        bf16x8 xh[2][TPW], xl[2][TPW];
#define S2_READ(P, BUFI)                                                                                         \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < TPW; ++i) {                                                        \
            xh[BUFI][i] = *reinterpret_cast<const bf16x8*>(lds + img + rbp[P] + (2 * i * IW) * 32);              \
            xl[BUFI][i] = *reinterpret_cast<const bf16x8*>(lds + img + rbp[P] + (2 * i * IW) * 32 + REGION);     \
        }                                                                                                        \
    }
        S2_READ(0, 0)
#pragma unroll
        for (int p = 0; p < kPairs; ++p) {
            if constexpr (SPREAD) {
                if (p < DPW) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) s2_dma_piece(dsc_n, lds + (IMG - img) + (wave + 4 * p) * 1024, voff[p]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (p + 1 < kPairs) S2_READ(p + 1, (p + 1) & 1)
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc1[i] = s2_mfma<F16>(wl[p], xh[p & 1][i], acc1[i]);
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc2[i] = s2_mfma<F16>(wh[p], xl[p & 1][i], acc2[i]);
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = s2_mfma<F16>(wh[p], xh[p & 1][i], acc[i]);
        }
#undef S2_READ
        if constexpr (NBUF == 1) {                       // every wave has its fragments: the window is free for brick u + 1
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (u + 1 < nmine) S2_STAGE(0, nb, nod, noh, now)
        }
        {
            const auto dsc_ = s2_desc(a.y, (long long)b_ * oframe_bytes + (((long long)(od + 1) * Hop + oh0 + 1) * Wop + ow0 + 1) * 128, ototal_bytes);
            const bool okc = ow0 + col < a.Wo;
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                f32x4 v = acc[i] + (acc1[i] + acc2[i]);          // the small terms first
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = lrelu(F16 ? v[e] * a.unscale : v[e], a.neg_slope);
                if constexpr (O32P) {       // the fp32-padded format's range: +-16376 (a Winograd layer sums four of these in front of its fp16 split)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -16376.f, 16376.f);
                    satm = sf_sat_acc(sf_sat_acc(satm, v[0], v[1]), v[2], v[3]);
                }
                u32x4 o;
                if constexpr (O32P) o = __builtin_bit_cast(u32x4, v);
                else o = s2_pack_split<F16>(v, satm);
                if (okc && oh0 + t0 + i < a.Ho) s2_store16(o, dsc_, vst, i * Wop * 128);
            }
        }
        b_ = nb; od = nod; oh0 = noh; ow0 = now;
    }
    if constexpr (F16) sf_sat_report(a.sat, O32P ? kSatWino : kSatSplit, satm, O32P ? 16376.f : kF16Max);
#undef S2_DECODE
#undef S2_STAGE
}

template <int TH, int NBUF, bool F16 = false, bool O32P = false>
int s2_launch(S2Args a, hipStream_t st) {
    constexpr int lds_bytes = NBUF * s2::Geo<TH>::IMG;
    constexpr int wgs = (TH == 4 && NBUF == 2) ? 1 : 2;
    static_assert(wgs * lds_bytes <= 160 * 1024, "LDS budget");
    a.tiles_h = (int)mvsgi::cdiv(a.Ho, TH);
    a.tiles_w = (int)mvsgi::cdiv(a.Wo, s2::TW);
    const long long nb = (long long)a.B * a.Do * a.tiles_h * a.tiles_w;
    MVSGI_REQUIRE(nb < (1ll << 31), "mvsgi_conv3d_s2rs: too many bricks");
    a.total_units = (int)nb;
    MVSGI_SAT_WORDS(sat_words_);
    a.sat = sat_words_;
    static mvsgi::PersistentGeom geo_cache[mvsgi::kMaxDevices] = {};
    mvsgi::PersistentGeom geo;
    if (mvsgi::persistent_geometry(conv3d_s2rs_kernel<TH, NBUF, F16, O32P>, 256, lds_bytes, wgs, geo_cache, "mvsgi_conv3d_s2rs", geo)) return 1;
    long long resident = ((long long)geo.cus * geo.wgs_per_cu) / 8 * 8;
    if (resident < 8) resident = 8;
    hipLaunchKernelGGL((conv3d_s2rs_kernel<TH, NBUF, F16, O32P>), dim3((unsigned)(nb <= resident ? nb : resident)), dim3(256), lds_bytes, st, a);
    return mvsgi::check_launch("mvsgi_conv3d_s2rs");
}

}  // namespace

extern "C" size_t mvsgi_conv3d_s2rs_packed_weight_bytes(void) { return (size_t)2 * s2::kPairs * 2 * 64 * 16; }

extern "C" int mvsgi_conv3d_s2rs_pack_weights_fmt(const float* w_oidhw, const float* scale, void* w_packed, int fmt, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(w_oidhw && scale && w_packed, "mvsgi_conv3d_s2rs_pack_weights: null pointer");
    MVSGI_REQUIRE(fmt == 0 || fmt == MVSGI_SPLIT_F16, "mvsgi_conv3d_s2rs_pack_weights: fmt %d not in {0, MVSGI_SPLIT_F16}", fmt);
    hipLaunchKernelGGL(s2rs_pack_weights_kernel, dim3(2 * s2::kPairs), dim3(64), 0, mvsgi::as_stream(stream), w_oidhw, scale,
                       static_cast<bf16x8*>(w_packed), fmt != 0);
    return mvsgi::check_launch("mvsgi_conv3d_s2rs_pack_weights");
}
extern "C" int mvsgi_conv3d_s2rs_pack_weights(const float* w_oidhw, const float* scale, void* w_packed, mvsgi_stream_t stream) {
    return mvsgi_conv3d_s2rs_pack_weights_fmt(w_oidhw, scale, w_packed, 0, stream);
}

// y = act( conv3d(x, w, stride 2, padding 1) * scale + shift ), 16 -> 32 channels, on split-padded activations:
// x_split [B][D+2][H+2][W+2][64 B], y_split [B][Do+2][Ho+2][Wo+2][128 B] with Do = (D - 1) / 2 + 1 (likewise Ho, Wo); both
// zero-bordered, only y's interior is written.  w_packed from mvsgi_conv3d_s2rs_pack_weights (scale folded in).
// _fmt: fmt = MVSGI_SPLIT_F16 runs the layer in the fp16 split; `scale` (at packing) and `shift` then carry a power of two 1 / unscale
// chosen by the caller (the epilogue has no per-channel multiplier to hide it in), and the accumulators are multiplied by `unscale`
// _out_fmt: y_f32p != 0 (fp16 split only) writes y "fp32-padded": the same padded geometry, plain fp32 records
extern "C" int mvsgi_conv3d_s2rs_out_fmt(const void* x_split, const void* w_packed, const float* shift, void* y_split, int B, int D, int H,
                                         int W, float neg_slope, float unscale, int fmt, int y_f32p, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(x_split && w_packed && shift && y_split, "mvsgi_conv3d_s2rs: null pointer");
    MVSGI_REQUIRE(!y_f32p || fmt == MVSGI_SPLIT_F16, "mvsgi_conv3d_s2rs: the fp32-padded output exists in the fp16 split only");
    MVSGI_REQUIRE(fmt == 0 || fmt == MVSGI_SPLIT_F16, "mvsgi_conv3d_s2rs: fmt %d not in {0, MVSGI_SPLIT_F16}", fmt);
    MVSGI_REQUIRE(fmt != 0 || unscale == 1.f, "mvsgi_conv3d_s2rs: unscale is a parameter of the fp16 split");
    MVSGI_REQUIRE(unscale > 0.f, "mvsgi_conv3d_s2rs: unscale must be positive");
    MVSGI_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "mvsgi_conv3d_s2rs: non-positive dimension");
    MVSGI_REQUIRE(neg_slope >= 0.f && neg_slope <= 1.f, "mvsgi_conv3d_s2rs: negative slope %g outside [0, 1]", (double)neg_slope);
    MVSGI_REQUIRE((long long)(D + 2) * (H + 2) * (W + 2) * 64 < 0x7fffff00ll, "mvsgi_conv3d_s2rs: input frame too large for 32-bit window offsets");
    S2Args a{};
    a.x = static_cast<const unsigned char*>(x_split);
    a.y = static_cast<unsigned char*>(y_split);
    a.wp = static_cast<const bf16x8*>(w_packed);
    a.shift = shift;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.Do = (D - 1) / 2 + 1; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    MVSGI_REQUIRE((long long)(a.Do + 2) * (a.Ho + 2) * (a.Wo + 2) * 128 < 0x7fffff00ll, "mvsgi_conv3d_s2rs: output frame too large for 32-bit offsets");
    a.neg_slope = neg_slope;
    a.unscale = unscale;
    hipStream_t st = mvsgi::as_stream(stream);
    if (fmt && y_f32p) return s2_launch<4, 2, true, true>(a, st);
    if (fmt) return s2_launch<4, 2, true>(a, st);
#ifdef MVSGI_EXPERIMENTAL      // measured equal or slower (DESIGN_HISTORY.md): 2-row bricks with two workgroups per CU, single-window workgroups
    const char* th_e = mvsgi::exp_env("MVSGI_S2RS_TH");          // brick height: 4 output rows (one workgroup per CU) or 2 (two)
    const char* nb_e = mvsgi::exp_env("MVSGI_S2RS_NBUF");        // windows per workgroup: 1 (two workgroups per CU) or 2 (TH = 4: one)
    if (th_e && atoi(th_e) == 2) return s2_launch<2, 2>(a, st);
    if (nb_e && atoi(nb_e) == 1) return s2_launch<4, 1>(a, st);
#endif
    return s2_launch<4, 2>(a, st);
}
extern "C" int mvsgi_conv3d_s2rs_fmt(const void* x_split, const void* w_packed, const float* shift, void* y_split, int B, int D, int H,
                                     int W, float neg_slope, float unscale, int fmt, mvsgi_stream_t stream) {
    return mvsgi_conv3d_s2rs_out_fmt(x_split, w_packed, shift, y_split, B, D, H, W, neg_slope, unscale, fmt, 0, stream);
}
extern "C" int mvsgi_conv3d_s2rs(const void* x_split, const void* w_packed, const float* shift, void* y_split, int B, int D, int H,
                                 int W, float neg_slope, mvsgi_stream_t stream) {
    return mvsgi_conv3d_s2rs_fmt(x_split, w_packed, shift, y_split, B, D, H, W, neg_slope, 1.f, 0, stream);
}
