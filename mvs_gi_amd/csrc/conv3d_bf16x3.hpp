// Split-bf16 ("bf16x3") implicit-GEMM 3x3x3 convolution on v_mfma_f32_16x16x32_bf16.
// Included by conv3d.hip (shares ConvArgs / f32x4 / kVS with the exact-fp32 kernels).
//
// Arithmetic: every fp32 operand is split as x = hi + lo, hi = bf16(x), lo = bf16(x - hi)
// (16 significant bits); a product is hi*hi + hi*lo + lo*hi with fp32 accumulation: 3 bf16
// MFMAs at 16x the fp32-MFMA rate = 5.3x the exact kernel, ~2^-16 relative error per product.
//
// Mapping (D[cout 16][voxel 16] += W[cout 16][k 32] * X[k 32][voxel 16]):
//   * one MFMA's K = 32 is a PAIR of taps x 16 channels: lane = (kg << 4) | col,
//     kg & 1 selects the tap of the pair, kg >> 1 the 8-channel half.  27 taps -> 14 pairs
//     (the last half empty: zero weights, 3.6 % padding work);
//   * a 16-voxel tile is 16 consecutive W positions of the brick (TW == 16), so the 16 lanes a
//     ds_read_b128 services together (8 lanes of one kg, 8 of its tap-partner) address 15
//     different consecutive voxels + 1 shared one at an 80-byte stride: bank-conflict free for the
//     in-row pairs (delta 1) and the row-wrap pairs (delta ITW - 2 = 16), see DESIGN.md;
//   * WAVE SPECIALISATION: a workgroup is 4 consumer waves (MFMA) + 4 producer waves.  Vector
//     memory loads return in issue order, so a wave that mixes slow activation loads (HBM /
//     Infinity Cache) with the fast weight-fragment loads (L1/L2) stalls every fragment behind
//     the slowest activation load in front of it.  Here the consumers' queues hold weight
//     fragments only; the producers fetch the NEXT unit's halo brick (one 16-channel slice, fp32
//     in HBM), split it and write it to the other half of a double-buffered LDS image
//     [hi0-7 | hi8-15 | lo0-7 | lo8-15 | pad] = 80 B per voxel, while the consumers multiply the
//     current one.  One __syncthreads() per unit hands the buffers over.  The producers' requests run two
//     units ahead (two register sets) and are issued item by item between the split / LDS writes of
//     the previous unit (a burst would queue in front of the consumers' weight fragments);
//   * weights arrive pre-split and lane-ordered straight from L2 (one coalesced 16-byte load per
//     lane per fragment), requested 3 slots ahead of their MFMAs into rotating registers;
//   * template flags select further schedules of the same arithmetic: UPS (trilinear x2 upsample evaluated
//     in the producers), PLANE (Cout == 16), V32 (32x32x16 MFMA) -- described at the kernel;
//   * workgroups are persistent (one per CU for the 4x4x16 bricks) and walk (brick, cout-block) units with
//     stride gridDim.x; the unit index space is re-mapped so that each XCD (= each private L2)
//     owns a contiguous run of bricks: neighbouring halos and the cout-blocks of a brick share L2.
#pragma once
#ifndef MVSGI_ABL
#define MVSGI_ABL 0   // diagnostic builds: 1 no weight loads, 2 no LDS fragment reads, 4 no split, 8 no staging, 16 no MFMA, 32 no 14th pair
#endif

#include "split_fmt.hpp"      // the range report of the fp16 split (sf_sat_acc / sf_sat_report)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int kVSB = 80;      // LDS bytes per staged voxel

// x = hi + lo for four fp32 values, hi = bf16(x) (RNE), lo = bf16(x - hi), packed two per dword:
// 2 x (v_cvt_pk_bf16_f32, shift, and, 2 v_sub_f32, v_cvt_pk_bf16_f32) VALU instructions (converting element by element
// cost 16).  The subtraction and the blends below are SCALAR fp32 on purpose: beside the consumers' MFMAs the packed forms
// (v_pk_add_f32 / v_pk_fma_f32, -DMVSGI_PK) are slower -- out_costs.0 2683 vs 2538 us, down.0.first 739 vs 712 us per 64 frames,
// the step 4689 vs 4742 frames/s (tools/ab_bench.py, one box, 3 rounds; MI355X_MICROARCH.md's packed-VALU-beside-MFMA note)
__device__ __forceinline__ void split_bf16x4(const f32x4 x, u32x2& hi, u32x2& lo) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2v v = {x[2 * p], x[2 * p + 1]};
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        const f32x2v hf = {__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
        hi[p] = hb;
#ifndef MVSGI_PK
        float r0 = v[0] - hf[0], r1 = v[1] - hf[1];
        asm volatile("" : "+v"(r0));
        asm volatile("" : "+v"(r1));
        lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{r0, r1}, bf16x2));
#else
        lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, bf16x2));
#endif
    }
}
// a * b + c on four lanes' worth of channels: four v_fma_f32 (two v_pk_fma_f32 in the -DMVSGI_PK diagnostic build, see above)
__device__ __forceinline__ f32x4 fma4(const f32x4 a, const f32x4 b, const f32x4 c) {
#ifndef MVSGI_PK
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = __builtin_fmaf(a[e], b[e], c[e]);
        asm volatile("" : "+v"(t));            // keeps the four apart
        r[e] = t;
    }
    return r;
#else
    return __builtin_elementwise_fma(a, b, c);
#endif
}
// ---- the second 16-bit split: fp16 ("f16x3") ----
// hi = fp16(x), lo = fp16(x - hi): 11 + 11 significant bits per operand instead of bf16's 8 + 8, the same three products at the same
// matrix rate (v_mfma_f32_16x16x32_f16), fp32 accumulation.  The split's error IS the arithmetic's error (tools/split_arith_emulation.py
// reproduces the bf16x3 path's inverse-distance error on the CPU from the split alone): ~2^-21 per operand against 2^-17 -- inverse
// distance 10x closer to the reference on a sharp softmax.  The price is fp16's range: an operand is CLAMPED to +-65504 (one
// v_med3_f32; fp32's range in the bf16 split), and lo parts below 2^-14 are fp16 subnormals (honoured by the matrix cores of gfx950,
// tools/ubench/mfma_f16_denorm.hip; absolute quantum 2^-24 -- why the host pre-scales the weights by a power of two per output
// channel, undone in the epilogue's per-channel scale).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_f16x4(const f32x4 x, u32x2& hi, u32x2& lo) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2v v = {__builtin_amdgcn_fmed3f(x[2 * p], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(x[2 * p + 1], -65504.f, 65504.f)};
        const f16x2v h = __builtin_convertvector(v, f16x2v);               // v_cvt_pk_f16_f32 (RNE)
        hi[p] = __builtin_bit_cast(unsigned, h);
        float r0 = v[0] - (float)h[0], r1 = v[1] - (float)h[1];            // exact in fp32
        asm volatile("" : "+v"(r0));
        asm volatile("" : "+v"(r1));
        lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{r0, r1}, f16x2v));
    }
}
template <bool F16>
__device__ __forceinline__ void split_x4(const f32x4 x, u32x2& hi, u32x2& lo) {
    if constexpr (F16) split_f16x4(x, hi, lo); else split_bf16x4(x, hi, lo);
}
// the same, keeping the lane's running maximum |clamped value| for the range report of the fp16 split (csrc/split_fmt.hpp)
template <bool F16>
__device__ __forceinline__ void split_x4(const f32x4 x, u32x2& hi, u32x2& lo, float& satm) {
    if constexpr (F16) {
        const f32x4 c = {__builtin_amdgcn_fmed3f(x[0], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(x[1], -65504.f, 65504.f),
                         __builtin_amdgcn_fmed3f(x[2], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(x[3], -65504.f, 65504.f)};
        satm = sf_sat_acc(sf_sat_acc(satm, c[0], c[1]), c[2], c[3]);
        split_f16x4(c, hi, lo);          // (its own clamp of the clamped values folds away)
    } else split_bf16x4(x, hi, lo);
}
// one matrix instruction of the split product: operands travel as 16-byte fragments whatever their element type
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const bf16x8 a, const bf16x8 b, const f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
typedef float f32x16v __attribute__((ext_vector_type(16)));
template <bool F16>
__device__ __forceinline__ f32x16v mfma32(const bf16x8 a, const bf16x8 b, const f32x16v c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// host / pack kernels: one fp32 weight -> (hi, lo) of either split, as raw 16-bit patterns
__device__ __forceinline__ void split_weight(float v, bool f16, unsigned short& hi, unsigned short& lo) {
    if (f16) {
        const _Float16 h = (_Float16)v;
        hi = __builtin_bit_cast(unsigned short, h);
        lo = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)h));
    } else {
        const __bf16 h = (__bf16)v;
        hi = __builtin_bit_cast(unsigned short, h);
        lo = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
    }
}
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int pairs_of(int kd) { return (kd * 9 + 1) / 2; }      // KD = 3: 14 pairs of 27 taps; KD = 1 (2-D 3x3): 5 of 9

// [Cout][Cin][taps] -> [Cin/16][Cout/16][pairs][hi|lo][64 lanes][8 bf16]   (taps = 27 or 9)
//   lane = (kg << 4) | i holds W[cout = ct*16+i][cin = cc*16 + (kg>>1)*8 + j][tap = 2p + (kg&1)]
__global__ void pack_weights_bf16x3_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cout, int Cin,
                                           int taps, bool f16 = false) {
    const int kPairs = (taps + 1) / 2;
    const int CT = Cout / 16;
    const long long total = (long long)(Cin / 16) * kPairs * CT * 64;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int ct = (int)(r % CT);
    r /= CT;
    const int p = (int)(r % kPairs);
    const int cc = (int)(r / kPairs);
    const int kg = lane >> 4;
    const int co = ct * 16 + (lane & 15);
    const int ci = cc * 16 + (kg >> 1) * 8;
    const int tap = 2 * p + (kg & 1);
    u16x8 hi, lo;
    for (int j = 0; j < 8; ++j) {
        const float v = tap < taps ? w[((long long)co * Cin + ci + j) * taps + tap] : 0.f;
        unsigned short h_, l_;
        split_weight(v, f16, h_, l_);
        hi[j] = h_;
        lo[j] = l_;
    }
    const long long o = ((((long long)cc * CT + ct) * kPairs + p) * 2) * 64 + lane;
    wp[o] = __builtin_bit_cast(bf16x8, hi);
    wp[o + 64] = __builtin_bit_cast(bf16x8, lo);
}

// 32-channel slices (D32): [Cout][Cin][27] -> [Cin/32][Cout/16][27 taps][hi|lo][64 lanes][8 bf16]
//   lane = (kg << 4) | i holds W[cout = ct*16 + i][cin = (2*cc + (kg & 1)) * 16 + (kg >> 1) * 8 + j][tap]: the K = 32 of an MFMA is ONE tap
//   of two 16-channel slices (the lane groups that took the pair's second tap take the second slice) -- 27 k-steps per 32 channels
//   where the tap-pair layout needs 28 (its 14th pair is half empty)
__global__ void pack_weights_bf16x3_d32_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cout, int Cin, bool f16 = false) {
    const int CT = Cout / 16;
    const long long total = (long long)(Cin / 32) * CT * 27 * 64;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int tap = (int)(r % 27);
    r /= 27;
    const int ct = (int)(r % CT);
    const int cc = (int)(r / CT);
    const int kg = lane >> 4;
    const int co = ct * 16 + (lane & 15);
    const int ci = (2 * cc + (kg & 1)) * 16 + (kg >> 1) * 8;
    u16x8 hi, lo;
    for (int j = 0; j < 8; ++j) {
        unsigned short h_, l_;
        split_weight(w[((long long)co * Cin + ci + j) * 27 + tap], f16, h_, l_);
        hi[j] = h_;
        lo[j] = l_;
    }
    const long long o = ((((long long)cc * CT + ct) * 27 + tap) * 2) * 64 + lane;
    wp[o] = __builtin_bit_cast(bf16x8, hi);
    wp[o + 64] = __builtin_bit_cast(bf16x8, lo);
}

// Cout == 16 plane schedule: [16][Cin][27] -> [Cin/16][5 pairs][3 kd][hi|lo][64 lanes][8 bf16]
//   lane = (kg << 4) | i holds W[cout = i][cin = cc*16 + (kg>>1)*8 + j][kd][in-plane tap 2p + (kg&1)]  (tap 9: zero)
__global__ void pack_weights_bf16x3_c16_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cin, bool f16 = false) {
    const int total = (Cin / 16) * 5 * 3 * 64;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63;
    int r = idx >> 6;
    const int kd = r % 3;
    r /= 3;
    const int p = r % 5;
    const int cc = r / 5;
    const int kg = lane >> 4, co = lane & 15;
    const int ci = cc * 16 + (kg >> 1) * 8;
    const int t2 = 2 * p + (kg & 1);
    u16x8 hi, lo;
    for (int j = 0; j < 8; ++j) {
        const float v = t2 < 9 ? w[((long long)co * Cin + ci + j) * 27 + kd * 9 + t2] : 0.f;
        unsigned short h_, l_;
        split_weight(v, f16, h_, l_);
        hi[j] = h_;
        lo[j] = l_;
    }
    const long long o = ((long long)((cc * 5 + p) * 3 + kd) * 2) * 64 + lane;
    wp[o] = __builtin_bit_cast(bf16x8, hi);
    wp[o + 64] = __builtin_bit_cast(bf16x8, lo);
}

// 32x32x16 schedule: [Cout][Cin][27] -> [Cin/16][Cout/32][27 taps][hi|lo][64 lanes][8 bf16]
//   lane = (khalf << 5) | r holds W[cout = ct*32 + r][cin = cc*16 + khalf*8 + j][tap]
__global__ void pack_weights_bf16x3_v32_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cout, int Cin, bool f16 = false) {
    const int CT = Cout / 32;
    const long long total = (long long)(Cin / 16) * CT * 27 * 64;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int tap = (int)(r % 27);
    r /= 27;
    const int ct = (int)(r % CT);
    const int cc = (int)(r / CT);
    const int co = ct * 32 + (lane & 31);
    const int ci = cc * 16 + (lane >> 5) * 8;
    u16x8 hi, lo;
    for (int j = 0; j < 8; ++j) {
        const float v = w[((long long)co * Cin + ci + j) * 27 + tap];
        unsigned short h_, l_;
        split_weight(v, f16, h_, l_);
        hi[j] = h_;
        lo[j] = l_;
    }
    const long long o = ((((long long)cc * CT + ct) * 27 + tap) * 2) * 64 + lane;
    wp[o] = __builtin_bit_cast(bf16x8, hi);
    wp[o + 64] = __builtin_bit_cast(bf16x8, lo);
}

// bijective XCD-aware remap of a flat block id (cdna_hip_programming.md T1): blocks b, b+8, ...
// share an XCD; give each XCD a contiguous run of the logical index space.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// KD = 3: 3x3x3 convolution of a volume; KD = 1: 3x3 convolution of images (a volume with D planes
// that do not see each other: the feature extractor's conv2d on [B*N] channels-last images)
//
// UPS = true fuses ResizeConv3d's trilinear x2 upsample (common_modules.py:332-341) into the producers: a.x is
// the LOW-resolution tensor [B][Din/2][Hin/2][Win/2][Cin] and (Din, Hin, Win) the upsampled size.  A producer
// item is one 2x2x2 cell of low-resolution voxels (8 loads, clamped = ATen's edge rule) from which the 8
// upsampled voxels strictly inside it are blended separably (weights 0.75 / 0.25), masked to zero outside the
// volume (the convolution's padding), split and written to the LDS image: the same one load per staged voxel
// as the plain producer, but 1/8 of the bytes, and neither the resize kernel nor the upsampled tensor exist.
//
// PLANE = true is the consumer schedule for Cout == 16 layers (one cout tile: an activation fragment would
// feed only 3 MFMAs and the loop is LDS-read bound).  A consumer wave owns one h-row of the brick and ALL its
// TD output planes; the 27 taps are walked as (in-plane tap pair p' of 5) x (input plane ip of TD + 2), and
// the fragment of (ip, p') is multiplied with the kd = 0, 1, 2 weights into the accumulators of output
// planes ip, ip - 1, ip - 2: up to 9 MFMAs per fragment, 60 instead of 112 ds_read_b128 per 16-channel slice.
// Weights are packed per (slice, pair, kd) by pack_weights_bf16x3_c16_kernel.
//
// V32 = true is the consumer schedule for Cout % 32 == 0, stride 1: v_mfma_f32_32x32x16_bf16 (32 couts x 32 voxels,
// K = 16 = ONE tap x 16 channels).  The same flops per cycle as 16x16x32, but an MFMA holds the SIMD's vector
// issue port for 8 of its 32 cycles instead of 8 of 16 (MI355X_MICROARCH.md), which is what the consumer's own
// fragment requests and the producer wave sharing its SIMD compete for; 27 taps instead of 14 pairs (no 3.6 %
// padding).  NW / MW count 32-wide tiles; a voxel tile is two 16-runs of adjacent h rows.  The LDS image's row
// pitch is padded to a multiple of 256 B, which makes the 16-lane ds_read_b128 groups conflict-free for every
// tap (rows differ by 0 mod 256 B, so the two half-rows of a group tile the 16 bank units exactly).
// Weights: pack_weights_bf16x3_v32_kernel.
//
// WLDS = true is the schedule for the small launches (a frame or four) where the four consumer waves of a workgroup share
// their cout tiles (WN == 1) and multiply ONE voxel tile each: every wave would fetch the same 2 KiB of weight fragments per
// 48-cycle slot, 170 B / clk / CU through a vector-memory path that returns 64 (measured: ~200 cycles per slot).  The producers
// stage the unit's weight slice (NW x 28 KiB, contiguous in the packed layout) into LDS beside the activation image, once per
// workgroup, under the same per-unit barrier; the consumers read weight fragments like activation fragments (XB-deep, restarted
// per unit): 16 ds_read_b128 per slot and CU instead of 8 KiB of vector-memory returns.
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int S, int KD, bool UPS, bool PLANE, bool V32, bool WLDS, bool F16, bool D32 = false, int DK = 3, int BRD = 0>
__device__ __forceinline__ void conv3d_x3_body(const ConvArgs& a) {
    // (NW == 1: the producers stage the tiles from pct0 = min(cb * NW, CT - NW) on, the consumers clamp tile by tile -- the two agree for one tile)
    static_assert(!WLDS || (WN == 1 && KD == 3 && !UPS && !PLANE && !V32 && S == 1 && NW == 1 && MW <= 2), "LDS-staged weights: one shared cout tile, one or two voxel tiles per wave");
    static_assert(!V32 || (S == 1 && KD == 3 && TW == 16 && TH % 2 == 0 && !PLANE), "32x32x16 schedule: stride 1, 16-wide even bricks");
    static_assert(!PLANE || (NW == 1 && WN == 1 && WM == 4 && TH == 4 && MW == TD && TW == 16 && S == 1 && KD == 3),
                  "plane schedule: Cout == 16, 4 h-rows x TD planes x 16 w per brick");
    static_assert(!UPS || (S == 1 && KD == 3 && TD % 2 == 0 && TH % 2 == 0 && TW % 2 == 0),
                  "fused upsample: stride 1, even bricks (halo bricks are whole 2x2x2 cells)");
    static_assert(WM * WN == 4, "4 consumer waves per workgroup");
    static_assert(WM * MW * (V32 ? 32 : 16) == TD * TH * TW, "brick must be covered by the voxel tiles");
    static_assert(KD == 3 || (KD == 1 && TD == 1), "2-D bricks are one plane thick");
    constexpr int SD = KD == 1 ? 1 : S;               // images are never strided over
    // D32: 32-channel slices -- an MFMA's K = 32 is one tap of TWO 16-channel sub-images (the slice's LDS image is the two side by
    // side, sub-image 1 at +SUB; the lane groups that take a pair's second tap take the second sub-image): 27 slots per 32 channels
    // instead of 2 x 14, half the slices (barriers) per unit.  `kPairs` is then the number of taps.
    static_assert(!D32 || (KD == 3 && S == 1 && !PLANE && !V32 && !WLDS), "32-channel slices: stride-1 3x3x3 variants (plain or fused upsample) only");
    // DK < 3 (D32 only): the depth skip.  A volume no deeper than the brick (UNet's coarsest levels, D = 2 and D = 1) has ONE brick along
    // D, and a wave whose tiles all lie in plane o of it needs the taps kd with 0 <= o + kd - 1 < D only -- the others meet the zero
    // padding: DK = 2 of 3 kd at D = 2, 1 of 3 at D = 1, for every wave of the workgroup alike.  With one tap per slot and the taps
    // kd-major that is a WINDOW of 9 DK slots: the wave's weight pointer and its fragment offsets start kd_lo taps (= planes of the
    // LDS image) in.  A tap PAIR straddles two kd: only the 32-channel-slice form can do this.
    static_assert(DK == 3 || (D32 && DK >= 1 && (TD == 1 ? DK <= 2 : (TD == 2 && WM == 2 && MW * 16 == TH * TW))), "depth skip: one- or two-plane bricks whose waves own whole planes");
    // (TD == 1 && DK == 2: ONE-plane bricks of a TWO-plane volume -- the small-launch units at UNet's level 2: the window of 18 slots starts
    // at kd = 1 for a brick of plane 0 and at kd = 0 for a brick of plane 1, per UNIT)
    constexpr bool DKU = D32 && TD == 1 && DK == 2;
    constexpr int kTaps = KD * 9, kPairs = D32 ? 9 * DK : pairs_of(KD);
    constexpr int kWBlk = D32 ? 27 : kPairs;       // weight fragments per (slice, cout tile) block of the packed weights
    constexpr int ITD = (TD - 1) * SD + KD, ITH = (TH - 1) * S + 3, ITW = (TW - 1) * S + 3;
    constexpr int IV = ITD * ITH * ITW;
    constexpr int QB = D32 ? 3 : 2;                // a staged voxel is 1 << QB items of 4 channels
    constexpr int NIT = (IV * (1 << QB) + 255) / 256;      // staging items per producer thread and unit
    constexpr int ROWP = V32 ? ((ITW * kVSB + 255) / 256) * 256 : ITW * kVSB;    // LDS bytes per halo row
    constexpr int SUB = ITD * ITH * ROWP;          // bytes of one 16-channel LDS image
    constexpr int BUF = D32 ? 2 * SUB : SUB;       // bytes of one slice's image
    // weight pipeline of the consumers: WB register buffers, fragments requested LA slots ahead.
    // A slice has NSLOT slots = the tap pairs rounded up to a multiple of WB (rotation-only slots), so
    // that every register-buffer index is a compile-time constant.
    // One or two tiles per wave (the one-frame variants): a slot is 3-6 MFMAs, far less than an L2 round trip, so the
    // fragments of (nearly) a whole slice are kept in flight -- 16 / 8 buffers instead of 4.
    constexpr int WB = WLDS ? (MW * NW == 1 ? 8 : 4) :         // = XB: weight and activation fragments of a slot travel together
                       (KD == 3 && !PLANE && !V32 && MW * NW == 1) ? 16 : (KD == 3 && !PLANE && !V32 && MW * NW == 2) ? 8 :
#ifndef MVSGI_WBX
#define MVSGI_WBX 5      // five buffers (four slots ahead) for the 2 x 4-tile variants: 214 + 16 registers; measured against 4 and 6
#endif
                       (KD == 3 && !PLANE && !V32 && NW == 2 && MW == 4) ? MVSGI_WBX :
                       NW <= 2 ? 4 : 2;      // NW = 3: three buffers (8 spilled registers) measured 2 % slower
    constexpr int LA = WB - 1;
    constexpr int NSLOT = ((kPairs + WB - 1) / WB) * WB;
    static_assert(NSLOT % WB == 0 && NSLOT >= kPairs, "pipeline geometry");
    constexpr bool RPRE = MW * NW <= 8;            // residual tiles requested during the last slot (register budget)
    // the producers' third register set (requests three units ahead) for the small-launch variants (one or two tiles per wave): an
    // experiment (-DMVSGI_PF3=1), measured round 6 -- 64 -> 64 on one [4, 20, 80] frame 12.9 vs 12.7 us, 128 -> 128 on [2, 10, 40]
    // 15.2 vs 13.8 (its LDS-staged weight sets spill), the one-frame step 0.498 vs 0.482 ms: the one-frame layers are not bound by
    // the age of their loads
#ifndef MVSGI_PF3
#define MVSGI_PF3 0
#endif
    constexpr bool PF3 = MVSGI_PF3 && KD == 3 && !UPS && !PLANE && !V32 && S == 1 && MW * NW <= 2;
    // WARM (an experiment, -DMVSGI_WARM=1): in the small launches (one or two tiles per wave, weights straight from L2) the idle
    // producers pull the weight slice of the unit they are staging through L2 with LDS-DMA requests into a dummy KiB per wave, on
    // the theory that the consumers' fragment requests (LA slots = 700 cycles of MFMAs ahead) meet cold lines.  Measured round 6:
    // 64 -> 64 on one [4, 20, 80] frame 14.4 vs 12.6 us, the one-frame step 0.501 vs 0.491 ms -- SLOWER: the weights are not cold
    // (442 KB per layer stay in a 4 MiB L2 between units) and the extra requests queue in the CU's address path.  In-kernel stamps of
    // that layer: a slice's 84 MFMAs (1344 cycles at the matrix rate) take 4200 cycles, its unit 5000, the producers idle 55 % of
    // every step: the four consumer waves each fetch the SAME weight fragments (16 KiB per 96-cycle slot and CU through a path
    // that returns 64 B / clk: 256 cycles per slot) and restart their fragment pipelines behind every slice's barrier.
#ifndef MVSGI_WARM
#define MVSGI_WARM 0
#endif
    constexpr bool WARM = MVSGI_WARM && KD == 3 && !UPS && !PLANE && !V32 && !WLDS && S == 1 && MW * NW <= 2;

    constexpr int WBYTES = WLDS ? NW * kPairs * 2048 : 0;      // the unit's weight slice in LDS, behind its activation image
    constexpr int BUFW = BUF + WBYTES;             // one buffer of the double-buffered LDS (image + weights)
    constexpr int WARM_LDS = 2 * BUFW;             // WARM: 4 KiB behind the two buffers, one dummy KiB per producer wave
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    float satm = 0.f;          // fp16 split: running maximum |value staged or written in fp16 pieces| (range report)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform
    const bool producer = wave >= 4;
    const int CT = a.Cout / (V32 ? 32 : 16);       // cout tiles
    const int ny = (CT + WN * NW - 1) / (WN * NW);
    const int nchunks = a.Cin / (D32 ? 32 : 16);
    const int total = a.total_units;               // bricks x cout blocks
    const int G = gridDim.x;
    // units of this workgroup: ids blockIdx.x + k*G (G % 8 == 0 or G == total, so id % 8 is this
    // block's XCD), each nchunks slices long
    const int nmine = (total - (int)blockIdx.x + G - 1) / G;
    const int U = nmine * nchunks;
#ifdef MVSGI_STAMPS   // diagnostic build only (tools/stamp_probe.py): s_memtime stamps of block 8
    int nst = 0;
#define STAMP()                                                                                     \
    if (a.dbg && blockIdx.x == 8 && nst < 120) {                                                    \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        if (lane == 0) a.dbg[wave * 128 + nst] = t_;                                                \
        nst++;                                                                                      \
    }
#else
#define STAMP()
#endif
    STAMP()      // kernel entry (stamp 0 of every wave; the exit stamp is its last)

// digit order of a unit id above the w-tile: D before H (bricks stacked along D share halo planes and stay an XCD round apart)
// or, in the -DMVSGI_ORDER_HD diagnostic build, H before D (round 1's order)
#ifdef MVSGI_ORDER_HD
#define MVSGI_DECODE_DH(T, OD, OH, B) { OH = (T % a.tiles_h) * TH; T /= a.tiles_h; OD = (T % a.tiles_d) * TD + a.od_off; B = T / a.tiles_d; }
#else
#define MVSGI_DECODE_DH(T, OD, OH, B) { OD = (T % a.tiles_d) * TD + a.od_off; T /= a.tiles_d; OH = (T % a.tiles_h) * TH; B = T / a.tiles_h; }
#endif
#define MVSGI_DECODE(ID, CB, B, OD, OH, OW)                       \
    {                                                             \
        int t_ = xcd_remap((ID), total);                          \
        CB = t_ % ny;                                             \
        t_ /= ny;                                                 \
        OW = (t_ % a.tiles_w) * TW;                               \
        t_ /= a.tiles_w;                                          \
        MVSGI_DECODE_DH(t_, OD, OH, B)                            \
    }

    if (producer) {
        // =========================== producers: HBM -> split bf16 -> LDS ===========================
        const int ptid = tid - 256;
        if constexpr (UPS) {
            constexpr int CD = ITD / 2, CH = ITH / 2, CW = ITW / 2, NC = CD * CH * CW;
            constexpr int NCQ = NC << QB;                      // cell items of a unit: (cell, channel quad), 4 | 8 quads per cell
            constexpr int NITU = (NCQ + 255) / 256;            // ... per producer thread
            // The last round is rarely full (324 items on 256 threads: 68 left).  When at most half of the threads
            // would be busy, its items are split in two along D -- thread pairs share a cell item, each blends and
            // writes the 4 upsampled voxels of one kd -- so the critical wave does 1 + 0.45 instead of 2 items.
            constexpr int RF = NCQ / 256, REM = NCQ - RF * 256;
            constexpr bool HALF = REM > 0 && 2 * REM <= 256;
#define MVSGI_UPS_CQ(IT) ((HALF && (IT) == RF) ? RF * 256 + (ptid >> 1) : ptid + (IT) * 256)      /* cell-quad index */
#define MVSGI_UPS_LIVE(IT) ((HALF && (IT) == RF) ? (ptid >> 1) < REM : ptid + (IT) * 256 < NCQ)
            const int Dl = a.Din >> 1, Hl = a.Hin >> 1, Wl = a.Win >> 1;
            const int sH_ = Wl * a.Cin, sD_ = Hl * sH_;        // element strides of the low-resolution frame (< 2^23: launcher)
            int lo_d[NITU], lo_h[NITU], lo_w[NITU];             // lower low-res corner of the cell (may be -1)
            unsigned inmask[NITU];                              // bit axis*2+k: upsampled voxel k of the cell is inside
            const float* xb = a.x;
            f32x4 crA[NITU][8], crB[NITU][8];                  // two register sets: loads run two units ahead
            unsigned imA[NITU], imB[NITU];
#define MVSGI_PLAN_UPS(UNIT)                                                                            \
            {                                                                                           \
                int cb_, b_, od_, oh_, ow_;                                                             \
                MVSGI_DECODE(UNIT, cb_, b_, od_, oh_, ow_)                                              \
                (void)cb_;                                                                              \
                _Pragma("unroll") for (int it = 0; it < NITU; ++it) {                                   \
                    const int c = MVSGI_UPS_CQ(it) >> QB;                                               \
                    const int cw = c % CW, ch = (c / CW) % CH, cd = c / (CW * CH);                      \
                    const int gd = od_ - 1 + 2 * cd, gh = oh_ - 1 + 2 * ch, gw = ow_ - 1 + 2 * cw;      \
                    lo_d[it] = (gd - 1) >> 1;                                                           \
                    lo_h[it] = (gh - 1) >> 1;                                                           \
                    lo_w[it] = (gw - 1) >> 1;                                                           \
                    inmask[it] = (gd >= 0 && gd < a.Din ? 1u : 0u) | (gd + 1 < a.Din ? 2u : 0u) |       \
                                 (gh >= 0 && gh < a.Hin ? 4u : 0u) | (gh + 1 < a.Hin ? 8u : 0u) |       \
                                 (gw >= 0 && gw < a.Win ? 16u : 0u) | (gw + 1 < a.Win ? 32u : 0u);      \
                }                                                                                       \
                xb = a.x + (long long)b_ * Dl * Hl * Wl * a.Cin;                                        \
            }
#define MVSGI_CLAMP(V, N) ((V) < 0 ? 0 : ((V) > (N) - 1 ? (N) - 1 : (V)))
            int k2 = 0, cc2 = 0;                               // next unit to request, in walking order
#define MVSGI_ISSUE_UPS_BEGIN()                                                                         \
            if (cc2 == 0 && k2 > 0) { MVSGI_PLAN_UPS((int)blockIdx.x + k2 * G) }
#define MVSGI_ISSUE_UPS1(CR, IM, IT)                                                                    \
            {                                                                                           \
                const int q = MVSGI_UPS_CQ(IT) & ((1 << QB) - 1);                                       \
                const bool live = MVSGI_UPS_LIVE(IT);                                                   \
                const int d0 = live ? MVSGI_CLAMP(lo_d[IT], Dl) : 0, d1 = live ? MVSGI_CLAMP(lo_d[IT] + 1, Dl) : 0; \
                const int h0 = live ? MVSGI_CLAMP(lo_h[IT], Hl) : 0, h1 = live ? MVSGI_CLAMP(lo_h[IT] + 1, Hl) : 0; \
                const int w0 = live ? MVSGI_CLAMP(lo_w[IT], Wl) : 0, w1 = live ? MVSGI_CLAMP(lo_w[IT] + 1, Wl) : 0; \
                const int cofs = cc2 * (D32 ? 32 : 16) + q * 4;                                         \
                /* element offset = d sD + h sH + w sW: six full-rate 24-bit multiplies instead of 14 quarter-rate v_mul_lo_u32 */ \
                const int od_[2] = {__mul24(d0, sD_), __mul24(d1, sD_)}, oh_[2] = {__mul24(h0, sH_), __mul24(h1, sH_)};   \
                const int ow_[2] = {__mul24(w0, a.Cin) + cofs, __mul24(w1, a.Cin) + cofs};              \
                _Pragma("unroll") for (int k = 0; k < 8; ++k)                                           \
                    CR[IT][k] = *reinterpret_cast<const f32x4*>(xb + (od_[(k >> 2) & 1] + oh_[(k >> 1) & 1] + ow_[k & 1])); \
                IM[IT] = inmask[IT];                                                                    \
            }
#define MVSGI_ISSUE_UPS_END()                                                                           \
            {                                                                                           \
                if (++cc2 == nchunks) { cc2 = 0; ++k2; }                                                \
                if (k2 >= nmine) { k2 = nmine - 1; cc2 = nchunks - 1; }                                 \
            }
#define MVSGI_ISSUE_UPS(CR, IM)                                                                         \
            {                                                                                           \
                MVSGI_ISSUE_UPS_BEGIN()                                                                 \
                _Pragma("unroll") for (int it = 0; it < NITU; ++it) MVSGI_ISSUE_UPS1(CR, IM, it)        \
                MVSGI_ISSUE_UPS_END()                                                                   \
            }
#define MVSGI_LERP2(A, B, O0, O1)   /* O0 = 0.75 A + 0.25 B, O1 = 0.25 A + 0.75 B as A + f (B - A): 12 v_fma_f32 */ \
            {                                                                                           \
                const f32x4 d_ = fma4((A), f32x4{-1.f, -1.f, -1.f, -1.f}, (B));   /* B - A, exact, packed */ \
                O0 = fma4(d_, f32x4{0.25f, 0.25f, 0.25f, 0.25f}, (A));             \
                O1 = fma4(d_, f32x4{0.75f, 0.75f, 0.75f, 0.75f}, (A));             \
            }
#define MVSGI_PUT_UPS1(CR, IM, DST, IT)                                                                 \
            if (HALF && (IT) == RF) {                                                                   \
                /* half item: this thread's kd = ptid & 1; blend along D first (one output plane), then H, then W */ \
                if (MVSGI_UPS_LIVE(IT)) {                                                               \
                    const int cq_ = MVSGI_UPS_CQ(IT), c = cq_ >> QB, q = cq_ & 3, kd = ptid & 1;        \
                    const int sub_ = D32 ? ((cq_ >> 2) & 1) * SUB : 0;                                  \
                    const int cw = c % CW, ch = (c / CW) % CH, cd = c / (CW * CH);                      \
                    const float f_ = kd ? 0.75f : 0.25f;                                                \
                    const f32x4 f4_ = {f_, f_, f_, f_};                                                 \
                    f32x4 xd[2][2], xh[2][2], xo[2][2];                                                 \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                     \
                        const f32x4 d_ = fma4(CR[IT][j], f32x4{-1.f, -1.f, -1.f, -1.f}, CR[IT][4 + j]); \
                        xd[j >> 1][j & 1] = fma4(d_, f4_, CR[IT][j]);              \
                    }                                                                                   \
                    _Pragma("unroll") for (int kw = 0; kw < 2; ++kw)                                    \
                        MVSGI_LERP2(xd[0][kw], xd[1][kw], xh[0][kw], xh[1][kw])                         \
                    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                    \
                        MVSGI_LERP2(xh[kh][0], xh[kh][1], xo[kh][0], xo[kh][1])                         \
                    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                    \
                        _Pragma("unroll") for (int kw = 0; kw < 2; ++kw) {                              \
                            const bool ok = ((IM[IT] >> kd) & (IM[IT] >> (2 + kh)) & (IM[IT] >> (4 + kw)) & 1u) != 0; \
                            u32x2 hi, lo;                                                               \
                            split_x4<F16>(xo[kh][kw], hi, lo, satm);                                           \
                            if (!ok) hi = lo = u32x2{0u, 0u};                                           \
                            const int vo_ = ((2 * cd + kd) * ITH + 2 * ch + kh) * ROWP + (2 * cw + kw) * kVSB; \
                            *reinterpret_cast<u32x2*>((DST) + sub_ + vo_ + q * 8) = hi;                 \
                            *reinterpret_cast<u32x2*>((DST) + sub_ + vo_ + 32 + q * 8) = lo;            \
                        }                                                                               \
                }                                                                                       \
            } else {                                                                                    \
                const int e = ptid + (IT) * 256;                                                        \
                if (e < NCQ) {                                                                          \
                    const int c = e >> QB, q = e & 3, sub_ = D32 ? ((e >> 2) & 1) * SUB : 0;            \
                    const int cw = c % CW, ch = (c / CW) % CH, cd = c / (CW * CH);                      \
                    /* separable blend: along W, then H, then D; index bit = upsampled voxel 0 / 1 of the cell */ \
                    f32x4 xw[4][2], xh[2][2][2], xo[2][2][2];                                           \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                       \
                        MVSGI_LERP2(CR[IT][2 * k], CR[IT][2 * k + 1], xw[k][0], xw[k][1])               \
                    _Pragma("unroll") for (int kd = 0; kd < 2; ++kd)                                    \
                        _Pragma("unroll") for (int kw = 0; kw < 2; ++kw)                                \
                            MVSGI_LERP2(xw[2 * kd][kw], xw[2 * kd + 1][kw], xh[kd][0][kw], xh[kd][1][kw]) \
                    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                    \
                        _Pragma("unroll") for (int kw = 0; kw < 2; ++kw)                                \
                            MVSGI_LERP2(xh[0][kh][kw], xh[1][kh][kw], xo[0][kh][kw], xo[1][kh][kw])      \
                    _Pragma("unroll") for (int kd = 0; kd < 2; ++kd)                                    \
                        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                \
                            _Pragma("unroll") for (int kw = 0; kw < 2; ++kw) {                          \
                                const bool ok = ((IM[IT] >> kd) & (IM[IT] >> (2 + kh)) & (IM[IT] >> (4 + kw)) & 1u) != 0; \
                                u32x2 hi, lo;                                                           \
                                split_x4<F16>(xo[kd][kh][kw], hi, lo, satm);                                   \
                                if (!ok) hi = lo = u32x2{0u, 0u};                                       \
                                const int vo_ = ((2 * cd + kd) * ITH + 2 * ch + kh) * ROWP + (2 * cw + kw) * kVSB; \
                                *reinterpret_cast<u32x2*>((DST) + sub_ + vo_ + q * 8) = hi;             \
                                *reinterpret_cast<u32x2*>((DST) + sub_ + vo_ + 32 + q * 8) = lo;        \
                            }                                                                           \
                }                                                                                       \
            }
#define MVSGI_PUT_UPS(CR, IM, DST)                                                                      \
            { _Pragma("unroll") for (int it = 0; it < NITU; ++it) MVSGI_PUT_UPS1(CR, IM, DST, it) }
// request unit u+2 / finish unit u+1 cell by cell (no request bursts in front of the consumers' weight fragments)
#define MVSGI_STEP_UPS(NEW, IMNEW, OLD, IMOLD, DST, DOPUT)                                              \
            {                                                                                           \
                MVSGI_ISSUE_UPS_BEGIN()                                                                 \
                _Pragma("unroll") for (int it = 0; it < NITU; ++it) {                                   \
                    MVSGI_ISSUE_UPS1(NEW, IMNEW, it)                                                    \
                    if (DOPUT) MVSGI_PUT_UPS1(OLD, IMOLD, DST, it)                                      \
                    __builtin_amdgcn_sched_barrier(0);                                                  \
                }                                                                                       \
                MVSGI_ISSUE_UPS_END()                                                                   \
            }
            STAMP()
            MVSGI_PLAN_UPS((int)blockIdx.x)
            MVSGI_ISSUE_UPS(crA, imA)                          // unit 0
            MVSGI_PUT_UPS(crA, imA, ldsb)
            MVSGI_ISSUE_UPS(crA, imA)                          // unit 1 in flight (requests are unconditional, see below)
            STAMP()
            __syncthreads();                                   // image 0 holds unit 0
            STAMP()
            for (int u = 0; u < U; u += 2) {
                MVSGI_STEP_UPS(crB, imB, crA, imA, ldsb + ((u + 1) & 1) * BUF, u + 1 < U)
                STAMP()
                __syncthreads();
                STAMP()
                if (u + 1 < U) {
                    MVSGI_STEP_UPS(crA, imA, crB, imB, ldsb + ((u + 2) & 1) * BUF, u + 2 < U)
                    STAMP()
                    __syncthreads();
                    STAMP()
                }
            }
#undef MVSGI_ISSUE_UPS
#undef MVSGI_PUT_UPS
#undef MVSGI_LERP2
#undef MVSGI_ISSUE_UPS_BEGIN
#undef MVSGI_ISSUE_UPS1
#undef MVSGI_ISSUE_UPS_END
#undef MVSGI_PUT_UPS1
#undef MVSGI_STEP_UPS
#undef MVSGI_PLAN_UPS
#undef MVSGI_CLAMP
#undef MVSGI_UPS_CQ
#undef MVSGI_UPS_LIVE
        } else {
        int goff[NIT];
        unsigned okA = 0, okB = 0;      // (kept for the macro signatures: out-of-volume items are zero-filled by the loads' range check)
        (void)okA; (void)okB;
        // Two register sets: the loads of unit u+2 are requested BEFORE the loads of unit u+1 are waited
        // for, split and written, so an HBM / L2 round trip overlaps a whole unit period instead of being
        // exposed once per unit (stamps: a staged unit took ~8k ticks of which ~5k were that wait, which made
        // the Cout == 16 layers producer-bound).
        f32x4 preA[NIT], preB[NIT];
        // staging plan of a brick: item e = ptid + it*256 -> (halo voxel e>>2, channel quad e&3); padding / surplus items read a
        // valid dummy address and are zeroed by a select (a branch per load would make hipcc wait for each one in turn).
        // An item's halo coordinates and its element offset RELATIVE to the brick's origin never change: they are worked out
        // once per launch (three divisions by constants, two multiplies), and a brick's plan is one add of the (scalar) origin
        // offset plus three range checks per item -- the per-brick plan used to redo all of it and was most of the ~40 vector
        // instructions a producer spent per 16 staged bytes in the single-slice layers (Cin = 16: down.0.first).
        int ibase[NIT];          // byte offset of the item relative to the brick's origin voxel
        // WLDS: this wave's share of the unit's weight slice -- pieces of 1 KiB (one per wave instruction), wave pw takes pieces
        // pw, pw + 4, ...; requested two units ahead and written one unit ahead, like the activation items
        constexpr int WPIECES = WBYTES / 1024, WNIT = WLDS ? WPIECES / 4 : 1;
        static_assert(WPIECES % 4 == 0, "weight pieces are dealt to four producer waves");
        f32x4 wpreA[WNIT], wpreB[WNIT];
        (void)wpreA; (void)wpreB;
        const char* wsrc = reinterpret_cast<const char*>(a.wp) + (wave - 4) * 1024 + lane * 16;
        // WARM: the packed weights as a buffer ([cin slice][cout tile][pair][hi | lo][64 lanes][16 B]; requests past its end move nothing)
        const __amdgpu_buffer_rsrc_t wdesc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(reinterpret_cast<const void*>(a.wp)), 0, WARM ? (int)((long long)nchunks * CT * kPairs * 2048) : 0, 0x00020000);
        (void)wdesc;
        int pct0 = 0;            // first cout tile of the unit the plan stands at
        int wct0 = 0;            // WARM: the same for units of WN * NW tiles
        (void)wct0;
        __amdgpu_buffer_rsrc_t xdesc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0, 0x00020000);
        unsigned cpk[NIT];       // id | ih << 8 | iw << 16
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = ptid + it * 256;
            const int v = e >> QB, q = e & ((1 << QB) - 1);
            const int iw = v % ITW, ih = (v / ITW) % ITH, id = v / (ITW * ITH);
            const bool live = e < IV * (1 << QB);             // surplus items of the last round: halo voxel 0 again (loaded, never stored)
            ibase[it] = (live && !D32) ? (((id * a.Hin + ih) * a.Win + iw) * a.Cin + q * 4) * 4 : 0;       // bytes (D32: unused, see MVSGI_PLAN)
            cpk[it] = live ? (unsigned)(id | (ih << 8) | (iw << 16)) : 0u;
        }
        static_assert(ITD < 255 && ITH < 255 && ITW < 255, "packed halo coordinates");
#define MVSGI_PLAN(UNIT)                                                                                \
        {                                                                                               \
            int cb_, b_, od_, oh_, ow_;                                                                 \
            MVSGI_DECODE(UNIT, cb_, b_, od_, oh_, ow_)                                                  \
            pct0 = cb_ * NW < CT - NW ? cb_ * NW : CT - NW;                                             \
            wct0 = cb_ * (WN * NW) < CT - WN * NW ? cb_ * (WN * NW) : (CT > WN * NW ? CT - WN * NW : 0);  /* first of the unit's WN * NW cout tiles */ \
            const int id0_ = od_ * SD - KD / 2, ih0_ = oh_ * S - 1, iw0_ = ow_ * S - 1;                 \
            const int bbase_ = ((id0_ * a.Hin + ih0_) * a.Win + iw0_) * a.Cin * 4;                      \
            _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                        \
                const unsigned gd = (unsigned)(id0_ + (int)(cpk[it] & 255u));                           \
                const unsigned gh = (unsigned)(ih0_ + (int)((cpk[it] >> 8) & 255u));                    \
                const unsigned gw = (unsigned)(iw0_ + (int)(cpk[it] >> 16));                            \
                const bool ok = gd < (unsigned)a.Din && gh < (unsigned)a.Hin && gw < (unsigned)a.Win;   \
                /* outside the volume: an offset the frame's buffer descriptor does not cover -- the load returns zeros (the \
                   convolution's padding) with no select behind it */                                  \
                /* D32 (twice the items per thread): the offset from the coordinates instead of a register per item */ \
                goff[it] = !ok ? (int)0x80000000 : D32 ? (((int)gd * a.Hin + (int)gh) * a.Win + (int)gw) * a.Cin * 4 + ((ptid + it * 256) & 7) * 16 \
                                                       : ibase[it] + bbase_;                           \
            }                                                                                           \
            xdesc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (long long)b_ * a.Din * a.Hin * a.Win * a.Cin, 0, \
                                                      a.Din * a.Hin * a.Win * a.Cin * 4, 0x00020000);  \
        }
        // request the next unit in walking order (k2, cc2) into a register set; the plan moves on with it
        int k2 = 0, cc2 = 0;
#define MVSGI_ISSUE_BEGIN()                                                                             \
        if (cc2 == 0 && k2 > 0) { MVSGI_PLAN((int)blockIdx.x + k2 * G) }
#define MVSGI_ISSUE1(PRE, IT)                                                                           \
        PRE[IT] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xdesc, goff[IT] + cc2 * (D32 ? 128 : 64), 0, 0));
#define MVSGI_ISSUE_END(OK)                                                                             \
        {                                                                                               \
            if (++cc2 == nchunks) { cc2 = 0; ++k2; }                                                    \
            if (k2 >= nmine) { k2 = nmine - 1; cc2 = nchunks - 1; }   /* past the end: re-request the last unit */ \
        }
#define MVSGI_WISSUE(WPRE)      /* the weight slice of unit (k2, cc2): [cc][cout tile][pair][hi | lo][64 lanes][16 B], NW tiles in a row */ \
        if constexpr (WLDS) {                                                                           \
            const char* ws_ = wsrc + ((long long)cc2 * CT + pct0) * (kPairs * 2048);                     \
            _Pragma("unroll") for (int it = 0; it < WNIT; ++it) WPRE[it] = *reinterpret_cast<const f32x4*>(ws_ + it * 4096); \
        }
#define MVSGI_WARM_ISSUE()      /* WARM: the weight slice of unit (k2, cc2), NW tiles in a row, through L2 into a dummy KiB of LDS */ \
        if constexpr (WARM) {                                                                           \
            const unsigned wo_ = (unsigned)(((long long)cc2 * CT + wct0) * (kPairs * 2048)) + (unsigned)((wave - 4) * 1024 + lane * 16); \
            _Pragma("unroll") for (int it = 0; it < WN * NW * kPairs * 2048 / 4096; ++it)               \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wdesc, (__attribute__((address_space(3))) void*)(ldsb + WARM_LDS + (wave - 4) * 1024), \
                                                         16, wo_ + it * 4096u, 0, 0, 0);                \
        }
#define MVSGI_WPUT(WPRE, DST)                                                                           \
        if constexpr (WLDS) {                                                                           \
            _Pragma("unroll") for (int it = 0; it < WNIT; ++it)                                         \
                *reinterpret_cast<f32x4*>((DST) + BUF + (wave - 4) * 1024 + lane * 16 + it * 4096) = WPRE[it]; \
        }
// (BRD: the image plane below a bottom brick / above a top brick lies outside the volume and its only readers are the skipped slots:
// the items that lie wholly in it -- a fifth of a unit's staging -- are neither requested nor written)
#define MVSGI_ITEM_DEAD(IT) (BRD == 1 ? ((IT) + 1) * 256 <= ITH * ITW * (1 << QB) : BRD == 2 ? (IT) * 256 >= (ITD - 1) * ITH * ITW * (1 << QB) : false)
#define MVSGI_ISSUE(PRE, OK, WPRE)                                                                      \
        {                                                                                               \
            MVSGI_ISSUE_BEGIN()                                                                         \
            MVSGI_WISSUE(WPRE)                                                                          \
            _Pragma("unroll") for (int it = 0; it < NIT; ++it) { if (!MVSGI_ITEM_DEAD(it)) MVSGI_ISSUE1(PRE, it) } \
            MVSGI_ISSUE_END(OK)                                                                         \
        }
#define MVSGI_PUT1(PRE, OK, DST, IT)                                                                    \
        {                                                                                               \
            const int e = ptid + (IT) * 256;                                                            \
            if (e < IV * (1 << QB)) {                                                                   \
                const int v = e >> QB, q = e & 3, sub_ = D32 ? ((e >> 2) & 1) * SUB : 0;                \
                u32x2 hi, lo;                                                                           \
                if (MVSGI_ABL & 4) {     /* diagnostic: what a pre-split input would cost the producers (no split, no mask) */ \
                    hi = u32x2{__builtin_bit_cast(unsigned, PRE[IT][0] + 0.f), __builtin_bit_cast(unsigned, PRE[IT][1] + 0.f)}; \
                    lo = u32x2{__builtin_bit_cast(unsigned, PRE[IT][2] + 0.f), __builtin_bit_cast(unsigned, PRE[IT][3] + 0.f)}; \
                } else {                                                                                \
                    split_x4<F16>(PRE[IT], hi, lo, satm);                                                      \
                }                                                                                       \
                const int vo_ = V32 ? (v / ITW) * ROWP + (v % ITW) * kVSB : v * kVSB;                   \
                *reinterpret_cast<u32x2*>((DST) + sub_ + vo_ + q * 8) = hi;                             \
                *reinterpret_cast<u32x2*>((DST) + sub_ + vo_ + 32 + q * 8) = lo;                        \
            }                                                                                           \
        }
#define MVSGI_PUT(PRE, OK, DST)                                                                         \
        { _Pragma("unroll") for (int it = 0; it < NIT; ++it) if (!MVSGI_ITEM_DEAD(it)) MVSGI_PUT1(PRE, OK, DST, it) }
// one producer step: request unit u+2 into NEW while unit u+1 (OLD, in flight since the last step) is split and
// written, ITEM BY ITEM.  A burst of NIT x 4 waves x 1 KiB requests would sit in the CU's vector-memory queue in
// front of the consumers' weight fragments (measured: the burst form of this pipeline was 7 % SLOWER end to end).
#define MVSGI_STEP(NEW, OKNEW, OLD, OKOLD, DST, DOPUT, WNEW, WOLD)                                      \
        {                                                                                               \
            MVSGI_ISSUE_BEGIN()                                                                         \
            MVSGI_WARM_ISSUE()                                                                          \
            MVSGI_WISSUE(WNEW)                                                                          \
            if (DOPUT) { MVSGI_WPUT(WOLD, DST) }                                                        \
            _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                        \
                if (MVSGI_ITEM_DEAD(it)) continue;                                                      \
                MVSGI_ISSUE1(NEW, it)                                                                   \
                if (DOPUT) MVSGI_PUT1(OLD, OKOLD, DST, it)                                              \
                __builtin_amdgcn_sched_barrier(0);                                                      \
            }                                                                                           \
            MVSGI_ISSUE_END(OKNEW)                                                                      \
        }
        STAMP()
        MVSGI_PLAN((int)blockIdx.x)
        if constexpr (PF3) {
            // Small launches (one or two tiles per wave: a frame to a few): a slice's MFMAs are 0.3-0.7 us, a global-load round trip
            // 1.5-2 us, and with two register sets a step waits for loads that are ONE step old -- the step time was the load latency
            // (profiles/r06_b1_kernel_stats.txt: 12.7-13.7 us per layer for 4-8 slices of ~0.4 us of MFMAs).  Three sets: the loads
            // of unit u+3 go out while unit u+1, requested TWO steps ago, is split and written.
            f32x4 preC[NIT], wpreC[WNIT];
            (void)wpreC;
            unsigned okC = 0;
            (void)okC;
            MVSGI_ISSUE(preA, okA, wpreA)                  // unit 0
            MVSGI_PUT(preA, okA, ldsb)
            MVSGI_WPUT(wpreA, ldsb)
            MVSGI_ISSUE(preA, okA, wpreA)                  // unit 1 in flight (unconditional requests: see below)
            MVSGI_ISSUE(preB, okB, wpreB)                  // unit 2 in flight
            STAMP()
            __syncthreads();                               // image 0 holds unit 0
            STAMP()
            for (int u = 0; u < U; u += 3) {
                // A holds unit u+1, B unit u+2 (both in flight); request u+3 into C, then finish u+1
                MVSGI_STEP(preC, okC, preA, okA, ldsb + ((u + 1) & 1) * BUFW, u + 1 < U && !(MVSGI_ABL & 8), wpreC, wpreA)
                STAMP()
                __syncthreads();                           // unit u multiplied, image of unit u+1 complete
                STAMP()
                if (u + 1 < U) {
                    MVSGI_STEP(preA, okA, preB, okB, ldsb + ((u + 2) & 1) * BUFW, u + 2 < U && !(MVSGI_ABL & 8), wpreA, wpreB)
                    STAMP()
                    __syncthreads();
                    STAMP()
                }
                if (u + 2 < U) {
                    MVSGI_STEP(preB, okB, preC, okC, ldsb + ((u + 3) & 1) * BUFW, u + 3 < U && !(MVSGI_ABL & 8), wpreB, wpreC)
                    STAMP()
                    __syncthreads();
                    STAMP()
                }
            }
        } else {
        MVSGI_ISSUE(preA, okA, wpreA)                      // unit 0
        MVSGI_PUT(preA, okA, ldsb)
        MVSGI_WPUT(wpreA, ldsb)
        // Every request below is UNCONDITIONAL (past the last unit it re-reads that unit): a request under
        // `if (u + 2 < U)` makes the register set a phi of old and new values, the compiler materialises the
        // phi as copies of the freshly loaded registers, and a copy is a use -- s_waitcnt right behind the loads.
        MVSGI_ISSUE(preA, okA, wpreA)                      // unit 1 in flight
        STAMP()
        __syncthreads();                                   // image 0 holds unit 0
        STAMP()
        for (int u = 0; u < U; u += 2) {
            // set A holds unit u+1 (in flight); request u+2 into B, then finish u+1
            MVSGI_STEP(preB, okB, preA, okA, ldsb + ((u + 1) & 1) * BUFW, u + 1 < U && !(MVSGI_ABL & 8), wpreB, wpreA)
            STAMP()
            __syncthreads();                               // unit u multiplied, image of unit u+1 complete
            STAMP()
            if (u + 1 < U) {
                MVSGI_STEP(preA, okA, preB, okB, ldsb + ((u + 2) & 1) * BUFW, u + 2 < U && !(MVSGI_ABL & 8), wpreA, wpreB)
                STAMP()
                __syncthreads();
                STAMP()
            }
        }
        }
#undef MVSGI_ISSUE
#undef MVSGI_WISSUE
#undef MVSGI_WPUT
#undef MVSGI_WARM_ISSUE
#undef MVSGI_PUT
#undef MVSGI_ISSUE_BEGIN
#undef MVSGI_ISSUE1
#undef MVSGI_ISSUE_END
#undef MVSGI_PUT1
#undef MVSGI_ITEM_DEAD
#undef MVSGI_STEP
#undef MVSGI_PLAN
        }
    } else if constexpr (V32) {
        // =========================== consumers, 32x32x16 schedule ===========================
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        const int wm = wave % WM, wn = wave / WM;
        const int n32 = lane & 31, kh2 = lane >> 5;
        const char* wpb = reinterpret_cast<const char*>(a.wp);
        const unsigned lane16 = lane * 16;
        constexpr int HP = TH / 2;                         // h-row pairs per plane
        int xbase[MW], tdv[MW], thv[MW], twv[MW];
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const int t = wm * MW + i;                     // 32-voxel tile: plane t / HP, rows 2*(t % HP) + {0, 1}
            tdv[i] = t / HP;
            thv[i] = 2 * (t % HP) + (n32 >> 4);
            twv[i] = n32 & 15;
            xbase[i] = (tdv[i] * ITH + thv[i]) * ROWP + twv[i] * kVSB + kh2 * 16;
        }
        const long long frame_elems = (long long)a.Do * a.Ho * a.Wo * a.Cout;
        int ctc[NW], ctn[NW];
#define MVSGI_CTILES(DST, CB)                                                         \
        _Pragma("unroll") for (int j = 0; j < NW; ++j) {                              \
            const int c_ = ((CB) * WN + wn) * NW + j;                                 \
            DST[j] = c_ < CT ? c_ : CT - 1;                                           \
        }
        // register buffers: 27 taps = 3 x 9 = 9 x 3, so buffer indices are compile-time constants.  Weight
        // fragments come from L2 (~1 k cycles under load): requested LAW taps (LAW x 192 cycles) ahead;
        // activation fragments come from LDS: 2 taps ahead.
        // activation fragments run XA taps ahead in VB rotating buffers: two ahead in three for the 64-voxel waves; ONE ahead in two
        // for the 128-voxel waves (MW == 4: half the weight fragments per MFMA, 64 accumulator + 64 fragment registers)
        constexpr int VB = MW >= 4 ? 2 : 3, XA = VB - 1, VWB = MW >= 4 ? 6 : 9, LAW = MW >= 4 ? 5 : NW == 1 ? 6 : 4;
        bf16x8 wh[VWB][NW], wl[VWB][NW], xh[VB][MW], xl[VB][MW];
        f32x16 acc[MW][NW];
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int j = 0; j < NW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#define MVSGI_V_LOADW(BUFI, CC, T, CTS)                                                                \
        _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                               \
            const char* q_ = wpb + (((long long)(CC) * CT + CTS[j]) * 27 + (T)) * 2048;  /* wave-uniform */ \
            wh[BUFI][j] = *reinterpret_cast<const bf16x8*>(q_ + l16);                                  \
            wl[BUFI][j] = *reinterpret_cast<const bf16x8*>(q_ + l16 + 1024u);                          \
        }
#define MVSGI_V_READX(BUFI, T)                                                                         \
        {                                                                                              \
            const int off_ = (((T) / 9) * ITH + ((T) / 3) % 3) * ROWP + ((T) % 3) * kVSB;              \
            _Pragma("unroll") for (int i = 0; i < MW; ++i) {                                           \
                xh[BUFI][i] = *reinterpret_cast<const bf16x8*>(img + xbase[i] + off_);                 \
                xl[BUFI][i] = *reinterpret_cast<const bf16x8*>(img + xbase[i] + off_ + 32);            \
            }                                                                                          \
        }
#define MVSGI_V_MFMAS(WBUFI, BUFI)                                                                          \
        _Pragma("unroll") for (int tr = 0; tr < 3; ++tr)                                               \
            _Pragma("unroll") for (int i = 0; i < MW; ++i)                                             \
                _Pragma("unroll") for (int j = 0; j < NW; ++j)                                         \
                    acc[i][j] = mfma32<F16>(tr == 0 ? wl[WBUFI][j] : wh[WBUFI][j], tr == 1 ? xl[BUFI][i] : xh[BUFI][i], acc[i][j]);
        int cb_, b_, od0, oh0, ow0;
        MVSGI_DECODE((int)blockIdx.x, cb_, b_, od0, oh0, ow0)
        MVSGI_CTILES(ctc, cb_)
        {
            unsigned l16 = lane16;
#pragma unroll
            for (int t0 = 0; t0 < LAW; ++t0) { MVSGI_V_LOADW(t0, 0, t0, ctc) }
        }
        STAMP()
        __syncthreads();                                   // image 0 holds unit 0
        STAMP()
        int k = 0, cc = 0;
        for (int u = 0; u < U; ++u) {
            const unsigned char* img = ldsb + (u & 1) * BUF;
            const bool last = cc + 1 == nchunks;
            const bool more = u + 1 < U;
            const int ncc = last ? 0 : cc + 1;
            int ncb = cb_, nb = b_, nod0 = od0, noh0 = oh0, now0 = ow0;
            if (last && more) {
                MVSGI_DECODE((int)blockIdx.x + (k + 1) * G, ncb, nb, nod0, noh0, now0)
                MVSGI_CTILES(ctn, ncb)
            } else {
#pragma unroll
                for (int j = 0; j < NW; ++j) ctn[j] = ctc[j];
            }
            int eoff[MW];                  // in-frame element offset of this lane's voxel (cout 0) or -1
            f32x4 rres[MW][NW][4], esc[NW][4], esh[NW][4];
            if (last) {
                // per-channel scale / shift of this wave's cout tiles: requested before the last slice is
                // multiplied, so the epilogue does not wait for them one group at a time
#pragma unroll
                for (int j = 0; j < NW; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        esc[j][g] = *reinterpret_cast<const f32x4*>(a.scale + ctc[j] * 32 + 8 * g + 4 * kh2);
                        esh[j][g] = *reinterpret_cast<const f32x4*>(a.shift + ctc[j] * 32 + 8 * g + 4 * kh2);
                    }
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    const int od = od0 + tdv[i], oh = oh0 + thv[i], ow = ow0 + twv[i];
                    const bool inside = od < a.Do && oh < a.Ho && ow < a.Wo;
                    eoff[i] = inside ? ((od * a.Ho + oh) * a.Wo + ow) * a.Cout + kh2 * 4 : -1;
                }
            }
            MVSGI_V_READX(0, 0)
            if (XA == 2) { MVSGI_V_READX(1, 1) }
#pragma unroll
            for (int t_ = 0; t_ < 27; ++t_) {
                const int cur = t_ % VB, nx2 = (t_ + XA) % VB, wcur = t_ % VWB;
                unsigned l16 = lane16;
                asm volatile("" : "+v"(l16));
                if (t_ + LAW < 27) {
                    MVSGI_V_LOADW((t_ + LAW) % VWB, cc, t_ + LAW, ctc)
                } else {
                    MVSGI_V_LOADW((t_ + LAW - 27) % VWB, ncc, t_ + LAW - 27, ctn)    // first taps of the next unit (unconditional)
                }
                if (t_ + XA < 27) { MVSGI_V_READX(nx2, t_ + XA) }
                if (t_ == 26 && last && a.res) {
                    const float* rb = a.res + (long long)b_ * frame_elems;
#pragma unroll
                    for (int i = 0; i < MW; ++i)
#pragma unroll
                        for (int j = 0; j < NW; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                rres[i][j][g] = *reinterpret_cast<const f32x4*>(rb + (eoff[i] >= 0 ? eoff[i] + ctc[j] * 32 + 8 * g : 0));
                }
                MVSGI_V_MFMAS(wcur, cur)
                if (t_ + XA < 27) {
                    constexpr int NMEM = 2 * NW + 2 * MW, NMFMA = 3 * MW * NW;
                    constexpr int RATIO = NMFMA / NMEM > 0 ? NMFMA / NMEM : 1;
#pragma unroll
                    for (int q_ = 0; q_ < NMEM; ++q_) {
                        __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);
                        __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            STAMP()
            if (last) {
                // lane (n32, kh2), register 4g + e of tile (i, j): cout ct*32 + 8g + 4*kh2 + e of voxel n32
                const int ct0 = (cb_ * WN + wn) * NW;
                float* yb = a.y + (long long)b_ * frame_elems;
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int j = 0; j < NW; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 r = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]} * esc[j][g] + esh[j][g];
                            if (a.res) r += rres[i][j][g];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                r[e] = r[e] > 0.f ? r[e] : r[e] * a.neg_slope;
                                acc[i][j][4 * g + e] = 0.f;
                            }
                            if (eoff[i] >= 0 && ct0 + j < CT) *reinterpret_cast<f32x4*>(yb + eoff[i] + (ct0 + j) * 32 + 8 * g) = r;
                        }
                k += 1;
                cb_ = ncb; b_ = nb; od0 = nod0; oh0 = noh0; ow0 = now0;
#pragma unroll
                for (int j = 0; j < NW; ++j) ctc[j] = ctn[j];
            }
            cc = ncc;
            STAMP()
            __syncthreads();
            STAMP()
        }
#undef MVSGI_CTILES
#undef MVSGI_V_LOADW
#undef MVSGI_V_READX
#undef MVSGI_V_MFMAS
    } else {
        // =========================== consumers: LDS + L2 weights -> MFMA ===========================
        const int wm = wave % WM, wn = wave / WM;
        const int col = lane & 15, kg = lane >> 4;
#ifndef MVSGI_DHW
#define MVSGI_DHW 0      // 1: the 10 x 8 bricks' tiles as one row of BOTH planes (conflict-free fragment reads, measured 0-1.7 %: DESIGN.md section 3);
                         // off since the depth skip (DK) wants every wave's tiles in ONE plane
#endif
        constexpr bool DHW = MVSGI_DHW && TW == 8 && TD == 2 && S == 1 && !PLANE && !V32 && (ITH * ITW * (kVSB / 16)) % 16 == 8;
        // BRD, the border-plane skip of the 2 x 4 x 16 bricks on 32-channel slices: a brick at the bottom of a volume (BRD = 1) spends the
        // kd = 0 taps of its plane 0 on the zero padding, a brick at the top (BRD = 2) the kd = 2 taps of its plane 1 -- a sixth of such
        // a brick's MFMAs.  Each wave owns two rows of BOTH planes (tiles 0, 1 in plane 0, tiles 2, 3 in plane 1) and the nine slots of
        // that kd read and multiply the tiles of the other plane only -- at COMPILE time (a wave-uniform run-time branch around a slot's
        // MFMAs was built first: 1-23 k spilled registers): the launcher covers the bottom bricks, the top bricks and the bricks between
        // with one launch each (ConvArgs::od_off, tiles_d).  Every 2 x 4 x 16 brick of a 32-channel-slice kernel has the tile order.
        constexpr bool BSK = D32 && DK == 3 && TD == 2 && WM == 2 && MW == 4 && TH == 4 && TW == 16 && S == 1;
        static_assert(BRD == 0 || BSK, "border-plane skip: the 2 x 4 x 16 bricks of the 32-channel-slice kernels");
        const bool second = kg & 1;       // this lane's k-range belongs to the pair's second tap
        const char* wpb = reinterpret_cast<const char*>(a.wp);    // uniform base; per-lane part is lane*16
        const unsigned lane16 = lane * 16;
        // per-lane LDS byte offset of each voxel tile's B fragment (hi part, tap offset excluded)
        int base[MW];
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const int v = (wm * MW + i) * 16 + col;
            // stride-1 two-plane bricks 8 wide: a tile is the SAME row of both planes (voxel order h, d, w) -- its halves are a plane
            // apart (ITH * ITW * 5 sixteen-byte units = 8 mod 16 for the 10 x 8 bricks: the two halves take disjoint units of a
            // bank row) where two rows of one plane (50 units = 2 mod 16 apart) collide on two of eight units
            const int w_ = v % TW, h_ = BSK ? 2 * wm + (i & 1) : DHW ? v / (TW * TD) : (v / TW) % TH, d_ = BSK ? (BRD == 2 ? 1 - (i >> 1) : (i >> 1)) : DHW ? (v / TW) % TD : v / (TW * TH);
            base[i] = (((d_ * SD) * ITH + h_ * S) * ITW + w_ * S) * kVSB + (kg >> 1) * 16;
        }
        // position of this lane's voxel of each tile inside the brick (unit-independent)
        int tdv[MW], thv[MW], twv[MW];
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const int v = PLANE ? (i * TH + wm) * TW + col : (wm * MW + i) * 16 + col;     // PLANE: tile i = plane i of row wm
            twv[i] = v % TW;
            thv[i] = BSK ? 2 * wm + (i & 1) : DHW ? v / (TW * TD) : (v / TW) % TH;
            tdv[i] = BSK ? (BRD == 2 ? 1 - (i >> 1) : (i >> 1)) : DHW ? (v / TW) % TD : v / (TW * TH);
        }
        if constexpr (D32 && DK < 3 && !DKU) {      // (the launcher checked a.Do == DK: one brick along D, its origin plane 0)
            static_assert(!DHW, "depth skip: plane-pure waves");
            const int kd_lo = (TD == 2 ? wm : 0) == 0 ? 1 : 0;      // plane 0 starts at kd = 1, plane 1 of a two-plane volume at kd = 0
            wpb += kd_lo * 9 * 2048;
#pragma unroll
            for (int i = 0; i < MW; ++i) base[i] += kd_lo * (ITH * ITW * kVSB);
        }
        const long long frame_elems = (long long)a.Do * a.Ho * a.Wo * a.Cout;     // < 2^31 (checked on the host)
        int ctc[NW], ctn[NW];             // clamped cout tiles of the current / the next unit
#define MVSGI_CTILES(DST, CB)                                                         \
        _Pragma("unroll") for (int j = 0; j < NW; ++j) {                              \
            const int c_ = ((CB) * WN + wn) * NW + j;                                 \
            DST[j] = c_ < CT ? c_ : CT - 1;                                           \
        }
        bf16x8 wh[WB][NW], wl[WB][NW];
        // activation fragments: double-buffered when they fit; the one-frame variants (3 or 6 MFMAs per slot, less than an
        // LDS round trip) keep 7 / 3 slots of fragments in flight (stamps: 210 ticks per 48-cycle slot with one)
        constexpr int XB = (KD == 3 && !PLANE && !V32 && MW * NW == 1) ? 8 : (KD == 3 && !PLANE && !V32 && MW * NW == 2) ? 4 : MW <= 4 ? 2 : 1;
        constexpr int XLA = XB > 1 ? XB - 1 : 1;       // slots ahead
        constexpr int MH = XB >= 2 ? MW : MW / 2;      // voxel tiles per half
        bf16x8 xh[XB][MW], xl[XB][MW];
// (BRD = 2, the bricks at the top of the volume: the slots walk kd = 2, 1, 0 and tiles 0, 1 lie in plane 1 -- the mirror image of
// BRD = 1, so that the skipped tiles are the first two of the first nine slots in both; the other order spilled hundreds of registers)
#define MVSGI_TAP(P) (BRD == 2 ? (2 - (P) / 9) * 9 + (P) % 9 : (P))
#define MVSGI_LOADW(BUFI, CC, P, CTS) MVSGI_LOADWB(wpb, BUFI, CC, P, CTS)
#define MVSGI_LOADWB(WPB, BUFI, CC, P, CTS)                                                           \
        _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                              \
            const char* q_ = (WPB) + (((long long)(CC) * CT + CTS[j]) * kWBlk) * 2048;  /* wave-uniform */ \
            wh[BUFI][j] = *reinterpret_cast<const bf16x8*>(q_ + (l16 + (unsigned)MVSGI_TAP(P) * 2048u));          \
            wl[BUFI][j] = *reinterpret_cast<const bf16x8*>(q_ + (l16 + (unsigned)MVSGI_TAP(P) * 2048u + 1024u));  \
        }
#define MVSGI_READX(BUFI, P, I0, I1)                                                                  \
        {                                                                                             \
            const int p_ = (P);                                                                       \
            const int t0_ = 2 * p_, t1_ = (2 * p_ + 1 < kTaps) ? 2 * p_ + 1 : 2 * p_;                 \
            const int o0_ = (((t0_ / 9) * ITH + (t0_ / 3) % 3) * ITW + t0_ % 3) * kVSB;               \
            const int o1_ = (((t1_ / 9) * ITH + (t1_ / 3) % 3) * ITW + t1_ % 3) * kVSB;               \
            const int tp_ = MVSGI_TAP(p_);                                                            \
            const int od_ = (((tp_ / 9) * ITH + (tp_ / 3) % 3) * ITW + tp_ % 3) * kVSB;   /* D32: tap tp_ of sub-image 0 | 1 */ \
            const int off_ = D32 ? od_ + (second ? SUB : 0) : (second ? o1_ : o0_);                   \
            _Pragma("unroll") for (int i = (I0); i < (I1); ++i) {                                     \
                xh[BUFI][i] = *reinterpret_cast<const bf16x8*>(img + base[i] + off_);                 \
                xl[BUFI][i] = *reinterpret_cast<const bf16x8*>(img + base[i] + off_ + 32);            \
            }                                                                                         \
            if constexpr (WLDS) {   /* the slot's weight fragments from the unit's LDS slice (lane-contiguous: conflict-free) */ \
                _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                      \
                    wh[BUFI][j] = *reinterpret_cast<const bf16x8*>(img + BUF + (j * kPairs + p_) * 2048 + (int)lane16);          \
                    wl[BUFI][j] = *reinterpret_cast<const bf16x8*>(img + BUF + (j * kPairs + p_) * 2048 + 1024 + (int)lane16);   \
                }                                                                                     \
            }                                                                                         \
        }
// term-major order: the three products of one accumulator are MW*NW MFMAs apart, never
// back to back (a dependent v_mfma_f32_16x16x32_bf16 issued right behind its producer waits
// for it: measured 19.1 instead of 16 cycles per MFMA with the accumulator-major order)
#define MVSGI_MFMAS(WBUF, XBUF, I0, I1)                                                               \
        _Pragma("unroll") for (int i = (I0); i < (I1); ++i)                                           \
            _Pragma("unroll") for (int j = 0; j < NW; ++j)                                            \
                acc[i][j] = mfma16<F16>(wl[WBUF][j], xh[XBUF][i], acc[i][j]); \
        _Pragma("unroll") for (int i = (I0); i < (I1); ++i)                                           \
            _Pragma("unroll") for (int j = 0; j < NW; ++j)                                            \
                (TACC ? acc1[i][j] : acc[i][j]) = mfma16<F16>(wh[WBUF][j], xl[XBUF][i], TACC ? acc1[i][j] : acc[i][j]); \
        _Pragma("unroll") for (int i = (I0); i < (I1); ++i)                                           \
            _Pragma("unroll") for (int j = 0; j < NW; ++j)                                            \
                (TACC ? acc2[i][j] : acc[i][j]) = mfma16<F16>(wh[WBUF][j], xh[XBUF][i], TACC ? acc2[i][j] : acc[i][j]);
        // one or two tiles per wave: the three products of a tile would sit back to back on ONE accumulator (each waits for
        // the one before: stamps, 200 ticks per 48-cycle slot); they get an accumulator each, summed in the epilogue
        constexpr bool TACC = KD == 3 && !PLANE && !V32 && MW * NW <= 2;
        f32x4 acc[MW][NW], acc1[MW][NW], acc2[MW][NW];      // (acc1, acc2 dead unless TACC)
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (TACC) acc1[i][j] = acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }

        int cb_, b_, od0, oh0, ow0;
        MVSGI_DECODE((int)blockIdx.x, cb_, b_, od0, oh0, ow0)
        MVSGI_CTILES(ctc, cb_)
        // ---- plane schedule (Cout == 16): fragments of one in-plane tap pair at a time ----
        constexpr int NPP = 5;                              // in-plane tap pairs (9 taps)
        constexpr int PD = PLANE ? TD + 2 : 1;
        bf16x8 pwh[3][PLANE ? 3 : 1], pwl[3][PLANE ? 3 : 1];             // [buffer][kd]: weights run two pairs ahead
        bf16x8 pxh[2][PD], pxl[2][PD];                                   // [buffer][input plane]
        const int pbase = (wm * ITW + col) * kVSB + (kg >> 1) * 16;      // row wm, plane 0, tap (0, 0)
#define MVSGI_PL_LOADW(BUFI, CC, P)                                                                    \
        _Pragma("unroll") for (int kd = 0; kd < 3; ++kd) {                                             \
            const char* q_ = wpb + ((long long)(((CC) * NPP + (P)) * 3 + kd)) * 2048;   /* wave-uniform */ \
            pwh[BUFI][kd] = *reinterpret_cast<const bf16x8*>(q_ + l16);                                \
            pwl[BUFI][kd] = *reinterpret_cast<const bf16x8*>(q_ + l16 + 1024u);                        \
        }
#define MVSGI_PL_READX(BUFI, P)                                                                        \
        {                                                                                              \
            const int t0_ = 2 * (P), t1_ = 2 * (P) + 1 < 9 ? 2 * (P) + 1 : 2 * (P);                    \
            const int o0_ = ((t0_ / 3) * ITW + t0_ % 3) * kVSB, o1_ = ((t1_ / 3) * ITW + t1_ % 3) * kVSB; \
            const int off_ = pbase + (second ? o1_ : o0_);                                             \
            _Pragma("unroll") for (int ip = 0; ip < PD; ++ip) {                                        \
                pxh[BUFI][ip] = *reinterpret_cast<const bf16x8*>(img + off_ + ip * (ITH * ITW * kVSB)); \
                pxl[BUFI][ip] = *reinterpret_cast<const bf16x8*>(img + off_ + ip * (ITH * ITW * kVSB) + 32); \
            }                                                                                          \
        }
// term-major, then kd, then plane: the three products of one accumulator are >= TD MFMAs apart
#define MVSGI_PL_MFMAS(WBUFI, BUFI)                                                                         \
        _Pragma("unroll") for (int tr = 0; tr < 3; ++tr)                                               \
            _Pragma("unroll") for (int kd = 0; kd < 3; ++kd)                                           \
                _Pragma("unroll") for (int i = 0; i < TD; ++i)                                         \
                    acc[i][0] = mfma16<F16>(tr == 0 ? pwl[WBUFI][kd] : pwh[WBUFI][kd], tr == 1 ? pxl[BUFI][i + kd] : pxh[BUFI][i + kd], acc[i][0]);
        if constexpr (PLANE) {
            unsigned l16 = lane16;
            MVSGI_PL_LOADW(0, 0, 0)
            MVSGI_PL_LOADW(1, 0, 1)
        } else if constexpr (!WLDS) {
            unsigned l16 = lane16;
#pragma unroll
            for (int s0 = 0; s0 < LA; ++s0) { MVSGI_LOADWB(wpb + (DKU && od0 == 0 ? 9 * 2048 : 0), s0, 0, s0, ctc) }
        }
        STAMP()
        __syncthreads();                                   // image 0 holds unit 0
        STAMP()
        int k = 0, cc = 0;
        for (int u = 0; u < U; ++u) {
            const unsigned char* img = ldsb + (u & 1) * BUFW + (DKU && od0 == 0 ? ITH * ITW * kVSB : 0);      // (DKU: plane 0 starts at kd = 1)
            const bool last = cc + 1 == nchunks;
            const bool more = u + 1 < U;
            const int ncc = last ? 0 : cc + 1;
            int ncb = cb_, nb = b_, nod0 = od0, noh0 = oh0, now0 = ow0;
            if (last && more) {
                MVSGI_DECODE((int)blockIdx.x + (k + 1) * G, ncb, nb, nod0, noh0, now0)
                MVSGI_CTILES(ctn, ncb)
            } else {
#pragma unroll
                for (int j = 0; j < NW; ++j) ctn[j] = ctc[j];
            }
            // weight bases of this unit's slice and of the slice after it (DKU: the unit's window of taps starts at kd = 1 in plane 0)
            const char* const wpc_ = wpb + (DKU && od0 == 0 ? 9 * 2048 : 0);
            const char* const wpn_ = wpb + (DKU && nod0 == 0 ? 9 * 2048 : 0);
            // per-channel scale / shift of this wave's cout tiles: requested before the last slice is
            // multiplied so the epilogue does not wait for them one tile at a time
            f32x4 esc[NW], esh[NW], rres[RPRE ? MW : 1][RPRE ? NW : 1];
            int eoff[MW];                  // in-frame element offset of (voxel, cout 4*kg) or -1 outside the volume
            int soff[MW];                  // split-padded output (a.y_split): in-frame byte offset of the voxel's record
            const int ypd = a.ys_2d ? 0 : 1, ypp = a.ys_2d ? 2 : 1;   // borders of the split-padded output along D and along H / W
            // (requested in every slice, not only the last: loads under a run-time `if` make every later wait of the unit
            // assume they were never issued, i.e. wait for younger weight fragments than needed)
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                esc[j] = *reinterpret_cast<const f32x4*>(a.scale + ctc[j] * 16 + kg * 4);
                esh[j] = *reinterpret_cast<const f32x4*>(a.shift + ctc[j] * 16 + kg * 4);
            }
            if (last) {
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    const int od = od0 + tdv[i], oh = oh0 + thv[i], ow = ow0 + twv[i];
                    const bool inside = od < a.Do && oh < a.Ho && ow < a.Wo;
                    eoff[i] = inside ? ((od * a.Ho + oh) * a.Wo + ow) * a.Cout + kg * 4 : -1;
                    soff[i] = ((((od + ypd) * (a.Ho + 2 * ypp) + oh + ypp) * (a.Wo + 2 * ypp) + ow + ypp) * a.Cout) * 4 + (kg >> 1) * 16 + (kg & 1) * 8;
                }
            }
            if constexpr (PLANE) {
                MVSGI_PL_READX(0, 0)
                // 6 slots = 5 pairs + 1 rotation-only slot, so that pair q of every unit sits in weight buffer q % 3.
                // Weight fragments are requested two slots (72 MFMAs, ~1.2 k cycles) ahead: one pair ahead is less than
                // an L2 round trip under load.
#pragma unroll
                for (int p_ = 0; p_ < NPP + 1; ++p_) {
                    const int cur = p_ & 1, nxt = cur ^ 1;
                    unsigned l16 = lane16;
                    asm volatile("" : "+v"(l16));
                    if (p_ + 2 < NPP) {
                        MVSGI_PL_LOADW((p_ + 2) % 3, cc, p_ + 2)
                    } else if (p_ + 2 > NPP) {
                        MVSGI_PL_LOADW(p_ + 2 - (NPP + 1), ncc, p_ + 2 - (NPP + 1))     // pairs 0, 1 of the next unit (unconditional, see below)
                    }
                    if (p_ < NPP) {
                        if (p_ + 1 < NPP) {
                            MVSGI_PL_READX(nxt, p_ + 1)
                        } else if (last && a.res) {
                            // residual planes of this row: behind the unit's last weight requests
                            const float* rb = a.res + (long long)b_ * frame_elems;
#pragma unroll
                            for (int i = 0; i < MW; ++i)
                                rres[i][0] = *reinterpret_cast<const f32x4*>(rb + (eoff[i] >= 0 ? eoff[i] : 0));
                        }
                        MVSGI_PL_MFMAS(p_ % 3, cur)
                        if (p_ + 1 < NPP) {
                            // the fragment requests of the slot spread over its 36 MFMAs
#pragma unroll
                            for (int q_ = 0; q_ < 6 + 2 * PD; ++q_) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 36 / (6 + 2 * PD) > 0 ? 36 / (6 + 2 * PD) : 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);
                            }
                            __builtin_amdgcn_sched_group_barrier(0x008, 36, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            if (XB >= 2) {
#pragma unroll
                for (int p0 = 0; p0 < XLA; ++p0) { MVSGI_READX(p0 % XB, p0, 0, MW) }
            } else {
                MVSGI_READX(0, 0, 0, MW)
            }
#pragma unroll
            for (int s_ = 0; s_ < NSLOT; ++s_) {
                const int wcur = s_ % WB;
                unsigned l16 = lane16;
                asm volatile("" : "+v"(l16));      // keep `lane*16 + const` from being hoisted 28x out of the loop
                const int xcur = XB >= 2 ? (s_ % XB) : 0, xnxt = XB >= 2 ? ((s_ + XLA) % XB) : 0;
                // weight fragments LA slots ahead (this slice, or the first slots of the next unit)
                if (!(MVSGI_ABL & 1) && !WLDS) {
                    if (s_ + LA < NSLOT) {
                        if (s_ + LA < kPairs) { MVSGI_LOADWB(wpc_, (s_ + LA) % WB, cc, s_ + LA, ctc) }
                    } else {
                        // unconditional (past the last unit: chunk 0 of this one again, unused): under `if (more)` the
                        // compiler must assume the requests were never made and waits for far younger loads than needed
                        MVSGI_LOADWB(wpn_, (s_ + LA - NSLOT) % WB, ncc, s_ + LA - NSLOT, ctn)
                    }
                }
                if (RPRE && s_ == kPairs - 1 && last && a.res) {
                    // residual tiles of the brick: behind the last weight requests of the unit, ahead
                    // of the last slot's MFMAs (loads return in issue order)
                    const float* rb = a.res + (long long)b_ * frame_elems;
#pragma unroll
                    for (int i = 0; i < MW; ++i)
#pragma unroll
                        for (int j = 0; j < NW; ++j)
                            rres[i][j] = *reinterpret_cast<const f32x4*>(rb + (eoff[i] >= 0 ? eoff[i] + ctc[j] * 16 : 0));
                }
                if (s_ < kPairs - ((MVSGI_ABL & 32) ? 1 : 0)) {       // (ABL 32: without the half-empty last pair -- the bound of 32-channel slices)
                    if (XB >= 2) {
                        // one scheduling region per slot: the fragment requests of a later slot (2*NW weight
                        // loads, 2*MW LDS reads) are interleaved one per RATIO MFMAs, so their issue
                        // cycles hide in the MFMA gaps instead of draining the matrix pipe between slots
                        // (BRD: the tiles of the plane whose tap of this slot meets the padding are neither read nor multiplied)
#define MVSGI_T0(SL) ((BRD != 0 && (SL) < 9) ? MW / 2 : 0)
#define MVSGI_T1(SL) (MW)
                        if (s_ + XLA < kPairs && !(MVSGI_ABL & 2)) MVSGI_READX(xnxt, s_ + XLA, MVSGI_T0(s_ + XLA), MVSGI_T1(s_ + XLA))
                        if (!(MVSGI_ABL & 16)) { MVSGI_MFMAS(wcur, xcur, MVSGI_T0(s_), MVSGI_T1(s_)) }
                        constexpr int NMEM = 2 * NW + 2 * MW, NMFMA = 3 * MW * NW;
                        constexpr int RATIO = NMFMA / NMEM > 0 ? NMFMA / NMEM : 1;
                        // (the interleave pattern counts a FULL slot: the border-skipped slots, which hold half the tiles, and the slot
                        // before them are left to the compiler -- with the pattern on them the BRD = 2 kernels spilled hundreds of registers)
                        const bool full_ = MVSGI_T1(s_) - MVSGI_T0(s_) == MW && (s_ + XLA >= kPairs || MVSGI_T1(s_ + XLA) - MVSGI_T0(s_ + XLA) == MW);
                        if (full_) {
#pragma unroll
                        for (int q_ = 0; q_ < NMEM; ++q_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, RATIO, 0);      // MFMA
                            __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);          // VMEM read | DS read
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);          // the rest
                        }
                    } else {
                        __builtin_amdgcn_sched_barrier(0);
                        MVSGI_MFMAS(wcur, 0, 0, MH)
                        __builtin_amdgcn_sched_barrier(0);
                        if (s_ + 1 < kPairs) MVSGI_READX(0, s_ + 1, 0, MH)
                        __builtin_amdgcn_sched_barrier(0);
                        MVSGI_MFMAS(wcur, 0, MH, MW)
                        __builtin_amdgcn_sched_barrier(0);
                        if (s_ + 1 < kPairs) MVSGI_READX(0, s_ + 1, MH, MW)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            STAMP()
            if (last) {
                // epilogue of the finished brick: lane (col, kg) of tile (i, j) holds couts
                // ct*16 + 4*kg + 0..3 of voxel i*16 + col.  The residual tiles were requested during
                // the last slot (one exposed round trip per brick, not one per tile).
                const int ct0 = (cb_ * WN + wn) * NW;
                float* yb = a.y + (long long)b_ * frame_elems;
                f32x4 rl[RPRE ? 1 : MW][RPRE ? 1 : NW];
                if (!RPRE && a.res) {         // many tiles: request them all here, still one round trip
                    const float* rb = a.res + (long long)b_ * frame_elems;
#pragma unroll
                    for (int i = 0; i < (RPRE ? 1 : MW); ++i)
#pragma unroll
                        for (int j = 0; j < (RPRE ? 1 : NW); ++j)
                            rl[i][j] = *reinterpret_cast<const f32x4*>(rb + (eoff[i] >= 0 ? eoff[i] + ctc[j] * 16 : 0));
                }
#pragma unroll
                for (int i = 0; i < MW; ++i) {
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        if (TACC) acc[i][j] = (acc[i][j] + acc1[i][j]) + acc2[i][j];    // small terms first
                        f32x4 r = acc[i][j] * esc[j] + esh[j];
                        if (a.res) r += RPRE ? rres[RPRE ? i : 0][RPRE ? j : 0] : rl[RPRE ? 0 : i][RPRE ? 0 : j];
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : r[e] * a.neg_slope;
                        if (a.y_split) {
                            // split-padded output: slice (ct0 + j) of the voxel record, this lane's 4 couts = 8 B of hi and 8 B of lo
                            u32x2 hi, lo;
                            split_x4<F16>(r, hi, lo, satm);
                            if (eoff[i] >= 0 && ct0 + j < CT) {
                                unsigned char* q = a.y_split + (long long)b_ * ((long long)(a.Do + 2 * ypd) * (a.Ho + 2 * ypp) * (a.Wo + 2 * ypp) * a.Cout * 4) +
                                                   soff[i] + (ct0 + j) * 64;
                                *reinterpret_cast<u32x2*>(q) = hi;
                                *reinterpret_cast<u32x2*>(q + 32) = lo;
                            }
                        } else if (eoff[i] >= 0 && ct0 + j < CT) *reinterpret_cast<f32x4*>(yb + eoff[i] + (ct0 + j) * 16) = r;
                        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (TACC) acc1[i][j] = acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                k += 1;
                cb_ = ncb; b_ = nb; od0 = nod0; oh0 = noh0; ow0 = now0;
#pragma unroll
                for (int j = 0; j < NW; ++j) ctc[j] = ctn[j];
            }
            cc = ncc;
            STAMP()
            __syncthreads();                               // image of unit u+1 complete, image u free
            STAMP()
        }
#undef MVSGI_T0
#undef MVSGI_T1
#undef MVSGI_TAP
#undef MVSGI_PL_LOADW
#undef MVSGI_PL_READX
#undef MVSGI_PL_MFMAS
#undef MVSGI_CTILES
#undef MVSGI_LOADW
#undef MVSGI_READX
#undef MVSGI_MFMAS
    }
#undef MVSGI_DECODE
    STAMP()
#undef STAMP
    if constexpr (F16) sf_sat_report(a.sat, kSatSplit, satm, 65504.f);
}

// the two arithmetics of the body as two kernels (the names the profiler and mvsgi_conv3d_variant_f32 report)
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int S, int KD = 3, bool UPS = false, bool PLANE = false,
          bool V32 = false, bool WLDS = false>
__global__ __launch_bounds__(512, 2) void conv3d_bf16x3_kernel(ConvArgs a) {      // 2 waves per SIMD: 256 registers
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, S, KD, UPS, PLANE, V32, WLDS, false>(a);
}
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int S, int KD = 3, bool UPS = false, bool PLANE = false,
          bool V32 = false, bool WLDS = false>
__global__ __launch_bounds__(512, 2) void conv3d_f16x3_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, S, KD, UPS, PLANE, V32, WLDS, true>(a);
}

// the same body on 32-channel slices (D32; plain stride-1 3x3x3 bricks): kernels of their own, so that the others keep their names
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW>
__global__ __launch_bounds__(512, 2) void conv3d_bf16x3_d32_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, false, false, false, false, false, true>(a);
}
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW>
__global__ __launch_bounds__(512, 2) void conv3d_f16x3_d32_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, false, false, false, false, true, true>(a);
}

// ... with the depth skip (DK = 2: two-plane volumes on two-plane bricks; DK = 1: one-plane volumes on one-plane bricks)
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW>
__global__ __launch_bounds__(512, 2) void conv3d_bf16x3_d32_dk_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, false, false, false, false, false, true, TD>(a);
}
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW>
__global__ __launch_bounds__(512, 2) void conv3d_f16x3_d32_dk_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, false, false, false, false, true, true, TD>(a);
}
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW>
__global__ __launch_bounds__(512, 2) void conv3d_bf16x3_d32u_dk_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, true, false, false, false, false, true, TD>(a);
}
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW>
__global__ __launch_bounds__(512, 2) void conv3d_f16x3_d32u_dk_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, true, false, false, false, true, true, TD>(a);
}

// ... one-plane bricks in a TWO-plane volume (the small-launch units at UNet's level 2): 18 slots, the window starts per unit
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW>
__global__ __launch_bounds__(512, 2) void conv3d_bf16x3_d32_dk2_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, false, false, false, false, false, true, 2>(a);
}
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW>
__global__ __launch_bounds__(512, 2) void conv3d_f16x3_d32_dk2_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, false, false, false, false, true, true, 2>(a);
}

// ... the bricks at the bottom (BRD = 1) / at the top (BRD = 2) of a volume: the border-plane skip
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int BRD>
__global__ __launch_bounds__(512, 2) void conv3d_bf16x3_d32_brd_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, false, false, false, false, false, true, 3, BRD>(a);
}
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int BRD>
__global__ __launch_bounds__(512, 2) void conv3d_f16x3_d32_brd_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, false, false, false, false, true, true, 3, BRD>(a);
}
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int BRD>
__global__ __launch_bounds__(512, 2) void conv3d_bf16x3_d32u_brd_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, true, false, false, false, false, true, 3, BRD>(a);
}
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int BRD>
__global__ __launch_bounds__(512, 2) void conv3d_f16x3_d32u_brd_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, true, false, false, false, true, true, 3, BRD>(a);
}

// ... with the trilinear x2 upsample in the producers (UPS)
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW>
__global__ __launch_bounds__(512, 2) void conv3d_bf16x3_d32u_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, true, false, false, false, false, true>(a);
}
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW>
__global__ __launch_bounds__(512, 2) void conv3d_f16x3_d32u_kernel(ConvArgs a) {
    conv3d_x3_body<NW, MW, WM, WN, TD, TH, TW, 1, 3, true, false, false, false, true, true>(a);
}

template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int S, int KD = 3, bool UPS = false, bool PLANE = false,
          bool V32 = false, bool WLDS = false, bool F16 = false, bool D32 = false, bool DSK = false, int BRD = 0>
int launch_bf16x3(ConvArgs a, hipStream_t st) {
    constexpr int ITD = (TD - 1) * (KD == 1 ? 1 : S) + KD, ITH = (TH - 1) * S + 3, ITW = (TW - 1) * S + 3;
    constexpr int ROWP = V32 ? ((ITW * kVSB + 255) / 256) * 256 : ITW * kVSB;
    constexpr bool WARM = MVSGI_WARM && KD == 3 && !UPS && !PLANE && !V32 && !WLDS && S == 1 && MW * NW <= 2;      // (conv3d_x3_body)
    static_assert(!D32 || (S == 1 && KD == 3 && !PLANE && !V32 && !WLDS), "32-channel slices: stride-1 3x3x3 variants");
    constexpr size_t lds_bytes = (size_t)2 * ((D32 ? 2 : 1) * ITD * ITH * ROWP + (WLDS ? NW * pairs_of(KD) * 2048 : 0)) + (WARM ? 4096 : 0);   // double-buffered image (+ weight slice) (+ the warm-up's dummy KiB per producer wave)
    static_assert(lds_bytes <= 160 * 1024, "LDS images too large");
    void (*kern)(ConvArgs);
    static_assert(!DSK || (D32 && TD <= 2), "depth skip: 32-channel slices, one- or two-plane bricks");
    constexpr bool DKU = DSK && BRD == 3;      // one-plane bricks in a two-plane volume (BRD = 3 here: the launcher's name for that mode)
    static_assert(!DKU || (TD == 1 && !UPS), "per-unit depth skip: one-plane bricks");
    MVSGI_REQUIRE(!DSK || a.Do == (DKU ? 2 : TD), "conv3d: the depth-skip kernels serve volumes exactly %d planes deep (got %d)", DKU ? 2 : TD, a.Do);
    if constexpr (DKU && F16) kern = conv3d_f16x3_d32_dk2_kernel<NW, MW, WM, WN, TD, TH, TW>;
    else if constexpr (DKU) kern = conv3d_bf16x3_d32_dk2_kernel<NW, MW, WM, WN, TD, TH, TW>;
    else if constexpr (BRD != 0 && UPS && F16) kern = conv3d_f16x3_d32u_brd_kernel<NW, MW, WM, WN, TD, TH, TW, BRD>;
    else if constexpr (BRD != 0 && UPS) kern = conv3d_bf16x3_d32u_brd_kernel<NW, MW, WM, WN, TD, TH, TW, BRD>;
    else if constexpr (BRD != 0 && F16) kern = conv3d_f16x3_d32_brd_kernel<NW, MW, WM, WN, TD, TH, TW, BRD>;
    else if constexpr (BRD != 0) kern = conv3d_bf16x3_d32_brd_kernel<NW, MW, WM, WN, TD, TH, TW, BRD>;
    else if constexpr (DSK && UPS && F16) kern = conv3d_f16x3_d32u_dk_kernel<NW, MW, WM, WN, TD, TH, TW>;
    else if constexpr (DSK && UPS) kern = conv3d_bf16x3_d32u_dk_kernel<NW, MW, WM, WN, TD, TH, TW>;
    else if constexpr (DSK && F16) kern = conv3d_f16x3_d32_dk_kernel<NW, MW, WM, WN, TD, TH, TW>;
    else if constexpr (DSK) kern = conv3d_bf16x3_d32_dk_kernel<NW, MW, WM, WN, TD, TH, TW>;
    else if constexpr (D32 && UPS && F16) kern = conv3d_f16x3_d32u_kernel<NW, MW, WM, WN, TD, TH, TW>;
    else if constexpr (D32 && UPS) kern = conv3d_bf16x3_d32u_kernel<NW, MW, WM, WN, TD, TH, TW>;
    else if constexpr (D32 && F16) kern = conv3d_f16x3_d32_kernel<NW, MW, WM, WN, TD, TH, TW>;
    else if constexpr (D32) kern = conv3d_bf16x3_d32_kernel<NW, MW, WM, WN, TD, TH, TW>;
    else if constexpr (F16) kern = conv3d_f16x3_kernel<NW, MW, WM, WN, TD, TH, TW, S, KD, UPS, PLANE, V32, WLDS>;
    else kern = conv3d_bf16x3_kernel<NW, MW, WM, WN, TD, TH, TW, S, KD, UPS, PLANE, V32, WLDS>;
    MVSGI_REQUIRE(!D32 || a.Cin % 32 == 0, "conv3d: the 32-channel-slice kernels need Cin %% 32 == 0 (got %d)", a.Cin);
    static mvsgi::PersistentGeom geo_cache[mvsgi::kMaxDevices] = {};
    mvsgi::PersistentGeom geo;
    if (mvsgi::persistent_geometry(kern, 512, lds_bytes, 2, geo_cache, "conv3d(bf16x3)", geo)) return 1;
    a.tiles_h = (int)mvsgi::cdiv(a.Ho, TH);
    a.tiles_w = (int)mvsgi::cdiv(a.Wo, TW);
    const int CT = a.Cout / (V32 ? 32 : 16);
    // The border-plane skip (conv3d_x3_body, BRD): the bricks at the bottom and at the top of the volume as launches of their own on
    // kernels that leave out the taps meeting the padding, the bricks between on this one -- where a layer of bricks is rounds of
    // the chip by itself (a few-round launch would pay three prologues and three tails for it)
    constexpr bool kBorderSplit = D32 && !DSK && BRD == 0 && TD == 2 && WM == 2 && MW == 4 && TH == 4 && TW == 16 && S == 1;
    if constexpr (kBorderSplit) {
        const long long layer = (long long)a.B * a.tiles_h * a.tiles_w * mvsgi::cdiv(CT, WN * NW);
        // (measured, tools/d32_probe.py: at D = 4 -- two launches, every brick a border brick -- 64 -> 64 [4,20,80] x 128 0.94 -> 0.90 of
        // the tap-pair kernel; at D = 8, 16 the third launch and the shorter walks cost what the skipped sixth of 2 of 4 / 2 of 8
        // bricks saves: 96 -> 96 [16,80,320] x 8 0.931 -> 0.964.  So: volumes four planes deep.)
        const char* dmax_ = mvsgi::exp_env("MVSGI_BSK_DMAX");
        if (a.od_cnt == 0 && a.Do >= 4 && a.Do <= (dmax_ ? atoi(dmax_) : 4) && a.Do % 2 == 0 && layer >= 4ll * geo.cus && !mvsgi::exp_env("MVSGI_NO_BSK")) {
            ConvArgs b = a;
            b.od_cnt = 1;
            b.od_off = 0;
            if (launch_bf16x3<NW, MW, WM, WN, TD, TH, TW, S, KD, UPS, PLANE, V32, WLDS, F16, D32, DSK, 1>(b, st)) return 1;
            b.od_off = a.Do - 2;
            if (launch_bf16x3<NW, MW, WM, WN, TD, TH, TW, S, KD, UPS, PLANE, V32, WLDS, F16, D32, DSK, 2>(b, st)) return 1;
            if (a.Do == 4) return 0;
            a.od_off = 2;
            a.od_cnt = a.Do / 2 - 2;
        }
    }
    a.tiles_d = a.od_cnt > 0 ? a.od_cnt : (int)mvsgi::cdiv(a.Do, TD);
    const long long nb = (long long)a.B * a.tiles_d * a.tiles_h * a.tiles_w * mvsgi::cdiv(CT, WN * NW);
    MVSGI_REQUIRE(nb < (1ll << 31), "conv3d: too many units");
    MVSGI_REQUIRE((long long)a.Din * a.Hin * a.Win * a.Cin < (1ll << 29), "conv3d: volume too large for 32-bit byte offsets");
    MVSGI_REQUIRE(!UPS || (a.Din % 2 == 0 && a.Hin % 2 == 0 && a.Win % 2 == 0), "conv3d: fused upsample needs even sizes");
    MVSGI_REQUIRE(!UPS || (long long)(a.Hin / 2) * (a.Win / 2) * a.Cin < (1ll << 23), "conv3d: fused upsample: low-resolution plane too large for 24-bit strides");
    MVSGI_REQUIRE((long long)a.Do * a.Ho * a.Wo * a.Cout < (1ll << 31), "conv3d: output frame too large for 32-bit element offsets");
    MVSGI_REQUIRE(!a.y_split || (!V32 && (long long)(a.Do + 2) * (a.Ho + 4) * (a.Wo + 4) * a.Cout * 4 < (1ll << 31)),
                  "conv3d: split-padded output not available for this kernel / size");
    a.total_units = (int)nb;
    MVSGI_SAT_WORDS(sat_words_);
    a.sat = sat_words_;
#ifdef MVSGI_STAMPS
    {   // stamps of the PREVIOUS launch are printed when MVSGI_STAMP=2
        static unsigned long long* dbgbuf = nullptr;
        const char* e_ = getenv("MVSGI_STAMP");
        if (e_ && !dbgbuf) { (void)hipMalloc(&dbgbuf, 8 * 128 * 8); (void)hipMemset(dbgbuf, 0, 8 * 128 * 8); }
        a.dbg = dbgbuf;
        if (e_ && atoi(e_) == 2 && dbgbuf) {
            static unsigned long long h[8 * 128];
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, dbgbuf, sizeof(h), hipMemcpyDeviceToHost);
            for (int w = 0; w < 8; ++w) {
                fprintf(stderr, "wave %d:", w);
                for (int i = 0; i < 120; ++i) fprintf(stderr, " %lld", (long long)(h[w * 128 + i] - h[0]));
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    // persistent grid (a multiple of 8 unless it covers every unit once): each workgroup walks the
    // units blockIdx.x + k*gridDim.x, which stay on its XCD's contiguous run of the index space
    // (the unit walk needs gridDim.x % 8 == 0 or gridDim.x == total: CU counts of whole-XCD devices are multiples of 8)
    const long long resident = ((long long)geo.cus * geo.wgs_per_cu) / 8 * 8 > 0 ? ((long long)geo.cus * geo.wgs_per_cu) / 8 * 8 : 8;
    const unsigned grid = (unsigned)(nb <= resident ? nb : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, st, a);
    return mvsgi::check_launch("mvsgi_conv3d_f32(bf16x3)");
}
