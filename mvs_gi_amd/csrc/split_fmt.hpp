// The two 16-bit splits of an fp32 operand, x = hi + lo, shared by every kernel that reads or writes split activations / weights.
//   F16 == false  "bf16x3":  hi = bf16(x) (RNE), lo = bf16(x - hi): 8 + 8 significant bits, fp32's range
//   F16 == true   "f16x3":   hi = fp16(x) (RNE), lo = fp16(x - hi): 11 + 11 significant bits; x is clamped to +-65504 first (fp16's
//                            range: never inf / nan from a finite input), lo parts below 2^-14 are fp16 subnormals, which the matrix
//                            cores of gfx950 honour (tools/ubench/mfma_f16_denorm.hip)
// Both are packed two elements per dword, low half first, and multiplied as hi*hi + hi*lo + lo*hi on the 16-bit matrix
// instructions of the same rate (v_mfma_f32_16x16x32_bf16 / _f16) with fp32 accumulation.  The bf16 forms are instruction for
// instruction what the kernels used before the fp16 split existed (the hand-placed schedules count on it).
// Include inside the translation unit's anonymous namespace.
#pragma once

typedef float sf_f32x4 __attribute__((ext_vector_type(4)));
typedef float sf_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned sf_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sf_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sf_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sf_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sf_f16x2 __attribute__((ext_vector_type(2)));

constexpr float kF16Max = 65504.f;

// the low / high 16-bit half of a packed dword as fp32: one VALU instruction each (shift | and; v_cvt_f32_f16 | its SDWA form)
template <bool F16>
__device__ __forceinline__ float sf_widen_lo(const unsigned u) {
    if constexpr (F16) return (float)__builtin_bit_cast(sf_f16x2, u)[0];
    else return __builtin_bit_cast(float, u << 16);
}
template <bool F16>
__device__ __forceinline__ float sf_widen_hi(const unsigned u) {
    if constexpr (F16) return (float)__builtin_bit_cast(sf_f16x2, u)[1];
    else return __builtin_bit_cast(float, u & 0xffff0000u);
}
// two fp32 values -> one packed dword, round to nearest even (v_cvt_pk_bf16_f32 | v_cvt_pk_f16_f32); no range handling
template <bool F16>
__device__ __forceinline__ unsigned sf_cvt_pk(const float a, const float b) {
    const sf_f32x2 v = {a, b};
    if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sf_f16x2));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sf_bf16x2));
}
// the range clamp of the fp16 split (identity for bf16): one v_med3_f32
template <bool F16>
__device__ __forceinline__ float sf_clamp(const float x) {
    if constexpr (F16) return __builtin_amdgcn_fmed3f(x, -kF16Max, kF16Max);
    else return x;
}
// ---- range report (include/mvsgi.h: mvsgi_saturation_flags) ----
// The fp16 split has a range the reference's fp32 does not (common_modules.py:105-115 computes in fp32, no clamp): every writer
// of fp16 pieces clamps to +-65504, the writers of the Winograd level's fp32-padded records to +-16376.  A clamp that ENGAGES is
// reported: each lane keeps the running maximum |value| it wrote (one v_max3_f32 per two elements, on the clamped values -- a
// value that reached the bound was clamped or sat exactly on it) and a wave that saw the bound stores 1 into the library's word
// of that kind -- pinned host memory (csrc/api.cpp), a plain store by one lane, no atomic, nothing on the normal path but the
// compare.  The words are sticky until the host clears them.
constexpr int kSatSweep = 0;       // the sweep's cost volume (the one un-normalised tensor of the path) left +-65504
constexpr int kSatSplit = 1;       // an activation a conv layer wrote or staged in fp16 pieces left +-65504
constexpr int kSatWino = 2;        // an activation of the Winograd level left +-16376 (or a transformed sum of four +-65504)
__device__ __forceinline__ float sf_sat_acc(const float m, const float a, const float b) {
    return __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(a)), __builtin_fabsf(b));       // v_max3_f32 m, |a|, |b|
}
__device__ __forceinline__ void sf_sat_report(unsigned* __restrict__ words, const int kind, const float m, const float bound) {
    if (__builtin_amdgcn_ballot_w64(m >= bound) != 0) {        // wave-uniform; never taken inside the range
        if ((threadIdx.x & 63) == 0) __builtin_nontemporal_store(1u, words + kind);
    }
}

// x = hi + lo for four fp32 values (clamped first in the fp16 split)
template <bool F16>
__device__ __forceinline__ void sf_split4(const sf_f32x4 x, sf_u32x2& hi, sf_u32x2& lo) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float a = sf_clamp<F16>(x[2 * p]), b = sf_clamp<F16>(x[2 * p + 1]);
        const unsigned hb = sf_cvt_pk<F16>(a, b);
        hi[p] = hb;
        lo[p] = sf_cvt_pk<F16>(a - sf_widen_lo<F16>(hb), b - sf_widen_hi<F16>(hb));
    }
}
// the same, keeping the lane's running maximum |clamped value| for the range report (fp16 split only)
template <bool F16>
__device__ __forceinline__ void sf_split4(const sf_f32x4 x, sf_u32x2& hi, sf_u32x2& lo, float& sat) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float a = sf_clamp<F16>(x[2 * p]), b = sf_clamp<F16>(x[2 * p + 1]);
        if constexpr (F16) sat = sf_sat_acc(sat, a, b);
        const unsigned hb = sf_cvt_pk<F16>(a, b);
        hi[p] = hb;
        lo[p] = sf_cvt_pk<F16>(a - sf_widen_lo<F16>(hb), b - sf_widen_hi<F16>(hb));
    }
}
template <bool F16>
__device__ __forceinline__ sf_f32x4 sf_join4(const sf_u32x2 hi, const sf_u32x2 lo) {
    sf_f32x4 r;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        r[2 * p] = sf_widen_lo<F16>(hi[p]) + sf_widen_lo<F16>(lo[p]);
        r[2 * p + 1] = sf_widen_hi<F16>(hi[p]) + sf_widen_hi<F16>(lo[p]);
    }
    return r;
}
// one matrix instruction of the split product; operands travel as 16-byte fragments whatever their element type
template <bool F16>
__device__ __forceinline__ sf_f32x4 sf_mfma16(const sf_bf16x8 a, const sf_bf16x8 b, const sf_f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sf_f16x8, a), __builtin_bit_cast(sf_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// a weight -> (hi, lo) bit patterns of either split, on the device (pack kernels) and on the host (plan builders); no clamp: weights
// are pre-scaled by the caller
__host__ __device__ inline void sf_split_weight(const float v, const bool f16, unsigned short& hi, unsigned short& lo) {
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
__host__ __device__ inline float sf_16_to_f32(const unsigned short h, const bool f16) {
    if (f16) return (float)__builtin_bit_cast(_Float16, h);
    return __builtin_bit_cast(float, (unsigned)h << 16);
}
