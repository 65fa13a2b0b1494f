// Shared by the Winograd-form kernels (csrc/conv3d_wino.hip: K2w, the level-0 residual convs; csrc/conv3d_wino_up2.hip: K3w, the
// polyphase ResizeConv3d): mixed-precision fma helpers on fp16 pairs, the split of a value pair, and the inline-asm MFMA forms that
// keep the weights in the accumulator half of the register file.  Include inside the translation unit's anonymous namespace, behind
// split_fmt.hpp.
#pragma once

// fp32 <- f16 half of a dword through the mixed-precision fma (one instruction where widen + add are two or three)
__device__ __forceinline__ float mix_sum_lo(unsigned h, unsigned l) {       // f16lo(h) + f16lo(l)
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
    return r;
}
__device__ __forceinline__ float mix_sum_hi(unsigned h, unsigned l) {       // f16hi(h) + f16hi(l)
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
    return r;
}
__device__ __forceinline__ float mix_fma_lo(unsigned h, float s, float c) {  // f16lo(h) * s + c
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(s), "v"(c));
    return r;
}
__device__ __forceinline__ float mix_fma_hi(unsigned h, float s, float c) {  // f16hi(h) * s + c
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(s), "v"(c));
    return r;
}
__device__ __forceinline__ float mix_add_lo(unsigned h, float c) {           // f16lo(h) + c
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ float mix_add_hi(unsigned h, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ float mix_sub_lo(float c, unsigned h) {           // c - f16lo(h)
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ float mix_sub_hi(float c, unsigned h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
    return r;
}
// (lo-half value, hi-half value) -> the split's packed hi and lo dwords (clamped to fp16's range first)
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo, float& satm) {
    v0 = sf_clamp<true>(v0);
    v1 = sf_clamp<true>(v1);
    satm = sf_sat_acc(satm, v0, v1);         // range report (csrc/split_fmt.hpp)
    hi = sf_cvt_pk<true>(v0, v1);
    lo = sf_cvt_pk<true>(mix_sub_lo(v0, hi), mix_sub_hi(v1, hi));
}

// The MFMAs are inline asm so that the weights can be pinned to the accumulator half of the register file ("a"); the output
// accumulators and the transformed activations are ordinary registers.  hipcc pads no hazards around asm: the step ends in WN_PAD
// before anything else may touch its accumulators, and the accumulators finished in a step get their last MFMAs in its first third.
#define WN_MF(ACC, WREG, XREG) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(WREG), "v"(XREG));
#define WN_MF0(ACC, WREG, XREG) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "+v"(ACC) : "a"(WREG), "v"(XREG));
// (accumulator slots 1 and 2 live in the accumulator file beside the weights -- 192 + 64 = all of it: with three slots in ordinary
// registers next to the V pairs (64) and the raw patches (64) hipcc shuttles ~100 registers per step through v_accvgpr moves)
#define WN_MFA(ACC, WREG, XREG) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "a"(WREG), "v"(XREG));
#define WN_MFA0(ACC, WREG, XREG) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "+a"(ACC) : "a"(WREG), "v"(XREG));
#define WN_PAD() asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

