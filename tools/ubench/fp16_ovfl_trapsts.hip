// Two hardware facilities the fp16 split could use instead of a v_med3 clamp + an explicit range test per element (gfx950):
//   (1) MODE.FP16_OVFL (hwreg MODE bit 23): "an overflowed fp16 result is clamped to +-MAX_FP16 instead of +-inf" -- does it apply
//       to v_cvt_pk_f16_f32 / v_cvt_f16_f32 ?
//   (2) TRAPSTS.EXCP (hwreg 3, bits 8:0; bit 3 = overflow): sticky per-wave IEEE exception flags, documented as accumulated
//       whatever EXCP_EN says -- are they, and does a saturated conversion under (1) still raise "overflow" ?
// Prints, for a list of inputs: the converted pair with and without FP16_OVFL, and TRAPSTS.EXCP before / after each conversion.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/fp16_ovfl_trapsts.hip -o tools/ubench/fp16_ovfl_trapsts
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// s_getreg / s_setreg immediates: (size - 1) << 11 | offset << 6 | id
#define HWREG(id, off, size) ((((size) - 1) << 11) | ((off) << 6) | (id))
#define HW_MODE 1
#define HW_TRAPSTS 3

__device__ __forceinline__ unsigned excp() { return __builtin_amdgcn_s_getreg(HWREG(HW_TRAPSTS, 0, 9)); }

__global__ void k(unsigned* out, const float* in, int n, int ovfl) {
    if (ovfl) __builtin_amdgcn_s_setreg(HWREG(HW_MODE, 23, 1), 1u);
    const unsigned mode = __builtin_amdgcn_s_getreg(HWREG(HW_MODE, 0, 32));
    __builtin_amdgcn_s_setreg(HWREG(HW_TRAPSTS, 0, 9), 0u);
    for (int i = 0; i < n; ++i) {
        const unsigned e0 = excp();
        float a = in[i], b = -in[i];
        unsigned pk;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n s_nop 4" : "=v"(pk) : "v"(a), "v"(b));
        const unsigned e1 = excp();
        __builtin_amdgcn_s_setreg(HWREG(HW_TRAPSTS, 0, 9), 0u);
        unsigned sc;
        asm volatile("v_cvt_f16_f32 %0, %1\n s_nop 4" : "=v"(sc) : "v"(a));
        const unsigned e2 = excp();
        __builtin_amdgcn_s_setreg(HWREG(HW_TRAPSTS, 0, 9), 0u);
        // the lo part the split would form: a - widen(hi), converted
        float hi_f;
        asm volatile("v_cvt_f32_f16 %0, %1\n s_nop 4" : "=v"(hi_f) : "v"(pk));
        float lo = a - hi_f;
        unsigned lo_h;
        asm volatile("v_cvt_f16_f32 %0, %1\n s_nop 4" : "=v"(lo_h) : "v"(lo));
        const unsigned e3 = excp();
        __builtin_amdgcn_s_setreg(HWREG(HW_TRAPSTS, 0, 9), 0u);
        if (threadIdx.x == 0) {
            out[8 * i + 0] = pk;
            out[8 * i + 1] = sc & 0xffffu;
            out[8 * i + 2] = e0;
            out[8 * i + 3] = e1;
            out[8 * i + 4] = e2;
            out[8 * i + 5] = e3;
            out[8 * i + 6] = lo_h & 0xffffu;
            out[8 * i + 7] = mode;
        }
    }
}

// only lane 37 of the wave overflows: is the flag wave-wide (any lane) ?
__global__ void k_lane(unsigned* out, int ovfl) {
    if (ovfl) __builtin_amdgcn_s_setreg(HWREG(HW_MODE, 23, 1), 1u);
    __builtin_amdgcn_s_setreg(HWREG(HW_TRAPSTS, 0, 9), 0u);
    float a = threadIdx.x == 37 ? 1.0e6f : 1.0f, b = 2.0f;
    unsigned pk;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n s_nop 4" : "=v"(pk) : "v"(a), "v"(b));
    const unsigned e = excp();
    // an exec-masked-off lane must not raise it
    __builtin_amdgcn_s_setreg(HWREG(HW_TRAPSTS, 0, 9), 0u);
    unsigned pk2 = 0;
    if (threadIdx.x != 37) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n s_nop 4" : "=v"(pk2) : "v"(a), "v"(b));
    const unsigned e2 = excp();
    if (threadIdx.x == 37) { out[0] = pk; out[1] = e; out[2] = e2; }
    out[3 + threadIdx.x] = pk + pk2;
}

int main() {
    const float h_in[] = {1.0f, 65504.f, 65519.f, 65520.f, 70000.f, 1.0e6f, 3.0e38f, __builtin_inff(), 1.0e-8f, 6.0e-8f};
    const int n = sizeof(h_in) / sizeof(float);
    float* d_in;
    unsigned* d_out;
    hipMalloc(&d_in, sizeof(h_in));
    hipMalloc(&d_out, 8 * n * 4 + 1024);
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    unsigned h[8 * 16];
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_in, n, ovfl);
        hipMemcpy(h, d_out, 8 * n * 4, hipMemcpyDeviceToHost);
        printf("FP16_OVFL = %d   (MODE = 0x%08x)\n", ovfl, h[7]);
        for (int i = 0; i < n; ++i)
            printf("  x = %-12g  cvt_pk(x, -x) = 0x%08x  cvt(x) = 0x%04x  lo = 0x%04x   EXCP before 0x%03x  after pk 0x%03x  after cvt 0x%03x  after lo 0x%03x\n",
                   h_in[i], h[8 * i], h[8 * i + 1], h[8 * i + 6], h[8 * i + 2], h[8 * i + 3], h[8 * i + 4], h[8 * i + 5]);
        hipLaunchKernelGGL(k_lane, dim3(1), dim3(64), 0, 0, d_out, ovfl);
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        printf("  one lane of 64 overflows: pk = 0x%08x EXCP = 0x%03x ; that lane masked off: EXCP = 0x%03x\n", h[0], h[1], h[2]);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
