// Does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs on gfx950 (or flush them to zero)?
// A = 2^-20 (fp16 subnormal 0x0010) everywhere, B = 1024: every D element = 32 * 2^-10 = 0.03125 when subnormals are honoured, 0 when
// flushed.  Second row of the report: the same with the normal number 2^-14.  Also prints the result of the float -> half conversions
// the split uses (RNE packed, overflow to inf?).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f16_denorm.hip -o tools/ubench/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void k(float* out, unsigned short abits, float x0, float x1) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = __builtin_bit_cast(_Float16, abits);
        b[j] = (_Float16)1024.0f;
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = c[0];
        f32x2 v = {x0, x1};
        half2v h = __builtin_convertvector(v, half2v);
        out[1] = (float)h[0];
        out[2] = (float)h[1];
        float lo0 = x0 - (float)h[0];
        out[3] = lo0;
        out[4] = (float)(_Float16)lo0;
    }
}

int main() {
    float* d;
    hipMalloc(&d, 64);
    float h[8];
    for (unsigned short bits : {(unsigned short)0x0010, (unsigned short)0x0400}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, bits, 0.1f, 70000.f);
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("A bits 0x%04x: D[0] = %g (expect %g if honoured)   cvt(0.1) = %.9g cvt(70000) = %g  lo(0.1) = %.9g -> half %.9g\n", bits, h[0],
               bits == 0x0010 ? 0.03125 : 32 * 1024.0 * 6.103515625e-05, h[1], h[2], h[3], h[4]);
    }
    return 0;
}
