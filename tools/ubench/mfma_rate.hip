// Microbenchmark: cycles per v_mfma_f32_16x16x32_bf16 for the accumulate patterns the conv kernel can use.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const bf16x8* in, float* out, unsigned long long* t) {
    bf16x8 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256], b[4];
    for (int i = 0; i < 4; ++i) b[i] = in[threadIdx.x + 512 + 256 * i];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {            // accumulator-major: 3 dependent MFMAs back to back
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[i & 3], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[i & 3], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[(i + 1) & 3], acc[i], 0, 0, 0);
            }
        } else if (MODE == 1) {     // term-major: dependent MFMAs 8 apart
#pragma unroll
            for (int tr = 0; tr < 3; ++tr)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr == 1 ? a1 : a0, b[(i + tr) & 3], acc[i], 0, 0, 0);
        } else {                    // pairs alternating (distance 2)
#pragma unroll
            for (int i = 0; i < 8; i += 2)
#pragma unroll
                for (int tr = 0; tr < 3; ++tr) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr == 1 ? a1 : a0, b[(i + tr) & 3], acc[i], 0, 0, 0);
                    acc[i + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr == 1 ? a1 : a0, b[(i + tr) & 3], acc[i + 1], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

int main() {
    bf16x8* in; float* out; unsigned long long* t;
    hipMalloc(&in, 2048 * 16); hipMemset(in, 0x3c, 2048 * 16);
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&t, 1024 * 8);
    for (int blocks : {256, 512}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, in, out, t);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out, t);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, in, out, t);
                hipDeviceSynchronize();
            }
            unsigned long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
            printf("blocks %d mode %d: %.2f cycles per MFMA (64 x 24 MFMAs per wave)\n", blocks, mode, (double)h[3] / (64.0 * 24));
        }
    }
    return 0;
}
