// What does one instruction placed behind a v_mfma_f32_16x16x32_bf16 cost a single wave per SIMD?  (calibration of the
// hand-interleaved schedule of csrc/conv3d_rs.hip)   hipcc --offload-arch=gfx950 -O3 mfma_filler.hip -o mfma_filler
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

#define MFA(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "a"(W), "v"(X));
#define MFV(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(W), "v"(X));
#define MFAV(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "a"(W), "v"(X));
#define MFVV(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(W), "v"(X));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* src, float* out, unsigned long long* ticks, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 w[8], x[8];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) { w[i] = src[lane + 64 * i]; x[i] = src[lane + 64 * (8 + i)]; acc[i] = f32x4{0, 0, 0, 0}; }
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    float f0 = lane, f1 = lane * 2.f, f2 = 1.f, f3 = 3.f;
    bf16x8 r0 = x[0], r1 = x[1];
    const int ra = lane * 16;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            bf16x8 xn[3];
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                if (MODE == 15 || MODE == 16 || MODE == 17) { MFAV(acc[a], w[g], x[a]) } else if (MODE == 10) { MFV(acc[a], w[g], x[a]) } else if (MODE == 11) { MFVV(acc[a], w[g], x[a]) } else { MFA(acc[a], w[g], x[a]) }
                if (MODE == 16) { f0 = f0 + f2; f1 = f1 * f3; }
                if (MODE == 17) { f0 = f0 + f2; f1 = f1 * f3; f2 = f2 + f3; f3 = f3 * f0; }
                if (MODE == 1) { f0 = f0 + f2; }                                   // one independent-ish VALU (chain across MFMAs)
                if (MODE == 2) { f0 = f0 + f2; f1 = f1 * f3; }                      // two VALU
                if (MODE == 3) { f0 = f0 + f2; f1 = f1 * f3; f2 = f2 + f3; }        // three VALU
                if (MODE == 4 && a % 3 == 0) { r0 = *reinterpret_cast<const bf16x8*>(lds + ra + (g * 8 + a) * 1024 % 32768); asm volatile("" : "+v"(r0)); }
                if (MODE == 5) { if (a % 3 == 0) { r0 = *reinterpret_cast<const bf16x8*>(lds + ra + (g * 8 + a) * 1024 % 32768); asm volatile("" : "+v"(r0)); } f0 = f0 + f2; }
                if (MODE == 12 || MODE == 13 || MODE == 14) {      // the real pattern: fragments for the NEXT group, waited for a group later
                    if (a % 3 == 0) xn[a / 3] = *reinterpret_cast<const bf16x8*>(lds + ra + ((g * 8 + a) * 1024) % 32768);
                    if (MODE >= 13) f0 = f0 + f2;
                    if (MODE == 14) f1 = f1 * f3;
                }
                if (MODE == 6) { asm volatile("s_nop 0"); }
                if (MODE == 7) { asm volatile("v_mov_b32 %0, %1" : "=v"(f1) : "v"(f0)); }
                if (MODE == 8) { unsigned u = __builtin_bit_cast(unsigned, f0), v = __builtin_bit_cast(unsigned, f1);
                                 auto s = __builtin_amdgcn_permlane16_swap(u, v, false, false); f0 = __builtin_bit_cast(float, s[0]); f1 = __builtin_bit_cast(float, s[1]); }
                if (MODE == 9) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(iters)); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE >= 12 && MODE <= 14) { x[0] = xn[0]; x[3] = xn[1]; x[6] = xn[2]; }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = f0 + f1 + f2 + (float)r0[0] + (float)r1[0];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, const bf16x8* d, float* o, unsigned long long* t, int blocks) {
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 65536, 0, d, o, t, iters);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-44s blocks %4d: %.2f ticks per MFMA\n", name, blocks, (double)h / (iters * 64.0));
}

int main() {
    bf16x8* d; float* o; unsigned long long* t;
    hipMalloc(&d, 64 * 16 * 16); hipMalloc(&o, 1024 * 256 * 4); hipMalloc(&t, 8);
    std::vector<unsigned short> h(64 * 16 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f80 + (i * 7919 % 64);       // bf16 values around 1..1.5
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int blocks : {256}) {
        run<0>("bare (A operand in AGPRs)", d, o, t, blocks);
        run<10>("bare (A operand in VGPRs)", d, o, t, blocks);
        run<11>("bare (A, C/D in VGPRs)", d, o, t, blocks);
        run<15>("bare (A in AGPRs, C/D in VGPRs)", d, o, t, blocks);
        run<16>("  the same + 2 VALU per MFMA", d, o, t, blocks);
        run<17>("  the same + 4 VALU per MFMA", d, o, t, blocks);
        run<1>("+1 v_add per MFMA", d, o, t, blocks);
        run<2>("+2 VALU per MFMA", d, o, t, blocks);
        run<3>("+3 VALU per MFMA", d, o, t, blocks);
        run<4>("+1 ds_read_b128 per 3 MFMAs", d, o, t, blocks);
        run<5>("+1 ds_read_b128 per 3 + 1 v_add per MFMA", d, o, t, blocks);
        run<6>("+1 s_nop 0 per MFMA", d, o, t, blocks);
        run<7>("+1 v_mov per MFMA", d, o, t, blocks);
        run<8>("+1 permlane16_swap per MFMA", d, o, t, blocks);
        run<12>("+1 ds_read_b128 per 3 MFMAs, used a group later", d, o, t, blocks);
        run<13>("  the same + 1 VALU per MFMA", d, o, t, blocks);
        run<14>("  the same + 2 VALU per MFMA", d, o, t, blocks);
    }
    return 0;
}
