// What does `buffer_load_dwordx4 ... lds` do (a) with an LDS destination beyond 64 KB and (b) with out-of-range lanes?
// hipcc --offload-arch=gfx950 -O2 ldsdma_probe.hip -o ldsdma_probe && ./ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
__global__ void k(const unsigned* src, unsigned* out, int dst_off, int records) {
    const int lane = threadIdx.x;
    // poison 160 KB
    for (int i = lane; i < 40960; i += 64) reinterpret_cast<unsigned*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, records, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + dst_off), 16, lane * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // report: words at dst_off.., and at (dst_off & 0xffff)..
    for (int i = lane; i < 256; i += 64) {
        out[i] = reinterpret_cast<unsigned*>(lds + dst_off)[i];
        out[256 + i] = reinterpret_cast<unsigned*>(lds + (dst_off & 0xffff))[i];
    }
}
int main() {
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
    unsigned *d, *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const int cases[][2] = {{4096, 4096}, {70000 / 16 * 16, 4096}, {131072, 4096}, {4096, 512}, {70000 / 16 * 16, 0}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 163840, 0, d, o, c[0], c[1]);
        std::vector<unsigned> r(512);
        hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
        printf("dst %6d records %4d: at dst: %08x %08x .. [128]=%08x [255]=%08x | at dst&0xffff: %08x %08x [128]=%08x\n", c[0], c[1], r[0], r[1],
               r[128], r[255], r[256], r[257], r[256 + 128]);
    }
    return 0;
}
