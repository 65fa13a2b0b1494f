// Layout helpers for the module boundary: [B][C][V] <-> [B][V][C] (V = D*H*W), fp32.
// The hot path itself is channels-last end to end; these only run when a caller hands the
// regulator a contiguous NCDHW tensor or asks for one back.  64x(C<=64 per pass) LDS tile
// transposes so both the read and the write side are coalesced.
#include "common.hpp"

namespace {

constexpr int TV = 64;   // voxels per tile
constexpr int TC = 16;   // channels per tile

// x [B][C][V] -> y [B][V][C]
__global__ __launch_bounds__(256) void ncv_to_nvc_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                         long long V) {
    __shared__ float tile[TC][TV + 1];
    const int b = blockIdx.z;
    const long long v0 = (long long)blockIdx.x * TV;
    const int c0 = blockIdx.y * TC;
    const float* xb = x + (long long)b * C * V;
    float* yb = y + (long long)b * C * V;
    for (int e = threadIdx.x; e < TC * TV; e += 256) {
        const int c = e / TV, v = e % TV;
        if (c0 + c < C && v0 + v < V) tile[c][v] = xb[(long long)(c0 + c) * V + v0 + v];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TC * TV; e += 256) {
        const int v = e / TC, c = e % TC;
        if (c0 + c < C && v0 + v < V) yb[(v0 + v) * C + c0 + c] = tile[c][v];
    }
}

// x [B][V][C] -> y [B][C][V]
__global__ __launch_bounds__(256) void nvc_to_ncv_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                         long long V) {
    __shared__ float tile[TC][TV + 1];
    const int b = blockIdx.z;
    const long long v0 = (long long)blockIdx.x * TV;
    const int c0 = blockIdx.y * TC;
    const float* xb = x + (long long)b * C * V;
    float* yb = y + (long long)b * C * V;
    for (int e = threadIdx.x; e < TC * TV; e += 256) {
        const int v = e / TC, c = e % TC;
        if (c0 + c < C && v0 + v < V) tile[c][v] = xb[(v0 + v) * C + c0 + c];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TC * TV; e += 256) {
        const int c = e / TV, v = e % TV;
        if (c0 + c < C && v0 + v < V) yb[(long long)(c0 + c) * V + v0 + v] = tile[c][v];
    }
}

int check(const float* x, float* y, int B, int C, long long V, const char* who) {
    MVSGI_REQUIRE(x && y, "%s: null pointer", who);
    MVSGI_REQUIRE(B > 0 && B < 65536 && C > 0 && V > 0, "%s: bad dims", who);
    MVSGI_REQUIRE(mvsgi::cdiv(C, TC) < 65536 && mvsgi::cdiv(V, TV) < (1ll << 31), "%s: grid too large", who);
    return 0;
}

}  // namespace

extern "C" int mvsgi_ncv_to_nvc_f32(const float* x, float* y, int B, int C, long long V, mvsgi_stream_t stream) {
    if (check(x, y, B, C, V, "mvsgi_ncv_to_nvc_f32")) return 1;
    dim3 grid((unsigned)mvsgi::cdiv(V, TV), (unsigned)mvsgi::cdiv(C, TC), (unsigned)B);
    hipLaunchKernelGGL(ncv_to_nvc_kernel, grid, dim3(256), 0, mvsgi::as_stream(stream), x, y, C, V);
    return mvsgi::check_launch("mvsgi_ncv_to_nvc_f32");
}

extern "C" int mvsgi_nvc_to_ncv_f32(const float* x, float* y, int B, int C, long long V, mvsgi_stream_t stream) {
    if (check(x, y, B, C, V, "mvsgi_nvc_to_ncv_f32")) return 1;
    dim3 grid((unsigned)mvsgi::cdiv(V, TV), (unsigned)mvsgi::cdiv(C, TC), (unsigned)B);
    hipLaunchKernelGGL(nvc_to_ncv_kernel, grid, dim3(256), 0, mvsgi::as_stream(stream), x, y, C, V);
    return mvsgi::check_launch("mvsgi_nvc_to_ncv_f32");
}
