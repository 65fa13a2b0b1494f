// Sampling-grid generator for the spherical sweep: the closed forms of
// dsta_mvs/support/dataset/torch_cuda_sweep.py as four element-wise kernels, composed by the host
// exactly as MultiViewCameraModelDataset.make_sweep_grid_cuda does (multi_view_camera_model_dataset.py:
// 474-521): candidate rays of the cost-volume camera -> inverse camera pose -> projection into the
// camera's image, normalised to [-1, 1] for grid_sample, plus the validity mask.
//
// Runs once per rig (the grids are constants of the path), so these are plain one-thread-per-point
// kernels; fp32 throughout with contraction off, operation order as in the reference's torch
// expressions (the device sin/cos/atan2/sqrt differ from the host libm by an ulp or two, which is
// the parity tolerance).
#include "common.hpp"

namespace {

// RayMaker_UEPanorama.make_rays_for_candidates (torch_cuda_sweep.py:76-132): rays [3][N][H][W] of the
// panorama frame (z backward, x left, y down) for N candidate distances on an H x W equirect grid.
__global__ __launch_bounds__(256) void rays_panorama_kernel(const float* __restrict__ dist, float* __restrict__ rays,
                                                            int N, int H, int W, float lat0, float lat_span,
                                                            float lon0, float lon_span) {
#pragma clang fp contract(off)
    const long long total = (long long)N * H * W;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % W), h = (int)((idx / W) % H), n = (int)(idx / ((long long)W * H));
    // :91-92, :99-100  (pixel centres: (i + 0.5) / size * span + start)
    const float phi = (((float)h + 0.5f) / (float)H * lat_span) + lat0;
    const float theta = (((float)w + 0.5f) / (float)W * lon_span) + lon0;
    const float d = dist[n];
    const float sp = sinf(phi), cp = cosf(phi), st = sinf(theta), ct = cosf(theta);   // :120-123
    const float ds = d * sp;                       // :125
    rays[idx] = ds * ct;                           // x  :126
    rays[total + idx] = -d * cp;                   // y  :127
    rays[2 * total + idx] = -ds * st;              // z  :128
}

// transform_3D_points_torch (torch_cuda_sweep.py:385-408): p' = R p + t per batch element,
// points [B][3][M], T [B][4][4] row-major.
__global__ __launch_bounds__(256) void transform_points_kernel(const float* __restrict__ T, const float* __restrict__ p,
                                                               float* __restrict__ q, int B, long long M) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * M) return;
    const int b = (int)(idx / M);
    const long long m = idx - (long long)b * M;
    const float* t = T + b * 16;
    const float* pb = p + (long long)b * 3 * M;
    const float x = pb[m], y = pb[M + m], z = pb[2 * M + m];
    float* qb = q + (long long)b * 3 * M;
#pragma unroll
    for (int i = 0; i < 3; ++i)   // matmul row (k-ordered accumulation) + translation
        qb[i * M + m] = fmaf(t[i * 4 + 2], z, fmaf(t[i * 4 + 1], y, t[i * 4 + 0] * x)) + t[i * 4 + 3];
}

struct DsParams {
    float xi, alpha, one_minus_alpha, fx, fy, cx, cy, wm1, hm1, neg_w2;
};

// DoubleSphereSampleGridMaker.make_grid (torch_cuda_sweep.py:262-298): points [B][3][M] ->
// grid [B][M][2] in [-1, 1] and mask [B][M] (1 = inside the model's field of view).
__global__ __launch_bounds__(256) void grid_double_sphere_kernel(const float* __restrict__ p, float* __restrict__ grid,
                                                                 unsigned char* __restrict__ mask, int B, long long M,
                                                                 DsParams c) {
#pragma clang fp contract(off)
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * M) return;
    const int b = (int)(idx / M);
    const long long m = idx - (long long)b * M;
    const float* pb = p + (long long)b * 3 * M;
    const float x = pb[m], y = pb[M + m], z = pb[2 * M + m];
    const float x2 = x * x, y2 = y * y, z2 = z * z;                      // :276-278
    const float d1 = sqrtf((x2 + y2) + z2);                             // :280
    const float s = c.xi * d1 + z;
    const float d2 = sqrtf((x2 + y2) + s * s);                          // :281
    const float t = c.alpha * d2 + c.one_minus_alpha * s;               // :283
    const float ux = ((c.fx / t * x + c.cx) / c.wm1) * 2.0f - 1.0f;     // :287
    const float uy = ((c.fy / t * y + c.cy) / c.hm1) * 2.0f - 1.0f;     // :288
    grid[idx * 2] = ux;
    grid[idx * 2 + 1] = uy;
    mask[idx] = z > c.neg_w2 * d1 ? 1 : 0;                              // :295
}

// EquirectangularSampleGridMaker.make_grid (torch_cuda_sweep.py:305-335): points [B][3][M] -> grid [B][M][2].
__global__ __launch_bounds__(256) void grid_equirect_kernel(const float* __restrict__ p, float* __restrict__ grid, int B,
                                                            long long M, float pi_f) {
#pragma clang fp contract(off)
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * M) return;
    const int b = (int)(idx / M);
    const long long m = idx - (long long)b * M;
    const float* pb = p + (long long)b * 3 * M;
    const float x = pb[m], y = pb[M + m], z = pb[2 * M + m];
    const float xz = sqrtf(x * x + z * z);                              // :316-320
    const float lon = -1.0f * atan2f(z, x);                             // :325
    const float lat = atan2f(y, xz);                                    // :326
    grid[idx * 2] = lon / pi_f;                                         // :331
    grid[idx * 2 + 1] = (2.0f * lat) / pi_f;                            // :332
}

inline int blocks_for(long long n, unsigned* out) {
    const long long nb = mvsgi::cdiv(n, 256);
    MVSGI_REQUIRE(n > 0 && nb < (1ll << 31), "grid generator: bad element count %lld", n);
    *out = (unsigned)nb;
    return 0;
}

}  // namespace

extern "C" int mvsgi_rays_panorama_f32(const float* dist, float* rays, int N, int H, int W, float lat0, float lat1,
                                       float lon0, float lon1, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(dist && rays, "mvsgi_rays_panorama_f32: null pointer");
    MVSGI_REQUIRE(N > 0 && H > 0 && W > 0, "mvsgi_rays_panorama_f32: non-positive dimension");
    unsigned nb;
    if (blocks_for((long long)N * H * W, &nb)) return 1;
    // the spans are the reference's Python-float differences, applied to fp32 tensors as fp32 scalars
    const float lat_span = (float)((double)lat1 - (double)lat0), lon_span = (float)((double)lon1 - (double)lon0);
    hipLaunchKernelGGL(rays_panorama_kernel, dim3(nb), dim3(256), 0, mvsgi::as_stream(stream), dist, rays, N, H, W, lat0,
                       lat_span, lon0, lon_span);
    return mvsgi::check_launch("mvsgi_rays_panorama_f32");
}

extern "C" int mvsgi_transform_points_f32(const float* T, const float* points, float* out, int B, long long M,
                                          mvsgi_stream_t stream) {
    MVSGI_REQUIRE(T && points && out, "mvsgi_transform_points_f32: null pointer");
    MVSGI_REQUIRE(B > 0 && M > 0, "mvsgi_transform_points_f32: non-positive dimension");
    unsigned nb;
    if (blocks_for((long long)B * M, &nb)) return 1;
    hipLaunchKernelGGL(transform_points_kernel, dim3(nb), dim3(256), 0, mvsgi::as_stream(stream), T, points, out, B, M);
    return mvsgi::check_launch("mvsgi_transform_points_f32");
}

extern "C" int mvsgi_grid_double_sphere_f32(const float* points, float* grid, unsigned char* mask, int B, long long M,
                                            float xi, float alpha, float fx, float fy, float cx, float cy, int calib_h,
                                            int calib_w, float w2, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(points && grid && mask, "mvsgi_grid_double_sphere_f32: null pointer");
    MVSGI_REQUIRE(B > 0 && M > 0 && calib_h > 1 && calib_w > 1, "mvsgi_grid_double_sphere_f32: bad dimension");
    unsigned nb;
    if (blocks_for((long long)B * M, &nb)) return 1;
    DsParams c{xi, alpha, (float)(1.0 - (double)alpha), fx, fy, cx, cy, (float)(calib_w - 1), (float)(calib_h - 1), -w2};
    hipLaunchKernelGGL(grid_double_sphere_kernel, dim3(nb), dim3(256), 0, mvsgi::as_stream(stream), points, grid, mask, B,
                       M, c);
    return mvsgi::check_launch("mvsgi_grid_double_sphere_f32");
}

extern "C" int mvsgi_grid_equirect_f32(const float* points, float* grid, int B, long long M, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(points && grid, "mvsgi_grid_equirect_f32: null pointer");
    MVSGI_REQUIRE(B > 0 && M > 0, "mvsgi_grid_equirect_f32: non-positive dimension");
    unsigned nb;
    if (blocks_for((long long)B * M, &nb)) return 1;
    hipLaunchKernelGGL(grid_equirect_kernel, dim3(nb), dim3(256), 0, mvsgi::as_stream(stream), points, grid, B, M,
                       3.14159274101257324f /* float32(np.pi) */);
    return mvsgi::check_launch("mvsgi_grid_equirect_f32");
}
