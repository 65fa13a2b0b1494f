// ResizeConv3d (dsta_mvs/model/common/common_modules.py:332-355) in POLYPHASE form for Cin = 32, Cout = 16 -- out_costs.0 of the
// (16, 32) regulator (cost_volume_regulator/unet_regulator.py:52-60), the largest single layer of the G16V path.
//
//   conv3d(interpolate(x, x2, trilinear, align_corners=False), w, padding=1)          [common_modules.py:335-341, :97-101]
// is, for each of the 8 output phases (pd, ph, pw), an ordinary 3x3x3 convolution over the LOW-resolution tensor:
//   out[2 i + p] = sum_t (sum_k M[p, class(i)][t][k] w[k]) x[i + t - 1]   per axis,  W_eff = (M_d x M_h x M_w) w  in 3-D,
// with 3x3 matrices M that fold the 0.25 / 0.75 blend, ATen's source-index clamp and the conv's zero padding of the UPSAMPLED grid
// (dropin/polyphase.py states and tests the algebra on the CPU).  The plan built here (host code, float64) holds
//   * 16 weight sets (pd, d-class, ph) x [pw * 16 + cout] for the register-stationary 32 -> 32 kernel (csrc/conv3d_rs.hip, MODE 2),
//     which evaluates every cell with INTERIOR matrices along H and W (a wave's weights are fixed for the launch; the D class is
//     per wave = per plane),
//   * the difference to the true matrices on the H and W faces of the volume, where only the CENTRE tap of that axis changes:
//     9-tap "face" convolutions (K = 9 x 32, matrix cores, weights in registers) for the H faces and the W faces, and the cells on
//     both (the four edge lines) by a small fp32 kernel.  They write RAW pre-scale corrections into the output voxels; the main
//     kernel's epilogue adds them to its sums before scale / shift / LeakyReLU and overwrites them with the result.
// Nothing is upsampled, no blend is evaluated at run time, and the staging is the plain LDS-DMA copy of the level-0 layers.
#include "common.hpp"

#include <cstring>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#include "split_fmt.hpp"

enum { FIRST = 0, INT = 1, LAST = 2, ONLY = 3 };

// ------------------------------------------------------------------------------------------------------------------
// host: the algebra
// ------------------------------------------------------------------------------------------------------------------
struct M3 {
    double m[3][3];      // [t][k]
};

// coefficient of x[i + t - 1] in up[2 i + p + k - 1] (first principles: clamp + zero padding)
M3 axis_matrix(int p, int i, int n) {
    M3 M{};
    for (int k = 0; k < 3; ++k) {
        const int v = 2 * i + p + k - 1;
        if (v < 0 || v >= 2 * n) continue;
        const int m = v >> 1, q = v & 1;
        const int src[2] = {q == 0 ? m - 1 : m, q == 0 ? m : m + 1};
        const double wgt[2] = {q == 0 ? 0.25 : 0.75, q == 0 ? 0.75 : 0.25};
        for (int e = 0; e < 2; ++e) {
            int s_ = src[e] < 0 ? 0 : (src[e] > n - 1 ? n - 1 : src[e]);
            M.m[s_ - i + 1][k] += wgt[e];
        }
    }
    return M;
}
M3 class_matrix(int p, int cls) {
    const int n = cls == ONLY ? 1 : 4;
    const int i = cls == FIRST || cls == ONLY ? 0 : (cls == INT ? 1 : n - 1);
    return axis_matrix(p, i, n);
}
// class matrix minus the interior one, centre row only (the other rows multiply the zero border where they differ)
M3 face_delta(int p, int cls) {
    const M3 a = class_matrix(p, cls), b = class_matrix(p, INT);
    M3 d{};
    for (int k = 0; k < 3; ++k) d.m[1][k] = a.m[1][k] - b.m[1][k];
    return d;
}
// out[o][i][a][b][c] = sum_klm Md[a][k] Mh[b][l] Mw[c][m] w[o][i][k][l][m];  w, out: [16][32][27]
void fold(const M3& Md, const M3& Mh, const M3& Mw, const float* w, float* out) {
    for (int oi = 0; oi < 16 * 32; ++oi) {
        const float* ws = w + (size_t)oi * 27;
        double t1[27], t2[27], t3[27];
        for (int a = 0; a < 3; ++a)
            for (int l = 0; l < 3; ++l)
                for (int m = 0; m < 3; ++m) {
                    double s = 0;
                    for (int k = 0; k < 3; ++k) s += Md.m[a][k] * ws[k * 9 + l * 3 + m];
                    t1[a * 9 + l * 3 + m] = s;
                }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                for (int m = 0; m < 3; ++m) {
                    double s = 0;
                    for (int l = 0; l < 3; ++l) s += Mh.m[b][l] * t1[a * 9 + l * 3 + m];
                    t2[a * 9 + b * 3 + m] = s;
                }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                for (int c = 0; c < 3; ++c) {
                    double s = 0;
                    for (int m = 0; m < 3; ++m) s += Mw.m[c][m] * t2[a * 9 + b * 3 + m];
                    t3[a * 9 + b * 3 + c] = s;
                }
        for (int q = 0; q < 27; ++q) out[(size_t)oi * 27 + q] = (float)t3[q];
    }
}

struct Group {      // planes of one D class
    int cls, first, count;
};
std::vector<Group> groups_of(int n) {
    if (n == 1) return {{ONLY, 0, 1}};
    if (n == 2) return {{FIRST, 0, 1}, {LAST, 1, 1}};
    return {{FIRST, 0, 1}, {INT, 1, n - 2}, {LAST, n - 1, 1}};
}
struct Face {
    int idx, cls;
};
std::vector<Face> faces_of(int n) {
    if (n == 1) return {{0, ONLY}};
    return {{0, FIRST}, {n - 1, LAST}};
}

// ------------------------------------------------------------------------------------------------------------------
// plan layout
// ------------------------------------------------------------------------------------------------------------------
constexpr int kMagic = 0x50325032;
constexpr int kRoleInts = 32;
enum { R_WOFF = 0, R_NTAPS, R_TAP0, R_XBASE = 11, R_XS0, R_XSRUN, R_N0, R_NRUN, R_YS0, R_YSRUN, R_MFIRST, R_MLAST, R_YBASE0 = 20 };   // R_YBASE0 .. +7: one per phase
enum { ST_WRITE = 0, ST_RMW = 1, ST_SKIP = 2 };
constexpr size_t kFaceRoleWBytes = (size_t)9 * 8 * 2 * 64 * 16;      // [tap][phase][hi|lo][lane] bf16x8 = 147,456 B: the role's LDS image
constexpr size_t kEdgeSetFloats = (size_t)9 * 32 * 16;               // [tap (td, th)][ci][co]

struct PolyHeader {
    int magic, D, H, W;
    int n_roles, n_edge_cells, n_groups, max_tpf;
    long long off_main, off_facew, off_roles, off_edgew, off_edgecells, total;
    int sum_tpf;        // 16-cell tiles per frame over all face roles
    int pad0;
    long long off_roles_split;      // the same roles addressing a SPLIT-PADDED output [B][2D+2][2H+2][2W+2][64 B]
    long long off_wino;             // fp16 plans where csrc/conv3d_wino_up2.hip applies: its two row-phase weight sets, else 0
    long long off_wino_unscale;     // ... and their [ph 2][pw 2][16] inverse pre-scaling
    int pad[4];
};
static_assert(sizeof(PolyHeader) == 128, "header");

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

PolyHeader layout(int D, int H, int W) {
    PolyHeader h{};
    h.magic = kMagic;
    h.D = D; h.H = H; h.W = W;
    const int ng = (int)groups_of(D).size(), nfh = (int)faces_of(H).size(), nfw = (int)faces_of(W).size();
    h.n_groups = ng;
    h.n_roles = (nfh + nfw) * ng;            // (face, D class): a role evaluates all 8 phases of its cells
    h.n_edge_cells = nfh * nfw;
    size_t o = sizeof(PolyHeader);
    h.off_main = (long long)(o = align256(o));
    o += 16 * mvsgi::kRs32PackedBytes;
    h.off_facew = (long long)(o = align256(o));
    o += (size_t)h.n_roles * kFaceRoleWBytes;
    h.off_roles = (long long)(o = align256(o));
    o += (size_t)h.n_roles * kRoleInts * 4;
    h.off_roles_split = (long long)(o = align256(o));
    o += (size_t)h.n_roles * kRoleInts * 4;
    h.off_edgew = (long long)(o = align256(o));
    o += (size_t)ng * h.n_edge_cells * 8 * kEdgeSetFloats * 4;
    h.off_edgecells = (long long)(o = align256(o));
    o += (size_t)h.n_edge_cells * 4 * 4;
    if (mvsgi::wino_up2_applies(D, H, W)) {      // (room in every plan of such a geometry: the layout does not depend on the split)
        h.off_wino = (long long)(o = align256(o));
        o += 2 * mvsgi::kWinoUp2RoleBytes;
        h.off_wino_unscale = (long long)(o = align256(o));
        o += 2 * 32 * 4;
    }
    h.total = (long long)align256(o);
    int tpf = 0, sum = 0;
    for (const Group& g : groups_of(D)) {
        const int a = g.count * (int)mvsgi::cdiv(W, 16), b = g.count * (int)mvsgi::cdiv(H, 16);
        tpf = a > tpf ? a : tpf;
        tpf = b > tpf ? b : tpf;
        sum += nfh * a + nfw * b;
    }
    h.max_tpf = tpf;
    h.sum_tpf = sum;
    return h;
}

// face-role weights: wsrc [8 phases][16][32][27] (folded), taps[9] = indices into the 27 -> [tap][phase][hi|lo][lane][8]
void pack_face_weights(const float* wsrc, const int* taps, unsigned short* out, bool f16) {
    for (int t = 0; t < 9; ++t)
        for (int ph = 0; ph < 8; ++ph)
            for (int lane = 0; lane < 64; ++lane) {
                const int kg = lane >> 4, co = lane & 15;
                const size_t o = (((size_t)(t * 8 + ph) * 2) * 64 + lane) * 8;
                for (int e = 0; e < 8; ++e) {
                    const float v = wsrc[(((size_t)ph * 16 + co) * 32 + kg * 8 + e) * 27 + taps[t]];
                    sf_split_weight(v, f16, out[o + e], out[o + 64 * 8 + e]);
                }
            }
}

// ------------------------------------------------------------------------------------------------------------------
// device: face convolutions on the matrix cores.  grid = (chunks, roles), role = (face, D class); one workgroup of 8 waves per CU: the
// role's 9 taps x 8 phases of weight fragments (hi, lo: 144 KiB) sit in LDS for the whole launch and every wave walks 16-cell
// tiles along the role's run axis, evaluating ALL 8 output phases of a tile from one set of 18 activation fragments (the
// first version kept 2 phases' weights in registers and re-read the cells per (pd, ph) role: 89 % of its wave cycles were
// waits on those loads).  A = weights [16 couts][32 cin] from LDS, B = the cells' split-padded records straight from global
// memory (already hi | lo), D[cout 4 kg + r][cell] -> 16-byte stores.
// ------------------------------------------------------------------------------------------------------------------
struct FaceGrid {          // workgroups of role r: [first[r], first[r + 1]) of a flat grid (every launched workgroup has work:
    int n_roles;           // a workgroup needs the CU's whole LDS, so idle ones would queue for a CU only to exit)
    int first[13];
};
template <bool F16>
__global__ __launch_bounds__(512, 2) void up2_face_kernel(const unsigned char* __restrict__ x, const unsigned char* __restrict__ plan,
                                                          unsigned char* __restrict__ y, int B, long long x_frame, long long y_frame,
                                                          long long off_facew, long long off_roles, FaceGrid fg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fw_lds[];
    int role = 0;
#pragma unroll
    for (int r = 1; r < 12; ++r) role += (r < fg.n_roles && (int)blockIdx.x >= fg.first[r]) ? 1 : 0;
    const int chunk = (int)blockIdx.x - fg.first[role], chunks = fg.first[role + 1] - fg.first[role];
    const int* R = reinterpret_cast<const int*>(plan + off_roles) + role * kRoleInts;
    const int n0 = R[R_N0], nrun = R[R_NRUN];
    const int tiles_run = (nrun + 15) >> 4, tpf = n0 * tiles_run;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kg = lane >> 4;
    {   // the role's weights -> LDS (16-byte pieces, coalesced)
        const u32x4* src = reinterpret_cast<const u32x4*>(plan + off_facew) + (long long)R[R_WOFF];
        for (int i = tid; i < (int)(kFaceRoleWBytes / 16); i += 512) reinterpret_cast<u32x4*>(fw_lds)[i] = src[i];
    }
    __syncthreads();
    int tap[9], yb[8];
#pragma unroll
    for (int t = 0; t < 9; ++t) tap[t] = R[R_TAP0 + t];
#pragma unroll
    for (int p = 0; p < 8; ++p) yb[p] = R[R_YBASE0 + p];
    const int x_base = R[R_XBASE], x_s0 = R[R_XS0], x_srun = R[R_XSRUN];
    const int y_s0 = R[R_YS0], y_srun = R[R_YSRUN];
    const int m_first = R[R_MFIRST], m_last = R[R_MLAST];
    const int lane_x = (kg >> 1) * 64 + (kg & 1) * 16;           // this lane's 8 input channels: hi piece (lo piece 32 B on)
    const int total = B * tpf, stride = chunks * 8;
    for (int t = chunk * 8 + wave; t < total; t += stride) {
        const int rt = t % tiles_run;
        int q = t / tiles_run;
        const int o0 = q % n0, b = q / n0;
        const int run = rt * 16 + col;
        const bool ok = run < nrun;
        const __amdgpu_buffer_rsrc_t dx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(x) + (long long)b * x_frame, 0,
                                                                            (int)x_frame, 0x00020000);
        const unsigned xo = ok ? (unsigned)(x_base + o0 * x_s0 + run * x_srun + lane_x) : 0xffffff00u;
        u32x4 fh[9], fl[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            fh[k] = __builtin_amdgcn_raw_buffer_load_b128(dx, xo + (unsigned)tap[k], 0, 0);
            fl[k] = __builtin_amdgcn_raw_buffer_load_b128(dx, xo + (unsigned)tap[k] + 32u, 0, 0);
        }
        f32x4 acc[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the weight fragments are the same LDS words for every tile: launder the OFFSET per tile (an integer: a laundered pointer
        // loses its LDS address space and the reads become flat loads) or the compiler hoists all 144 fragment reads out of the
        // tile loop and spills them.  A tap's 16 fragments are requested one tap ahead; its 24 MFMAs run term-major over the 8
        // independent accumulators.
        int woff = lane * 16;
        asm volatile("" : "+v"(woff));
        bf16x8 wb[2][16];
#pragma unroll
        for (int i = 0; i < 16; ++i) wb[0][i] = *reinterpret_cast<const bf16x8*>(fw_lds + woff + i * 1024);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (k + 1 < 9) {
#pragma unroll
                for (int i = 0; i < 16; ++i) wb[(k + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(fw_lds + woff + ((k + 1) * 16 + i) * 1024);
            }
            const bf16x8 xh = __builtin_bit_cast(bf16x8, fh[k]), xl = __builtin_bit_cast(bf16x8, fl[k]);
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[p] = sf_mfma16<F16>(wb[k & 1][2 * p + 1], xh, acc[p]);
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[p] = sf_mfma16<F16>(wb[k & 1][2 * p], xl, acc[p]);
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[p] = sf_mfma16<F16>(wb[k & 1][2 * p], xh, acc[p]);
        }
        const int mode = run == 0 ? m_first : (run == nrun - 1 ? m_last : ST_WRITE);
        if (ok && mode != ST_SKIP) {
            unsigned char* yf = y + (long long)b * y_frame + (long long)o0 * y_s0 + (long long)run * y_srun + kg * 16;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                f32x4* dst = reinterpret_cast<f32x4*>(yf + yb[p]);
                *dst = mode == ST_RMW ? *dst + acc[p] : acc[p];
            }
        }
    }
}

// the edge lines (cells on an H face AND a W face), every plane, all 8 phases: fp32 dot products of 9 taps (td, th) x 32 channels.
// thread = (b, plane, edge cell, phase, cout); writes (the H-face roles accumulate onto it afterwards).
template <bool F16>
__global__ __launch_bounds__(256) void up2_edge_kernel(const unsigned char* __restrict__ x, const unsigned char* __restrict__ plan,
                                                       float* __restrict__ y, int B, int D, int H, int W, int n_cells, int n_groups,
                                                       long long off_edgew, long long off_edgecells, int ob) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)B * D * n_cells * 128;
    if (idx >= n) return;
    const int co = (int)(idx & 15), phase = (int)((idx >> 4) & 7);
    long long q = idx >> 7;
    const int e = (int)(q % n_cells);
    q /= n_cells;
    const int i_d = (int)(q % D), b = (int)(q / D);
    const int* cell = reinterpret_cast<const int*>(plan + off_edgecells) + e * 4;      // i_h, i_w
    const int i_h = cell[0], i_w = cell[1];
    const int g = n_groups == 1 ? 0 : (i_d == 0 ? 0 : (i_d == D - 1 ? n_groups - 1 : 1));
    const float* wt = reinterpret_cast<const float*>(plan + off_edgew) + ((size_t)(g * n_cells + e) * 8 + phase) * kEdgeSetFloats + co;
    const int Hp = H + 2, Wp = W + 2;
    const unsigned char* xb = x + ((((long long)b * (D + 2) + i_d + 1) * Hp + i_h + 1) * Wp + i_w + 1) * 128;
    float s = 0.f;
    for (int td = 0; td < 3; ++td)
        for (int th = 0; th < 3; ++th) {
            const unsigned char* xv = xb + ((long long)(td - 1) * Hp + (th - 1)) * Wp * 128;
            const float* wk = wt + (size_t)(td * 3 + th) * 32 * 16;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const unsigned* rec = reinterpret_cast<const unsigned*>(xv + sl * 64);      // [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15]
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const unsigned hi = rec[p], lo = rec[p + 8];
                    const float a0 = sf_widen_lo<F16>(hi) + sf_widen_lo<F16>(lo);
                    const float a1 = sf_widen_hi<F16>(hi) + sf_widen_hi<F16>(lo);
                    s = fmaf(a0, wk[(sl * 16 + 2 * p) * 16], s);
                    s = fmaf(a1, wk[(sl * 16 + 2 * p + 1) * 16], s);
                }
            }
        }
    const int pd = phase >> 2, ph = (phase >> 1) & 1, pw = phase & 1;
    // ob = 1: the output is the split-padded tensor (one-voxel border); the raw correction occupies the voxel's 64-byte record as fp32
    y[((((long long)b * (2 * D + 2 * ob) + 2 * i_d + pd + ob) * (2 * H + 2 * ob) + 2 * i_h + ph + ob) * (2 * W + 2 * ob) + 2 * i_w + pw + ob) * 16 + co] = s;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------
extern "C" size_t mvsgi_conv3d_up2_poly_plan_bytes(int D, int H, int W) {
    if (D < 1 || H < 1 || W < 1) return 0;
    return (size_t)layout(D, H, W).total;
}

// HOST function (no GPU call): w_oidhw_host [16][32][3][3][3] fp32 in host memory -> plan_host (mvsgi_conv3d_up2_poly_plan_bytes
// bytes, position independent: copy it to the device as it is).  D, H, W: the LOW-resolution size of the layer's input.
// _fmt: the folded weights in either split (0 = bf16, MVSGI_SPLIT_F16; the caller pre-scales the weights of an fp16 plan, see
// MVSGI_CONV_F16); the plan is run by the *_fmt launchers with the same fmt
extern "C" int mvsgi_conv3d_up2_poly_plan_fmt(const float* w_oidhw_host, void* plan_host, int D, int H, int W, int fmt) {
    MVSGI_REQUIRE(w_oidhw_host && plan_host, "mvsgi_conv3d_up2_poly_plan: null pointer");
    MVSGI_REQUIRE(fmt == 0 || fmt == MVSGI_SPLIT_F16, "mvsgi_conv3d_up2_poly_plan: fmt %d not in {0, MVSGI_SPLIT_F16}", fmt);
    const bool f16 = fmt != 0;
    MVSGI_REQUIRE(D > 0 && H > 0 && W > 0, "mvsgi_conv3d_up2_poly_plan: bad dims");
    MVSGI_REQUIRE((long long)(D + 2) * (H + 2) * (W + 2) * 128 < (1ll << 31) && (long long)D * H * W * 512 < (1ll << 31),
                  "mvsgi_conv3d_up2_poly_plan: frame too large for 32-bit offsets");
    const PolyHeader h = layout(D, H, W);
    unsigned char* P = static_cast<unsigned char*>(plan_host);
    memset(P, 0, (size_t)h.total);
    memcpy(P, &h, sizeof(h));
    const float* w = w_oidhw_host;
    std::vector<float> f0(16 * 32 * 27), w32(32 * 32 * 27);
    // ---- main sets: (pd, d class, ph) x [pw * 16 + co], interior matrices along H and W ----
    for (int pd = 0; pd < 2; ++pd)
        for (int cls = 0; cls < 4; ++cls)
            for (int ph = 0; ph < 2; ++ph) {
                for (int pw = 0; pw < 2; ++pw) {
                    fold(class_matrix(pd, cls), class_matrix(ph, INT), class_matrix(pw, INT), w, f0.data());
                    memcpy(w32.data() + (size_t)pw * 16 * 32 * 27, f0.data(), f0.size() * 4);
                }
                mvsgi::rs32_pack_weights_host(w32.data(), P + h.off_main + (size_t)((pd * 4 + cls) * 2 + ph) * mvsgi::kRs32PackedBytes, f16);
            }
    // ---- the main kernel in Winograd form (csrc/conv3d_wino_up2.hip; fp16 split): per row phase ph the 32 -> 32 layer
    //      [pw * 16 + co] with interior matrices along H and W and the ORIGINAL depth taps (the kernel upsamples along D itself) ----
    if (f16 && h.off_wino) {
        M3 ident{};
        for (int t = 0; t < 3; ++t) ident.m[t][t] = 1.0;
        for (int ph = 0; ph < 2; ++ph) {
            for (int pw = 0; pw < 2; ++pw) {
                fold(ident, class_matrix(ph, INT), class_matrix(pw, INT), w, f0.data());
                memcpy(w32.data() + (size_t)pw * 16 * 32 * 27, f0.data(), f0.size() * 4);
            }
            mvsgi::wino_up2_pack_role(w32.data(), reinterpret_cast<unsigned short*>(P + h.off_wino + (size_t)ph * mvsgi::kWinoUp2RoleBytes),
                                      reinterpret_cast<float*>(P + h.off_wino_unscale) + ph * 32);
        }
    }
    // ---- face roles ----
    const int Hp = H + 2, Wp = W + 2, Hh = 2 * H, Wh = 2 * W;
    const std::vector<Group> gs = groups_of(D);
    const std::vector<Face> fhs = faces_of(H), fws = faces_of(W);
    int* roles = reinterpret_cast<int*>(P + h.off_roles);
    int r = 0;
    std::vector<float> w8(8 * 16 * 32 * 27);
    int* roles_split = reinterpret_cast<int*>(P + h.off_roles_split);
    // a role is written twice: addressing the plain fp32 output (geometry 0) and the split-padded one (geometry 1: every voxel
    // index + 1, rows of 2W + 2, planes of (2H + 2) rows); the weights are shared
    auto emit = [&](const int* taps27, const int* tap_off, int x_base, int x_s0, int x_srun, int n0, int nrun, const int (*yb)[8],
                    const int* y_s0, const int* y_srun, int m_first, int m_last) {
        pack_face_weights(w8.data(), taps27, reinterpret_cast<unsigned short*>(P + h.off_facew + (size_t)r * kFaceRoleWBytes), f16);
        for (int geo = 0; geo < 2; ++geo) {
            int* R = (geo ? roles_split : roles) + (size_t)r * kRoleInts;
            R[R_WOFF] = (int)((size_t)r * kFaceRoleWBytes / 16);
            R[R_NTAPS] = 9;
            for (int t = 0; t < 9; ++t) R[R_TAP0 + t] = tap_off[t];
            R[R_XBASE] = x_base; R[R_XS0] = x_s0; R[R_XSRUN] = x_srun; R[R_N0] = n0; R[R_NRUN] = nrun;
            for (int p = 0; p < 8; ++p) R[R_YBASE0 + p] = yb[geo][p];
            R[R_YS0] = y_s0[geo]; R[R_YSRUN] = y_srun[geo]; R[R_MFIRST] = m_first; R[R_MLAST] = m_last;
        }
        ++r;
    };
    const int Hq[2] = {Hh, Hh + 2}, Wq[2] = {Wh, Wh + 2};      // rows / columns of the output tensor per geometry
    for (const Group& g : gs) {
        // H faces: (Md x delta_h x Mw_interior), taps (td, tw) at th = 1, run axis = w; the run ends are edge cells: the edge
        // kernel has written there, accumulate
        for (const Face& fh : fhs) {
            int yb[2][8], ys0[2], ysr[2];
            for (int p = 0; p < 8; ++p) {
                const int pd = p >> 2, ph = (p >> 1) & 1, pw = p & 1;
                fold(class_matrix(pd, g.cls), face_delta(ph, fh.cls), class_matrix(pw, INT), w, w8.data() + (size_t)p * 16 * 32 * 27);
                for (int geo = 0; geo < 2; ++geo)
                    yb[geo][p] = (((2 * g.first + pd + geo) * Hq[geo] + 2 * fh.idx + ph + geo) * Wq[geo] + geo) * 64 + pw * 64;
            }
            for (int geo = 0; geo < 2; ++geo) { ys0[geo] = 2 * Hq[geo] * Wq[geo] * 64; ysr[geo] = 128; }
            int taps27[9], tap_off[9];
            for (int td = 0; td < 3; ++td)
                for (int tw = 0; tw < 3; ++tw) {
                    taps27[td * 3 + tw] = td * 9 + 3 + tw;
                    tap_off[td * 3 + tw] = ((td - 1) * Hp * Wp + (tw - 1)) * 128;
                }
            emit(taps27, tap_off, (((g.first + 1) * Hp + fh.idx + 1) * Wp + 1) * 128, Hp * Wp * 128, 128, g.count, W, yb, ys0, ysr,
                 ST_RMW, ST_RMW);
        }
        // W faces: (Md x Mh_interior x delta_w), taps (td, th) at tw = 1, run axis = h; the run ends belong to the edge kernel
        // (which applies the H CLASS matrix there): skipped
        for (const Face& fw : fws) {
            int yb[2][8], ys0[2], ysr[2];
            for (int p = 0; p < 8; ++p) {
                const int pd = p >> 2, ph = (p >> 1) & 1, pw = p & 1;
                fold(class_matrix(pd, g.cls), class_matrix(ph, INT), face_delta(pw, fw.cls), w, w8.data() + (size_t)p * 16 * 32 * 27);
                for (int geo = 0; geo < 2; ++geo)
                    yb[geo][p] = (((2 * g.first + pd + geo) * Hq[geo] + ph + geo) * Wq[geo] + 2 * fw.idx + pw + geo) * 64;
            }
            for (int geo = 0; geo < 2; ++geo) { ys0[geo] = 2 * Hq[geo] * Wq[geo] * 64; ysr[geo] = 2 * Wq[geo] * 64; }
            int taps27[9], tap_off[9];
            for (int td = 0; td < 3; ++td)
                for (int th = 0; th < 3; ++th) {
                    taps27[td * 3 + th] = td * 9 + th * 3 + 1;
                    tap_off[td * 3 + th] = ((td - 1) * Hp + (th - 1)) * Wp * 128;
                }
            emit(taps27, tap_off, (((g.first + 1) * Hp + 1) * Wp + fw.idx + 1) * 128, Hp * Wp * 128, Wp * 128, g.count, H, yb, ys0, ysr,
                 ST_SKIP, ST_SKIP);
        }
    }
    MVSGI_REQUIRE(r == h.n_roles, "mvsgi_conv3d_up2_poly_plan: internal role count mismatch");
    // ---- edge cells: (Md x Mh_class x delta_w) over taps (td, th), [group][cell][phase][tap][ci][co] fp32 ----
    int* cells = reinterpret_cast<int*>(P + h.off_edgecells);
    float* ew = reinterpret_cast<float*>(P + h.off_edgew);
    int e = 0;
    for (const Face& fh : fhs)
        for (const Face& fw : fws) {
            cells[e * 4] = fh.idx;
            cells[e * 4 + 1] = fw.idx;
            for (size_t gi = 0; gi < gs.size(); ++gi)
                for (int phase = 0; phase < 8; ++phase) {
                    const int pd = phase >> 2, ph = (phase >> 1) & 1, pw = phase & 1;
                    fold(class_matrix(pd, gs[gi].cls), class_matrix(ph, fh.cls), face_delta(pw, fw.cls), w, f0.data());
                    float* dst = ew + ((gi * h.n_edge_cells + e) * 8 + phase) * kEdgeSetFloats;
                    for (int td = 0; td < 3; ++td)
                        for (int th = 0; th < 3; ++th)
                            for (int ci = 0; ci < 32; ++ci)
                                for (int co = 0; co < 16; ++co)
                                    dst[((size_t)(td * 3 + th) * 32 + ci) * 16 + co] = f0[((size_t)co * 32 + ci) * 27 + td * 9 + th * 3 + 1];
                }
            ++e;
        }
    return 0;
}
extern "C" int mvsgi_conv3d_up2_poly_plan(const float* w_oidhw_host, void* plan_host, int D, int H, int W) {
    return mvsgi_conv3d_up2_poly_plan_fmt(w_oidhw_host, plan_host, D, H, W, 0);
}

namespace {

// edges (writes), faces (the H-face roles accumulate on the edge cells), then the register-stationary main kernel; ob = 1: the
// output is a split-padded tensor [B][2D+2][2H+2][2W+2][64 B] (the corrections use its voxel records as fp32 until the main
// kernel overwrites them with the split result)
// form: 0 = the dispatcher's choice of main kernel, 1 = the direct register-stationary kernel always, 2 = the Winograd form always (A/B, tests)
int poly_launch(const void* x_split, const void* plan_dev, const float* scale, const float* shift, void* y, int ob, int B, int D, int H,
                int W, float neg_slope, int fmt, hipStream_t st, int form = 0) {
    MVSGI_REQUIRE(x_split && plan_dev && scale && shift && y, "mvsgi_conv3d_up2_poly: null pointer");
    MVSGI_REQUIRE(fmt == 0 || fmt == MVSGI_SPLIT_F16, "mvsgi_conv3d_up2_poly: fmt %d not in {0, MVSGI_SPLIT_F16}", fmt);
    MVSGI_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "mvsgi_conv3d_up2_poly: bad dims");
    MVSGI_REQUIRE(neg_slope >= 0.f && neg_slope <= 1.f, "mvsgi_conv3d_up2_poly: neg_slope %g not in [0, 1]", (double)neg_slope);
    MVSGI_REQUIRE((long long)(D + 2) * (H + 2) * (W + 2) * 128 < (1ll << 31) &&
                      (long long)(2 * D + 2) * (2 * H + 2) * (2 * W + 2) * 64 < (1ll << 31),
                  "mvsgi_conv3d_up2_poly: frame too large for 32-bit offsets");
    const PolyHeader h = layout(D, H, W);
    const unsigned char* P = static_cast<const unsigned char*>(plan_dev);
    const long long x_frame = (long long)(D + 2) * (H + 2) * (W + 2) * 128;
    const long long y_frame = (long long)(2 * D + 2 * ob) * (2 * H + 2 * ob) * (2 * W + 2 * ob) * 64;
    const long long n = (long long)B * D * h.n_edge_cells * 128;
    MVSGI_REQUIRE(mvsgi::cdiv(n, 256) < (1ll << 31), "mvsgi_conv3d_up2_poly: too many edge threads");
    if (fmt)
        hipLaunchKernelGGL(up2_edge_kernel<true>, dim3((unsigned)mvsgi::cdiv(n, 256)), dim3(256), 0, st, static_cast<const unsigned char*>(x_split), P,
                           static_cast<float*>(y), B, D, H, W, h.n_edge_cells, h.n_groups, h.off_edgew, h.off_edgecells, ob);
    else
        hipLaunchKernelGGL(up2_edge_kernel<false>, dim3((unsigned)mvsgi::cdiv(n, 256)), dim3(256), 0, st, static_cast<const unsigned char*>(x_split), P,
                           static_cast<float*>(y), B, D, H, W, h.n_edge_cells, h.n_groups, h.off_edgew, h.off_edgecells, ob);
    if (mvsgi::check_launch("mvsgi_conv3d_up2_poly(edges)")) return 1;
    // one workgroup per CU is resident (its LDS holds a role's weights): size the roles' workgroups so that all of them fit one round
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    static bool attr_set[mvsgi::kMaxDevices] = {};
    if (dev >= 0 && dev < mvsgi::kMaxDevices && !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(up2_face_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kFaceRoleWBytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(up2_face_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kFaceRoleWBytes);
        MVSGI_REQUIRE(e == hipSuccess, "mvsgi_conv3d_up2_poly: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set[dev] = true;
    }
    MVSGI_REQUIRE(h.n_roles <= 12, "mvsgi_conv3d_up2_poly: internal: %d face roles", h.n_roles);
    long long slots = (long long)cus - h.n_roles;     // every role rounds its share up
    if (slots < 8) slots = 8;
    long long tpw = mvsgi::cdiv((long long)B * h.sum_tpf, slots);
    if (tpw < 8) tpw = 8;                             // at least one tile per wave
    FaceGrid fg{};
    fg.n_roles = h.n_roles;
    {   // tiles per frame of every role, in the plan's role order (groups outermost; per group the H faces, then the W faces)
        int r = 0, acc = 0;
        for (const Group& g : groups_of(D)) {
            const int nfh = (int)faces_of(H).size(), nfw = (int)faces_of(W).size();
            for (int f = 0; f < nfh + nfw; ++f) {
                const long long tpf = (long long)g.count * mvsgi::cdiv(f < nfh ? W : H, 16);
                fg.first[r++] = acc;
                acc += (int)mvsgi::cdiv((long long)B * tpf, tpw);
            }
        }
        fg.first[r] = acc;
    }
    if (fmt)
        hipLaunchKernelGGL(up2_face_kernel<true>, dim3((unsigned)fg.first[h.n_roles]), dim3(512), kFaceRoleWBytes, st,
                           static_cast<const unsigned char*>(x_split), P, static_cast<unsigned char*>(y), B, x_frame, y_frame, h.off_facew,
                           ob ? h.off_roles_split : h.off_roles, fg);
    else
        hipLaunchKernelGGL(up2_face_kernel<false>, dim3((unsigned)fg.first[h.n_roles]), dim3(512), kFaceRoleWBytes, st,
                           static_cast<const unsigned char*>(x_split), P, static_cast<unsigned char*>(y), B, x_frame, y_frame, h.off_facew,
                           ob ? h.off_roles_split : h.off_roles, fg);
    if (mvsgi::check_launch("mvsgi_conv3d_up2_poly(faces)")) return 1;
    // the main kernel: in the fp16 split with a split-padded output, the Winograd form (2.25 x fewer matrix instructions) where it
    // applies and its units fill their rounds of one unit per CU (a unit occupies a CU for ~25 us)
    MVSGI_REQUIRE(form != 2 || (fmt && ob && h.off_wino), "mvsgi_conv3d_up2_poly: the Winograd form needs the fp16 split, a split-padded output, "
                  "D == 8, H %% 2 == 0 and W %% 32 == 0 (got D, H, W = %d, %d, %d)", D, H, W);
    if (form == 2)
        return mvsgi::wino_up2_launch(x_split, P + h.off_wino, reinterpret_cast<const float*>(P + h.off_wino_unscale), scale, shift, y,
                                      B, D, H, W, neg_slope, st);
    if (fmt && ob && form == 0 && h.off_wino) {
        const long long units = (long long)B * (H / 2) * (W / 32);
        long long nwalk = cus / 16;
        if (nwalk > mvsgi::cdiv(units, 8)) nwalk = mvsgi::cdiv(units, 8);
        if (nwalk < 1) nwalk = 1;
        const long long rounds = mvsgi::cdiv(mvsgi::cdiv(units, 8), nwalk);
        if (10 * units >= 7 * rounds * 8 * nwalk)
            return mvsgi::wino_up2_launch(x_split, P + h.off_wino, reinterpret_cast<const float*>(P + h.off_wino_unscale), scale, shift, y,
                                          B, D, H, W, neg_slope, st);
    }
    return mvsgi::rs32_up2_launch(x_split, P + h.off_main, scale, shift, y, ob, B, D, H, W, neg_slope, fmt != 0, st);
}

}  // namespace

// ResizeConv3d.forward (common_modules.py:332-355) for Cin = 32, Cout = 16, no skip: x split-padded [B][D+2][H+2][W+2][32]
// (low resolution), y fp32 [B][2D][2H][2W][16] = act(conv(up2(x)) * scale + shift); plan_dev: the plan of
// mvsgi_conv3d_up2_poly_plan(D, H, W) in device memory; neg_slope in [0, 1].
extern "C" int mvsgi_conv3d_up2_poly_f32(const void* x_split, const void* plan_dev, const float* scale, const float* shift, float* y,
                                         int B, int D, int H, int W, float neg_slope, mvsgi_stream_t stream) {
    return poly_launch(x_split, plan_dev, scale, shift, y, 0, B, D, H, W, neg_slope, 0, mvsgi::as_stream(stream));
}

// The same with the result in the split-padded format, [B][2D+2][2H+2][2W+2][16] (zero border, never written): the input of
// mvsgi_conv3d_head_split (the cost head reads hi | lo fragments straight from it).
extern "C" int mvsgi_conv3d_up2_poly_split(const void* x_split, const void* plan_dev, const float* scale, const float* shift, void* y_split,
                                           int B, int D, int H, int W, float neg_slope, mvsgi_stream_t stream) {
    return poly_launch(x_split, plan_dev, scale, shift, y_split, 1, B, D, H, W, neg_slope, 0, mvsgi::as_stream(stream));
}

// either output (y_is_split: 0 = fp32 [B][2D][2H][2W][16], 1 = split-padded, 3 = split-padded with the DIRECT main kernel whatever
// the dispatcher would choose, 5 = split-padded with the WINOGRAD-form main kernel or an error) in either split: input, plan and a
// split output all in `fmt`
extern "C" int mvsgi_conv3d_up2_poly_fmt(const void* x_split, const void* plan_dev, const float* scale, const float* shift, void* y,
                                         int y_is_split, int B, int D, int H, int W, float neg_slope, int fmt, mvsgi_stream_t stream) {
    MVSGI_REQUIRE(y_is_split == 0 || y_is_split == 1 || y_is_split == 3 || y_is_split == 5, "mvsgi_conv3d_up2_poly: y_is_split %d not in {0, 1, 3, 5}",
                  y_is_split);
    return poly_launch(x_split, plan_dev, scale, shift, y, y_is_split ? 1 : 0, B, D, H, W, neg_slope, fmt, mvsgi::as_stream(stream),
                       y_is_split == 3 ? 1 : (y_is_split == 5 ? 2 : 0));
}

// the main kernel the dispatcher launches for a split-padded fp16 output of this geometry and batch: 1 = the Winograd form
// (csrc/conv3d_wino_up2.hip), 0 = the direct register-stationary kernel
extern "C" int mvsgi_conv3d_up2_poly_wino_pays(int B, int D, int H, int W) {
    if (!mvsgi::wino_up2_applies(D, H, W) || B < 1) return 0;
    const int cus = mvsgi::device_cus();
    const long long units = (long long)B * (H / 2) * (W / 32);
    long long nwalk = cus / 16;
    if (nwalk > mvsgi::cdiv(units, 8)) nwalk = mvsgi::cdiv(units, 8);
    if (nwalk < 1) nwalk = 1;
    const long long rounds = mvsgi::cdiv(mvsgi::cdiv(units, 8), nwalk);
    return 10 * units >= 7 * rounds * 8 * nwalk ? 1 : 0;
}
