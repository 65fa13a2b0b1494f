// K2w: Winograd F(2x2, 3x3) over (H, W), direct over D, for the 32 -> 32 residual convolutions of UNet level 0
// (BaseConvBlk3d.forward, dsta_mvs/model/common/common_modules.py:107-115, as used by ResConvBlk3d :231-244) in the fp16 split.
//
// Why.  conv3d_rs32_kernel sits at the matrix rate the board's power cap allows: the only lever left on these
// layers is fewer MFMAs.  With 2x2 output tiles in the plane, y = A^T [ sum_kd U_kd (.) V_(d + kd - 1) ] A,  U = G g G^T (4 x 4 per
// (cout, cin, kd)),  V = B^T x B (4 x 4 per (tile, cin, plane)), and the 16 element-wise products are 16 independent
// [32 couts x 32 cins] x [32 cins x tiles] GEMMs: 16 x 3 "taps" per 4 output voxels instead of 27 per voxel -- 2.25 x fewer MFMAs.
// In the fp16 split (x = hi + lo, 11 + 11 bits; the TRANSFORMED operands are split) the result is still ~25 x closer to the
// reference than the direct convolution in the bf16 split (tools/winograd_split_emulation.py).
//
// Shape of the kernel.  A workgroup is 4 waves, one per SIMD; wave `a` owns row a of the 4 x 4 transform space: its 4 (b) x 3 (kd) x
// 2 (cout tiles) x (hi, lo) weight fragments = 192 registers stay in the accumulator half of the register file for the whole launch.
// A unit is 16 tiles in a row (2 output rows x 32 columns) marched through all D (8 or 16) planes: per plane a lane (tile n, channel
// group kg) reads the two patch rows its wave's `a` needs (B^T has two non-zeros per row) from the plane's LDS image (every global read
// is an LDS-DMA request, three planes ahead), transforms 8 channels in registers straight into the MFMA's B-operand layout (no LDS
// round trip for V) piece by piece behind the MFMAs, and issues 72 MFMAs: plane p adds U_kd V_p to the open output planes
// p + 1 - kd.  A finished output plane leaves through A^T: the (b) half inside the wave, the (a) half across the waves through a
// 16 KiB LDS exchange, after which wave (pa, q) owns output voxel (2 r + pa, 2 n + q) of every tile: scale / shift, residual,
// LeakyReLU, (split,) store.  Tensors are padded (conv3d_rs.hip's geometry: the zero border is the convolution's padding) and hold
// either fp16 pairs (split-padded) or plain fp32 records ("fp32-padded": what the layers of a chain hand to each other).
// DESIGN.md section 2, K2w, has the measurements and what it took to make hipcc emit it.
#include "common.hpp"

#include <cstring>
#ifdef MVSGI_WINO_STAMPS
#include <cstdio>
#include <cstdlib>
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#include "split_fmt.hpp"

struct WinoArgs {
    const unsigned char* x;    // padded [B][D+2][H+2][W+2][32 x 4 bytes]: fp16 pairs (split-padded) or plain fp32 (A32)
    unsigned char* y;          // the same geometry and format, or a plain fp32 [B][D][H][W][32] tensor (OUT32)
    const unsigned char* res;  // residual in x's format or nullptr
    const u32x4* wp;           // [a 4][b 4][kd 3][cout tile 2][hi | lo][64 lanes] 16-byte fragments
    const float* scale;        // (carries the inverse of the weights' pre-scaling)
    const float* shift;
    int B, D, H, W;
    float neg_slope;
    int tiles_h, groups_w, total_units;
    unsigned long long* dbg;   // -DMVSGI_WINO_STAMPS builds (ABL = 16): per wave of workgroup 0, cycles per step section
    unsigned* sat;             // the range report's words (csrc/api.cpp)
};

#include "conv3d_wino_common.hpp"

#ifndef WN_VCLAMP_A32
#define WN_VCLAMP_A32 0     // (A/B builds: 1 = clamp the transform's sums on fp32-padded input too)
#endif
constexpr float kWinoActMax = 16376.f;     // fp32-padded activations are written clamped to this: the transform's sums of four stay inside fp16's range

namespace wn {
// LDS image of one input plane of a unit: 4 rows x 34 columns x 8 sixteen-byte pieces per voxel, as TWO sub-images: the pieces read by
// the even channel groups (kg = 0, 2) and those read by the odd ones, SUB slots apart (a multiple of 16); inside a sub-image even and
// odd columns apart, 5 slots per voxel (its 4 pieces + 1 pad).  ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): tiles n = 0-3, 12-15 of one channel group with tiles 4-11 of the next.  The two
// read the same slot of their own sub-image's records, so a group walks 16 same-parity columns with stride 5, odd: all 16
// sixteen-byte units of the 256-byte bank row (a single image with 9 slots per voxel is conflict-free for 16 CONSECUTIVE lanes and
// two-way conflicted under the real groups: 8 LDS cycles per read instead of 4).  The staging side stays pieces of 128-byte records.
constexpr int HALF = 17, PITCH = 5, RPITCH = 9;   // slots per voxel: plane sub-images | residual records (8 pieces + 1 pad)
constexpr int NVOX = 4 * 2 * HALF;                // 136 voxels of a plane region
constexpr int SUB = (NVOX * PITCH + 15) / 16 * 16;    // 688
constexpr int PLANE_SLOTS = 2 * SUB;              // 1376
constexpr int NDMA = 22, DPW = 6;                 // 1 KiB pieces per plane; per wave (waves 2, 3 aim their sixth at a dummy KiB)
constexpr int PLANE_LDS = NDMA * 1024;            // 22,528
constexpr int NBUF = 4;                           // the staging stream runs three planes ahead of the arithmetic

constexpr int ZB = NBUF * PLANE_LDS;              // the exchange: 2 x [a 4][q 2][cout tile 2][64 lanes][16 B]
// residual records of one output plane of a unit: 2 rows x 32 voxels at the same 9-slot pitch (the epilogue's 8-byte reads walk them
// with stride 18 slots: two-way conflicts at worst), 576 slots in 9 one-KiB pieces; three images (a unit's last plane is requested a
// step early)
constexpr int RB = ZB + 2 * 16384, RDPW = 3, RES_LDS = 4 * RDPW * 1024, NRES = 3;
constexpr int DUMMY = RB + NRES * RES_LDS;        // where the staging requests that have no piece land (zeros)
constexpr int LDS_BYTES = DUMMY + 1024;           // 160,768
static_assert(PLANE_SLOTS <= NDMA * 64, "DMA pieces cover the image");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
}  // namespace wn

// OUT32: the output is a plain fp32 channels-last tensor [B][D][H][W][32] (the hand-over to a kernel that stages fp32)
// ABL: 0, or 16 = the section stamps of a -DMVSGI_WINO_STAMPS build
// A32: activations (input, residual, padded output) are "fp32-padded" -- the split-padded geometry with plain fp32 records
//      [B][D+2][H+2][W+2][32 floats], zero border -- instead of fp16 pairs: the hand-over BETWEEN Winograd layers (their operands are
//      split after the transform anyway: no join in front of it, no split behind the epilogue: ~90 of ~420 vector instructions a step)
template <int ABL, bool RES, bool OUT32, int DEPTH, bool A32>
__global__ __launch_bounds__(256, 1) void conv3d_wino32_kernel(WinoArgs a) {
    using namespace wn;
    static_assert(DEPTH % NBUF == 0 && DEPTH % 2 == 0, "image / V pair phases of the unrolled unit");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);                 // this wave's row of the transform space
    const int n = lane & 15, kg = lane >> 4;
    const int Hp = a.H + 2, Wp = a.W + 2;
    const long long plane_bytes = (long long)Hp * Wp * 128;
    const long long frame_bytes = (a.D + 2) * plane_bytes;
    const long long total_bytes = frame_bytes * a.B;
    // B^T row a has two non-zeros: t = x[i0] + sgn * x[i1]
    const int i0 = wv == 0 ? 0 : (wv == 2 ? 2 : 1);
    const int i1 = wv == 0 ? 2 : (wv == 1 ? 2 : (wv == 2 ? 1 : 3));
    const float sgn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(wv == 1 ? 0x3f800000 : 0xbf800000));
    // after the exchange this wave finishes output voxel (pa, q) of every tile
    const int pa = wv & 1, q = wv >> 1;
    const float osg = pa ? -1.f : 1.f;

    // ---- weights: resident for the whole launch ----
    u32x4 wh[4][3][2], wl[4][3][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const u32x4* p_ = a.wp + ((long long)((((wv * 4 + b) * 3 + kd) * 2 + c) * 2)) * 64 + lane;
                wh[b][kd][c] = p_[0];
                wl[b][kd][c] = p_[64];
            }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int c = 0; c < 2; ++c) asm volatile("" : "+a"(wh[b][kd][c]), "+a"(wl[b][kd][c]));
    const f32x4 esc_[2] = {*reinterpret_cast<const f32x4*>(a.scale + kg * 4), *reinterpret_cast<const f32x4*>(a.scale + 16 + kg * 4)};
    const f32x4 esh_[2] = {*reinterpret_cast<const f32x4*>(a.shift + kg * 4), *reinterpret_cast<const f32x4*>(a.shift + 16 + kg * 4)};
    __builtin_amdgcn_sched_barrier(0);

    const int lane_out = (2 * n) * 128 + (kg >> 1) * 16 + (kg & 1) * 8;       // this lane's 8-byte hi piece of an output voxel's slice 0
    const int lane_out32 = (2 * n) * 128 + kg * 16;                           // (fp32 output: its 16 bytes = couts 4 kg .. 4 kg + 3 of the voxel's first 16)
    const long long oplane_bytes = (long long)a.H * a.W * 128;
    // fragment reads: this lane's hi piece (slice kg >> 1, channel half kg & 1) of column 2 n in patch rows i0 / i1, image 0
    // (fp32 records: channels 8 kg .. 8 kg + 7 are pieces 2 kg, 2 kg + 1)
    const int rd0 = ((kg & 1) * SUB + (i0 * 2 * HALF + n) * PITCH + (kg >> 1) * 2) * 16;
    const int rd1 = ((kg & 1) * SUB + (i1 * 2 * HALF + n) * PITCH + (kg >> 1) * 2) * 16;
    constexpr int RD2 = 16;                           // from a lane's first piece to its second (hi -> lo, or channels 0-3 -> 4-7)
    // ---- DMA plans: piece m = wv + 4 k fills slots [64 m, 64 m + 64) of an image ----
    unsigned voff[DPW], rvoff[RDPW];
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
        const int m = wv + 4 * k;
        const int sl = m * 64 + lane;
        const int sub = sl >= SUB ? 1 : 0, rem = sl - sub * SUB;
        const int vox = rem / PITCH, c = rem - vox * PITCH;
        const int row = vox / (2 * HALF), rr = vox - row * (2 * HALF);
        const int par = rr / HALF, idx = rr - par * HALF;
        // piece c of a record: fp32 records -- even groups read pieces 0, 1 | 4, 5, odd ones 2, 3 | 6, 7; fp16 pairs -- [slice][hi | lo]
        // [channels 0-7 | 8-15]: even groups read the "0-7" pieces (0, 2 | 4, 6), odd ones the "8-15" pieces (1, 3 | 5, 7)
        const int piece = A32 ? (c >> 1) * 4 + sub * 2 + (c & 1) : c * 2 + sub;
        voff[k] = (m < NDMA && sl < PLANE_SLOTS && vox < NVOX && c < 4) ? (unsigned)((row * Wp + 2 * idx + par) * 128 + piece * 16)
                                                                        : 0xffffff00u;          // pads: beyond num_records, zero-filled
    }
#pragma unroll
    for (int k = 0; k < RDPW; ++k) {
        const int sl = (wv + 4 * k) * 64 + lane;
        const int vox = sl / RPITCH, piece = sl - vox * RPITCH;
        rvoff[k] = (vox < 64 && piece < 8) ? (unsigned)(((vox >> 5) * Wp + (vox & 31)) * 128 + piece * 16) : 0xffffff00u;
    }
    // the epilogue's residual reads: voxel (row pa, column 2 n + q), this lane's 8 bytes of slice 0's hi piece
    const int rrd = A32 ? ((pa * 32 + 2 * n + q) * RPITCH + kg) * 16                      // (fp32 records: 16 bytes = couts 4 kg .. 4 kg + 3 of a cout tile)
                        : ((pa * 32 + 2 * n + q) * RPITCH + (kg >> 1)) * 16 + (kg & 1) * 8;

    f32x4 Y[3][4][2];
    u32x4 vh[2][4], vl[2][4];      // V of the plane being multiplied | of the next one (written while the first is read)
    u32x4 raw[2][4][2];            // [patch row i0 | i1][column][hi | lo] of the plane being transformed
    float T[2][4];                 // h-pass of one channel pair: [lo | hi half][column]

    const int G = gridDim.x;
    // The walk: workgroups are dealt to the 8 XCDs round-robin; XCD x owns a contiguous eighth of the unit space and its workgroups
    // walk it side by side (ids ubase + k * ustep), so that the units sharing patch rows -- tile rows r and r + 1, groups_w ids apart --
    // are read through ONE L2 at about the same time (the plain walk blockIdx + k * G fetched every shared row from HBM twice:
    // 1313 MB per launch against 840 MB of tensors)
    const int T_ = a.total_units;
    int ubase = (int)blockIdx.x, ustep = G, nmine = (T_ - (int)blockIdx.x + G - 1) / G;
    if ((G & 7) == 0 && T_ >= G) {
        const int x = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3, q8 = T_ >> 3, r8 = T_ & 7;
        const int lo = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8, cnt = q8 + (x < r8 ? 1 : 0);
        ustep = G >> 3;
        ubase = lo + i;
        nmine = i < cnt ? (lo + cnt - ubase + ustep - 1) / ustep : 0;
    }
    if (nmine == 0) return;         // (the whole workgroup, before anything was requested)

// descriptor whose base is the patch origin of unit U (padded rows 2 r .., columns 32 c ..); no records past the workgroup's last unit:
// requests through it write zeros
#define WN_DESC(U, LIVE)                                                                                    \
    ({                                                                                                      \
        const int c_ = (U) % a.groups_w, t_ = (U) / a.groups_w;                                             \
        const int r_ = t_ % a.tiles_h, b_ = t_ / a.tiles_h;                                                 \
        const long long off_ = (LIVE) ? b_ * frame_bytes + ((long long)(2 * r_) * Wp + 32 * c_) * 128 : 0;  \
        const long long left_ = total_bytes - off_;                                                         \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.x) + off_, 0,                        \
                                          (LIVE) ? (left_ > 0x7fffff00ll ? 0x7fffff00 : (int)left_) : 0, 0x00020000); \
    })
// Staging: input plane (P + 3) of the plane stream -- of this unit, or of the next one from P = D - 3 on -- goes to image (P + 3) % 4
// (D % 4 == 0: the images keep their phase over units); the residual records of plane P to image `ri` of a ring of three.
#define WN_DMA_ISSUE(P3, K0, K1)                                                                            \
    _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_)                                                  \
        __builtin_amdgcn_raw_ptr_buffer_load_lds((P3) < DEPTH ? dsc : dsc_next,                             \
                                                 (__attribute__((address_space(3))) void*)(lds + (wv + 4 * k_ < NDMA ? ((P3) % NBUF) * PLANE_LDS + (wv + 4 * k_) * 1024 : DUMMY)), \
                                                 16, voff[k_], (unsigned)(((P3) % DEPTH + 1) * plane_bytes), 0, 0);
#define WN_RES_DMA(IMG, OPLANE)                                                                             \
    {                                                                                                       \
        const unsigned so_ = (unsigned)(((OPLANE) + 1) * plane_bytes);                                      \
        _Pragma("unroll") for (int k_ = 0; k_ < RDPW; ++k_)                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdsc, (__attribute__((address_space(3))) void*)(lds + RB + (IMG) * RES_LDS + (wv + 4 * k_) * 1024), \
                                                     16, rvoff[k_], so_, 0, 0);                             \
    }
#define WN_READ(BUF_)                                                                                       \
    {                                                                                                       \
        const unsigned char* im_ = lds + (BUF_) * PLANE_LDS;                                                \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                  \
            const int o_ = (((j_ & 1) * HALF + (j_ >> 1)) * PITCH) * 16;                                    \
            raw[0][j_][0] = *reinterpret_cast<const u32x4*>(im_ + rd0 + o_);                                \
            raw[0][j_][1] = *reinterpret_cast<const u32x4*>(im_ + rd0 + o_ + RD2);                          \
            raw[1][j_][0] = *reinterpret_cast<const u32x4*>(im_ + rd1 + o_);                                \
            raw[1][j_][1] = *reinterpret_cast<const u32x4*>(im_ + rd1 + o_ + RD2);                          \
        }                                                                                                   \
    }
// One piece of the transform raw -> V[VN] (all four b of this wave's a; split; B-operand layout: lane (tile n, kg) holds channels
// 8 kg .. 8 kg + 7), sized to ride behind one MFMA: M = 0 .. 71 -> channel pair d = M / 18 and, within it, 8 pieces of the h-pass (one
// (column, half) each: 3 dependent mixed-precision fmas), then per b one piece of the w-pass + range clamp (4 instructions) and one of
// the split (4); piece 16 closes the last b, piece 17 carries the step's staging requests.
#define WN_TPIECE(M, VN)                                                                                    \
    {                                                                                                       \
        constexpr int d_ = (M) / 18, r_ = (M) % 18;                                                         \
        if constexpr (r_ < 8) {                                                                             \
            constexpr int j_ = r_ >> 1;                                                                     \
            /* ONE asm statement per chain: hipcc pads every VGPR an asm statement defines against a use by the very next instruction \
               (it must assume a partial-register write), 4 cycles per pad, and only its own instructions count as distance */ \
            if constexpr (A32) {    /* channel 2 d_ + (r_ & 1) of the lane's eight: one fma */            \
                constexpr int e_ = (d_ & 1) * 2 + (r_ & 1);                                                 \
                T[r_ & 1][j_] = __builtin_fmaf(__builtin_bit_cast(float, raw[1][j_][d_ >> 1][e_] | 0u), sgn, \
                                               __builtin_bit_cast(float, raw[0][j_][d_ >> 1][e_] | 0u));    \
            } else if constexpr ((r_ & 1) == 0) {                                                           \
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"                    \
                    "v_fma_mix_f32 %0, %3, %5, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                     \
                    "v_fma_mix_f32 %0, %4, %5, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]"                          \
                    : "=&v"(T[0][j_]) : "v"(raw[0][j_][0][d_]), "v"(raw[0][j_][1][d_]), "v"(raw[1][j_][0][d_]), "v"(raw[1][j_][1][d_]), "s"(sgn)); \
            } else {                                                                                        \
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n\t"                    \
                    "v_fma_mix_f32 %0, %3, %5, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                     \
                    "v_fma_mix_f32 %0, %4, %5, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]"                          \
                    : "=&v"(T[1][j_]) : "v"(raw[0][j_][0][d_]), "v"(raw[0][j_][1][d_]), "v"(raw[1][j_][0][d_]), "v"(raw[1][j_][1][d_]), "s"(sgn)); \
            }                                                                                               \
        } else if constexpr (r_ < 16) {                                                                     \
            constexpr int b_ = (r_ - 8) >> 1;                                                               \
            constexpr int x_ = b_ == 0 ? 0 : (b_ == 2 ? 2 : 1), y_ = b_ == 0 ? 2 : (b_ == 1 ? 2 : (b_ == 2 ? 1 : 3)); \
            if constexpr (((r_ - 8) & 1) == 0) {                                                            \
                X0 = b_ == 1 ? T[0][x_] + T[0][y_] : T[0][x_] - T[0][y_];                                   \
                X1 = b_ == 1 ? T[1][x_] + T[1][y_] : T[1][x_] - T[1][y_];                                   \
                if constexpr (!A32 || WN_VCLAMP_A32) {      /* (fp32-padded records arrive clamped to +-16376: their sums of four cannot leave fp16's range) */ \
                    X0 = sf_clamp<true>(X0);                                                                \
                    X1 = sf_clamp<true>(X1);                                                                \
                    satm = sf_sat_acc(satm, X0, X1);        /* range report: a sum of four left fp16's range */ \
                }                                                                                           \
                if constexpr (b_ > 0) vl[VN][b_ - 1][d_] = sf_cvt_pk<true>(L0, L1);     /* (the lo halves of the b before: not right behind the asm that made them) */ \
            } else {                                                                                        \
                const unsigned h_ = sf_cvt_pk<true>(X0, X1);                                                \
                vh[VN][b_][d_] = h_;                                                                        \
                asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                   \
                    "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"                        \
                    : "=&v"(L0), "=&v"(L1) : "v"(h_), "v"(X0), "v"(X1));                                    \
            }                                                                                               \
        } else if constexpr (r_ == 16) {                                                                    \
            vl[VN][3][d_] = sf_cvt_pk<true>(L0, L1);                                                        \
        }                                                                                                   \
    }
#define WN_DPIECE(M, P)                                                                                     \
    if constexpr ((M) == 17) { if constexpr (RES) { WN_RES_DMA(ri, P) } }                                   \
    else if constexpr ((M) == 35) { WN_DMA_ISSUE((P) + 3, 0, 2) }                                           \
    else if constexpr ((M) == 53) { WN_DMA_ISSUE((P) + 3, 2, 4) }                                           \
    else if constexpr ((M) == 71) { WN_DMA_ISSUE((P) + 3, 4, 6) }
#define WN_STAMP(K)                                                                                         \
    if constexpr (ABL & 16) {                                                                               \
        unsigned long long t_;                                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
        tsum[K] += t_ - tlast;                                                                              \
        tlast = t_;                                                                                         \
    }
// the finished accumulators of slot SF leave through A^T: over b inside the wave, over a across the waves (LDS); then wave (pa, q)
// finishes its output voxel of every tile
#define WN_OUT_WRITE(SF, ZIMG)                                                                              \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) {                                                      \
        f32x4 y0_ = Y[SF][0][c_], y1_ = Y[SF][1][c_], y2_ = Y[SF][2][c_], y3_ = Y[SF][3][c_];               \
        if constexpr ((SF) >= 1) {      /* out of the accumulator file WHOLE (element-wise reads of an asm "+a" vector gave stale values, hipcc 7.2) */ \
            asm volatile("" : "+v"(y0_), "+v"(y1_), "+v"(y2_), "+v"(y3_));                                  \
        }                                                                                                   \
        const f32x4 z0_ = (y0_ + y1_) + y2_;                                                                \
        const f32x4 z1_ = (y1_ - y2_) - y3_;                                                                \
        *reinterpret_cast<f32x4*>(lds + ZB + (ZIMG) * 16384 + ((wv * 2 + 0) * 2 + c_) * 1024 + lane * 16) = z0_; \
        *reinterpret_cast<f32x4*>(lds + ZB + (ZIMG) * 16384 + ((wv * 2 + 1) * 2 + c_) * 1024 + lane * 16) = z1_; \
    }
#define WN_EPI_READS(ZIMG)                                                                                  \
    f32x4 zz_[2][3];                                                                                        \
    u32x2 rh_[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}}, rl_[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};                 \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_)                                                        \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_)                                                    \
            zz_[c_][k_] = *reinterpret_cast<const f32x4*>(lds + ZB + (ZIMG) * 16384 + (((pa + k_) * 2 + q) * 2 + c_) * 1024 + lane * 16); \
    f32x4 rf_[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};                                 \
    if constexpr (RES) {                                                                                    \
        _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) {                                                  \
            if constexpr (A32) {                                                                            \
                rf_[c_] = *reinterpret_cast<const f32x4*>(rim_ + rrd + c_ * 64);                            \
            } else {                                                                                        \
                rh_[c_] = *reinterpret_cast<const u32x2*>(rim_ + rrd + c_ * 64);                            \
                rl_[c_] = *reinterpret_cast<const u32x2*>(rim_ + rrd + c_ * 64 + 32);                       \
            }                                                                                               \
        }                                                                                                   \
    }
#define WN_EPILOGUE(O)                                                                                      \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) {                                                      \
        f32x4 t_ = zz_[c_][0] + osg * (zz_[c_][1] + zz_[c_][2]);                                            \
        t_ = t_ * esc_[c_] + esh_[c_];                                                                      \
        if constexpr (RES && A32) {                                                                         \
            t_ = t_ + rf_[c_];                                                                              \
        } else if constexpr (RES) {                                                                         \
            t_[0] = mix_add_lo(rl_[c_][0], mix_add_lo(rh_[c_][0], t_[0]));                                  \
            t_[1] = mix_add_hi(rl_[c_][0], mix_add_hi(rh_[c_][0], t_[1]));                                  \
            t_[2] = mix_add_lo(rl_[c_][1], mix_add_lo(rh_[c_][1], t_[2]));                                  \
            t_[3] = mix_add_hi(rl_[c_][1], mix_add_hi(rh_[c_][1], t_[3]));                                  \
        }                                                                                                   \
        if constexpr (A32 && !OUT32) {      /* LeakyReLU, THEN the fp32-padded format's range (clamping inside the activation's med3 let \
                                               t * neg_slope > 16376 through: any t > 16376 at slope 1, t > 1.6e6 at 0.01) + the range report */ \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                                \
                t_[e_] = __builtin_amdgcn_fmed3f(__builtin_fmaxf(t_[e_], t_[e_] * a.neg_slope), -kWinoActMax, kWinoActMax); \
            satm = sf_sat_acc(sf_sat_acc(satm, t_[0], t_[1]), t_[2], t_[3]);                                \
        } else {                                                                                            \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) t_[e_] = __builtin_fmaxf(t_[e_], t_[e_] * a.neg_slope); \
        }                                                                                                   \
        if constexpr (OUT32) {                                                                              \
            *reinterpret_cast<f32x4*>(yb + (long long)(O) * oplane_bytes + lane_out32 + c_ * 64) = t_;      \
        } else if constexpr (A32) {                                                                         \
            *reinterpret_cast<f32x4*>(yb + (long long)((O) + 1) * plane_bytes + lane_out32 + c_ * 64) = t_; \
        } else {                                                                                            \
            unsigned h0_, l0_, h1_, l1_;                                                                    \
            split_pair(t_[0], t_[1], h0_, l0_, satm);                                                       \
            split_pair(t_[2], t_[3], h1_, l1_, satm);                                                       \
            unsigned char* q_ = yb + (long long)((O) + 1) * plane_bytes + lane_out + c_ * 64;               \
            *reinterpret_cast<u32x2*>(q_) = u32x2{h0_, h1_};                                                \
            *reinterpret_cast<u32x2*>(q_ + 32) = u32x2{l0_, l1_};                                           \
        }                                                                                                   \
    }
// one MFMA and the pieces that ride behind it (term-major inside a depth tap: the three products of an accumulator are 8 MFMAs apart;
// the accumulators finished in this step first).  Depth taps that fall on the zero border have no MFMAs: kd = 2 of plane 0 (output
// plane -1), kd = 0 of plane D - 1 (output plane D); an output plane's first contribution starts its accumulators (C = 0).
#define WN_SLOT(M, P, SI, SM, SF, VB)                                                                       \
    {                                                                                                       \
        constexpr int g_ = (M) / 24, w_ = (M) % 24, tm_ = w_ / 8, b_ = (w_ % 8) / 2, c_ = w_ % 2;           \
        constexpr int kd_ = 2 - g_, s_ = g_ == 0 ? (SF) : (g_ == 1 ? (SM) : (SI));                          \
        constexpr bool live_ = !(g_ == 0 && (P) == 0) && !(g_ == 2 && (P) == DEPTH - 1);                    \
        constexpr bool first_ = tm_ == 0 && (g_ == 2 || (g_ == 1 && (P) == 0));                             \
        if constexpr (live_) {                                                                              \
            if constexpr (s_ >= 1) {                                                                        \
                if constexpr (first_) { WN_MFA0(Y[s_][b_][c_], wl[b_][kd_][c_], vh[VB][b_]) }               \
                else if constexpr (tm_ == 0) { WN_MFA(Y[s_][b_][c_], wl[b_][kd_][c_], vh[VB][b_]) }         \
                else if constexpr (tm_ == 1) { WN_MFA(Y[s_][b_][c_], wh[b_][kd_][c_], vl[VB][b_]) }         \
                else { WN_MFA(Y[s_][b_][c_], wh[b_][kd_][c_], vh[VB][b_]) }                                 \
            } else {                                                                                        \
                if constexpr (first_) { WN_MF0(Y[s_][b_][c_], wl[b_][kd_][c_], vh[VB][b_]) }                \
                else if constexpr (tm_ == 0) { WN_MF(Y[s_][b_][c_], wl[b_][kd_][c_], vh[VB][b_]) }          \
                else if constexpr (tm_ == 1) { WN_MF(Y[s_][b_][c_], wh[b_][kd_][c_], vl[VB][b_]) }          \
                else { WN_MF(Y[s_][b_][c_], wh[b_][kd_][c_], vh[VB][b_]) }                                  \
            }                                                                                               \
        }                                                                                                   \
        WN_TPIECE(M, (VB) ^ 1)                                                                              \
        WN_DPIECE(M, P)                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
#define WN_S8(M, P, SI, SM, SF, VB)                                                                         \
    WN_SLOT(M, P, SI, SM, SF, VB) WN_SLOT(M + 1, P, SI, SM, SF, VB) WN_SLOT(M + 2, P, SI, SM, SF, VB) WN_SLOT(M + 3, P, SI, SM, SF, VB) \
    WN_SLOT(M + 4, P, SI, SM, SF, VB) WN_SLOT(M + 5, P, SI, SM, SF, VB) WN_SLOT(M + 6, P, SI, SM, SF, VB) WN_SLOT(M + 7, P, SI, SM, SF, VB)

// One plane step, P = 0 .. D - 1 (a compile-time constant: a unit is ONE basic block -- hipcc re-homes asm-tied accumulators at block
// boundaries, behind the back of the hazards it cannot see).  V of plane P in vh / vl[VB]; raw holds the NEXT plane of the stream (the
// next unit's first after P = D - 1), whose V goes to vh / vl[VB ^ 1] piece by piece behind this plane's MFMAs.  Open output planes:
// P + 1 in slot SI, P in slot SM, P - 1 in slot SF (last contribution: leaves through the exchange at the end of the step).
#define WN_STEP(P, SI, SM, SF, VB)                                                                          \
    {                                                                                                       \
        WN_STAMP(0)                                                                                         \
        /* every global read of the kernel is an LDS-DMA request waited for by hand (hipcc's vmcnt bookkeeping does not count them: \
           its own waits for ordinary loads would drain the stream).  Per step, in this order: the residual records of plane P (for the \
           epilogue ONE step later -- the next step's, or the unit's closing one), then the input plane three steps ahead */ \
        unsigned char* rim_ = lds + RB + rprev * RES_LDS;                                                   \
        asm volatile("s_nop 1");                                                                            \
        {                                                                                                   \
            float X0 = 0.f, X1 = 0.f, L0 = 0.f, L1 = 0.f;                                                   \
            WN_S8(0, P, SI, SM, SF, VB) WN_S8(8, P, SI, SM, SF, VB) WN_S8(16, P, SI, SM, SF, VB)            \
            WN_S8(24, P, SI, SM, SF, VB) WN_S8(32, P, SI, SM, SF, VB) WN_S8(40, P, SI, SM, SF, VB)          \
            WN_S8(48, P, SI, SM, SF, VB) WN_S8(56, P, SI, SM, SF, VB) WN_S8(64, P, SI, SM, SF, VB)          \
        }                                                                                                   \
        WN_PAD()                                                                                            \
        WN_STAMP(2)                                                                                         \
        if constexpr ((P) > 0) { WN_OUT_WRITE(SF, ((P) - 1) & 1) }                                          \
        /* everything older than this step's requests has landed: the residual asked for a step ago, the image of the plane after the next */ \
        if constexpr (RES) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");                      \
        else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                                    \
        WN_STAMP(3)                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                       \
        WN_STAMP(4)                                                                                         \
        if constexpr ((P) > 0) {                                                                            \
            WN_EPI_READS(((P) - 1) & 1)                                                                     \
            WN_EPILOGUE((P) - 1)                                                                            \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        WN_READ(((P) + 2) % NBUF)                       /* the plane after the next: its latency under the next step's start */ \
        WN_STAMP(5)                                                                                         \
        rprev = ri;                                                                                         \
        ri = ri == NRES - 1 ? 0 : ri + 1;                                                                   \
    }
// the unit's last output plane (its slot got the last contribution in step D - 1: plane D is the zero border)
#define WN_FINISH(SF)                                                                                       \
    {                                                                                                       \
        unsigned char* rim_ = lds + RB + rprev * RES_LDS;                                                   \
        WN_OUT_WRITE(SF, (DEPTH - 1) & 1)                                                                   \
        /* its residual was requested in the step before, in front of that step's input plane (6 requests) and output stores (4, or 2 fp32) */ \
        if constexpr (OUT32 || A32) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");             \
        else asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");                                   \
        __builtin_amdgcn_s_barrier();                                                                       \
        WN_EPI_READS((DEPTH - 1) & 1)                                                                       \
        WN_EPILOGUE(DEPTH - 1)                                                                              \
    }
#define WN_T8(M) WN_TPIECE(M, 0) WN_TPIECE(M + 1, 0) WN_TPIECE(M + 2, 0) WN_TPIECE(M + 3, 0) WN_TPIECE(M + 4, 0) WN_TPIECE(M + 5, 0) WN_TPIECE(M + 6, 0) WN_TPIECE(M + 7, 0)

    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    if constexpr (ABL & 16) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
    int ri = 0, rprev = 0;                             // residual image to request into / requested a step ago
    float satm = 0.f;                                  // running maximum |clamped value| (range report, csrc/split_fmt.hpp)
    auto dsc = WN_DESC(ubase, true);
    auto dsc_next = dsc;
    // the first three planes of the stream; V of the first; raw of the second
    WN_DMA_ISSUE(0, 0, DPW)
    WN_DMA_ISSUE(1, 0, DPW)
    WN_DMA_ISSUE(2, 0, DPW)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    WN_READ(0)
    {
        float X0 = 0.f, X1 = 0.f, L0 = 0.f, L1 = 0.f;
        WN_T8(0) WN_T8(8) WN_T8(16) WN_T8(24) WN_T8(32) WN_T8(40) WN_T8(48) WN_T8(56) WN_T8(64)
    }
    WN_READ(1)
    for (int k = 0; k < nmine; ++k) {
        const int u = ubase + k * ustep;
        const int c = u % a.groups_w;
        const int t = u / a.groups_w;
        const int r = t % a.tiles_h;
        const int b = t / a.tiles_h;
        // this wave's output voxel origin (padded row 2 r + pa + 1, column 32 c + q + 1)
        unsigned char* yb = OUT32 ? a.y + b * (a.D * oplane_bytes) + ((long long)(2 * r + pa) * a.W + 32 * c + q) * 128
                                  : a.y + b * frame_bytes + ((long long)(2 * r + pa + 1) * Wp + 32 * c + q + 1) * 128;
        const long long roff = b * frame_bytes + ((long long)(2 * r + 1) * Wp + 32 * c + 1) * 128;      // the unit's output rows in the residual tensor
        const long long rleft = total_bytes - roff;
        const auto rdsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(RES ? a.res : a.x) + roff, 0,
                                                            rleft > 0x7fffff00ll ? 0x7fffff00 : (int)rleft, 0x00020000);
        dsc_next = WN_DESC(u + ustep, k + 1 < nmine);
        // slots follow the plane mod 3, the V pair the plane mod 2
#define WN_STEPP(P) WN_STEP(P, ((P) + 1) % 3, (P) % 3, ((P) + 2) % 3, (P) & 1)
        WN_STEPP(0) WN_STEPP(1) WN_STEPP(2) WN_STEPP(3) WN_STEPP(4) WN_STEPP(5) WN_STEPP(6) WN_STEPP(7)
        if constexpr (DEPTH == 16) {
            WN_STEPP(8) WN_STEPP(9) WN_STEPP(10) WN_STEPP(11) WN_STEPP(12) WN_STEPP(13) WN_STEPP(14) WN_STEPP(15)
        }
        WN_FINISH((DEPTH - 1) % 3)
        dsc = dsc_next;
    }
    // fp32-padded records are clamped to +-16376 where they are written; on fp16 pairs the transform's sums and the output to +-65504
    if constexpr (!(A32 && OUT32)) sf_sat_report(a.sat, kSatWino, satm, A32 ? kWinoActMax : kF16Max);
    if constexpr (ABL & 16) {
        if (blockIdx.x == 0 && lane == 0 && a.dbg)
            for (int k = 0; k < 8; ++k) a.dbg[wv * 8 + k] = tsum[k];
    }
}

template <int ABL, bool RES, bool OUT32, int DEPTH, bool A32>
int wino_launch5(const WinoArgs& a, long long units, hipStream_t st) {
    static mvsgi::PersistentGeom geo_cache[mvsgi::kMaxDevices] = {};
    mvsgi::PersistentGeom geo;
    if (mvsgi::persistent_geometry(conv3d_wino32_kernel<ABL, RES, OUT32, DEPTH, A32>, 256, wn::LDS_BYTES, 1, geo_cache, "conv3d(winograd)", geo)) return 1;
    const unsigned grid = (unsigned)(units < geo.cus ? units : geo.cus);
    hipLaunchKernelGGL((conv3d_wino32_kernel<ABL, RES, OUT32, DEPTH, A32>), dim3(grid), dim3(256), wn::LDS_BYTES, st, a);
    return 0;
}
template <int ABL, int DEPTH, bool A32>
int wino_launch3(const WinoArgs& a, long long units, bool out32, hipStream_t st) {
    if (out32) return a.res ? wino_launch5<ABL, true, true, DEPTH, A32>(a, units, st) : wino_launch5<ABL, false, true, DEPTH, A32>(a, units, st);
    return a.res ? wino_launch5<ABL, true, false, DEPTH, A32>(a, units, st) : wino_launch5<ABL, false, false, DEPTH, A32>(a, units, st);
}
template <int ABL>
int wino_launch(const WinoArgs& a, long long units, bool out32, bool act32, hipStream_t st) {
    if (act32) return a.D == 16 ? wino_launch3<ABL, 16, true>(a, units, out32, st) : wino_launch3<ABL, 8, true>(a, units, out32, st);
    return a.D == 16 ? wino_launch3<ABL, 16, false>(a, units, out32, st) : wino_launch3<ABL, 8, false>(a, units, out32, st);
}

// [32][32][27] -> U = G g G^T per (cout, cin, kd), pre-scaled per cout by a power of two so that max |U| lies in (512, 1024], split, in
// the MFMA's A-operand order [a][b][kd][cout tile][hi | lo][lane = (cin group) * 16 + cout][8 cins]; unscale[cout] = 2^-k.
// One workgroup per cout: thread (ci = t & 31, group = t >> 5) computes 6 of the 48 (a, b, kd).
__global__ __launch_bounds__(256) void wino_pack_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, float* __restrict__ unscale) {
    __shared__ float red[256];
    const int co = blockIdx.x, t = threadIdx.x, ci = t & 31, grp = t >> 5;
    constexpr float Gm[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const float* g = w + ((long long)co * 32 + ci) * 27;
    float u[6], amax = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int e = grp * 6 + i, kd = e % 3, ab = e / 3, a_ = ab >> 2, b_ = ab & 3;
        float acc = 0.f;
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) acc += Gm[a_][kh] * Gm[b_][kw] * g[kd * 9 + kh * 3 + kw];
        u[i] = acc;
        amax = fmaxf(amax, fabsf(acc));
    }
    red[t] = amax;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if (t < sft) red[t] = fmaxf(red[t], red[t + sft]);
        __syncthreads();
    }
    amax = red[0];
    int k = 0;
    if (amax > 0.f) {
        int e;
        const float m = frexpf(amax, &e);          // amax = m * 2^e, m in [0.5, 1)
        k = m == 0.5f ? 11 - e : 10 - e;            // amax * 2^k in (512, 1024]
        k = k < -100 ? -100 : (k > 100 ? 100 : k);
    }
    const float up = ldexpf(1.f, k);
    if (t == 0) unscale[co] = ldexpf(1.f, -k);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int e = grp * 6 + i, kd = e % 3, ab = e / 3;
        unsigned short hi, lo;
        sf_split_weight(u[i] * up, true, hi, lo);
        const long long frag = (((long long)ab * 3 + kd) * 2 + (co >> 4)) * 2;
        const int lane = (ci >> 3) * 16 + (co & 15);
        wp[((frag + 0) * 64 + lane) * 8 + (ci & 7)] = hi;
        wp[((frag + 1) * 64 + lane) * 8 + (ci & 7)] = lo;
    }
}

}  // namespace

extern "C" {

size_t mvsgi_conv3d_wino32_packed_weight_bytes(void) { return (size_t)4 * 4 * 3 * 2 * 2 * 64 * 16; }

int mvsgi_conv3d_wino32_applies(int Cin, int Cout, int D, int H, int W, int stride, float neg_slope) {
    return Cin == 32 && Cout == 32 && stride == 1 && (D == 8 || D == 16) && H > 0 && W > 0 && H % 2 == 0 && W % 32 == 0 && neg_slope >= 0.f && neg_slope <= 1.f &&
           (long long)(D + 2) * (H + 2) * (W + 2) * 128 < (1ll << 31);
}

int mvsgi_conv3d_wino32_pack_weights(const float* w_oidhw, void* w_packed, float* unscale, void* stream) {
    MVSGI_REQUIRE(w_oidhw && w_packed && unscale, "mvsgi_conv3d_wino32_pack_weights: null pointer");
    hipLaunchKernelGGL(wino_pack_weights_kernel, dim3(32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w_oidhw,
                       reinterpret_cast<unsigned short*>(w_packed), unscale);
    return mvsgi::check_launch("mvsgi_conv3d_wino32_pack_weights");
}

int mvsgi_conv3d_wino32_f16(const void* x_split, const void* w_packed, const float* scale, const float* shift, const void* res_split,
                            void* y, int y_is_f32, int act_f32p, int B, int D, int H, int W, float neg_slope, void* stream) {
    MVSGI_REQUIRE(x_split && w_packed && scale && shift && y, "mvsgi_conv3d_wino32_f16: null pointer");
    MVSGI_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "mvsgi_conv3d_wino32_f16: bad dims");
    MVSGI_REQUIRE((D == 8 || D == 16) && H % 2 == 0 && W % 32 == 0,
                  "mvsgi_conv3d_wino32_f16: needs D == 8 or 16, H %% 2 == 0 and W %% 32 == 0 (got %d, %d, %d)", D, H, W);
    MVSGI_REQUIRE(neg_slope >= 0.f && neg_slope <= 1.f, "mvsgi_conv3d_wino32_f16: neg_slope outside [0, 1]");
    MVSGI_REQUIRE((long long)B * (D + 2) * (H + 2) * (W + 2) * 128 < (1ll << 40), "mvsgi_conv3d_wino32_f16: tensor too large");
    MVSGI_REQUIRE((long long)(D + 2) * (H + 2) * (W + 2) * 128 < (1ll << 31), "mvsgi_conv3d_wino32_f16: frame too large for 32-bit plane offsets");
    WinoArgs a;
    memset(&a, 0, sizeof(a));
    a.x = reinterpret_cast<const unsigned char*>(x_split);
    a.y = reinterpret_cast<unsigned char*>(y);
    a.res = reinterpret_cast<const unsigned char*>(res_split);
    a.wp = reinterpret_cast<const u32x4*>(w_packed);
    a.scale = scale;
    a.shift = shift;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.neg_slope = neg_slope;
    a.tiles_h = H / 2;
    a.groups_w = W / 32;
    const long long units = (long long)B * a.tiles_h * a.groups_w;
    MVSGI_REQUIRE(units < (1ll << 31), "mvsgi_conv3d_wino32_f16: too many units");
    a.total_units = (int)units;
    MVSGI_SAT_WORDS(sat_words_);
    a.sat = sat_words_;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#ifdef MVSGI_WINO_STAMPS      // diagnostic build: cycles per step section of workgroup 0 (tools/wino_probe.py)
    if (getenv("MVSGI_WINO_STAMP")) {
        static unsigned long long* dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, 32 * 8);
        a.dbg = dbg;
        if (wino_launch<16>(a, units, y_is_f32 != 0, act_f32p != 0, st)) return 1;
        (void)hipDeviceSynchronize();
        unsigned long long h[32];
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 4; ++w) {
            fprintf(stderr, "wave %d:", w);
            for (int k = 0; k < 7; ++k) fprintf(stderr, " %llu", h[w * 8 + k]);
            fprintf(stderr, "\n");
        }
        return mvsgi::check_launch("mvsgi_conv3d_wino32_f16");
    }
#endif
    if (wino_launch<0>(a, units, y_is_f32 != 0, act_f32p != 0, st)) return 1;
    return mvsgi::check_launch("mvsgi_conv3d_wino32_f16");
}

}  // extern "C"
