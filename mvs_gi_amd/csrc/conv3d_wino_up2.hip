// K3w: ResizeConv3d 32 -> 16 (out_costs.0 of the (16, 32) regulator: dsta_mvs/model/common/common_modules.py:332-355,
// cost_volume_regulator/unet_regulator.py:52-60) in the fp16 split as POLYPHASE over (H, W) x WINOGRAD F(2x2, 3x3) over the
// low-resolution (H, W) grid x an EXPLICIT trilinear upsample along D.
//
//   conv3d(interpolate(x, x2, trilinear), w, padding = 1)
// Along H and W the layer is, per output phase (ph, pw), a 3x3 convolution over the low-resolution plane with folded weights
// (M_h x M_w) w (csrc/conv3d_up2poly.hip, dropin/polyphase.py); along D it stays what it is: the three depth taps w[kd] applied to
// the UPSAMPLED planes up[o + kd - 1], up[2 j + 1] = 0.75 x[j] + 0.25 x[j + 1], up[2 j + 2] = 0.25 x[j] + 0.75 x[j + 1], up[0] = x[0],
// up[2 D - 1] = x[D - 1] (ATen's clamp), zero outside (the convolution's padding of the upsampled grid).  The Winograd transform
// V = B^T x B is linear, so the depth blend is taken on the TRANSFORMED planes: every low-resolution plane is transformed once
// (X_j, fp32) and serves two upsampled planes V_up = X_j + c (X_(j+1) - X_j), c = 0.25 | 0.75, which are split (hi | lo fp16) and
// multiplied exactly as csrc/conv3d_wino.hip multiplies its planes: y = A^T [ sum_kd U_kd (.) V_up[o + kd - 1] ] A with
// U = G ((M_h x M_w) w[kd]) G^T.  No boundary classes along D (the polyphase-in-D form folds the clamp into other weights for the first
// and the last plane: a second resident weight set, or linear combinations on scaled copies of V), the transform is shared by the two
// depth phases, and a workgroup's role is the row phase ph alone: its two "cout tiles" are the column phases pw of the same 16
// couts, so an output cell's two tiles are the hi-res voxel pair (2 i_w, 2 i_w + 1): 128 contiguous bytes of the split-padded output,
// the addressing of conv3d_wino.hip's 32-channel records.  2.25 x fewer matrix instructions than the direct polyphase kernel
// (csrc/conv3d_rs.hip MODE 3).
//
// Cells on the H and W faces of the volume need other centre taps along that axis: csrc/conv3d_up2poly.hip's face / edge kernels have
// written the difference as RAW fp32 corrections into the output voxels (all planes, with the true depth matrices); they arrive
// through the residual's LDS-DMA path -- per unit the request offsets of voxels that are not on a face are sent out of range, so the
// ring holds zeros there -- and join the sums in front of scale / shift.
//
// Shape: conv3d_wino.hip's (4 waves, wave a = row a of the transform space, 192 weight registers in the accumulator file, a unit =
// 2 low-res rows x 32 columns marched through the 2 D upsampled planes, exchange through LDS, wave (pa, q) finishes cell (2 r + pa,
// 2 n + q)), with: TWO plane images (a low-resolution plane is consumed every second step), X_prev in LDS (32 KiB, lane-private: the
// register file has no room for two transformed planes beside V pairs, raw patches and an accumulator slot), four step types.
#include "common.hpp"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#include "split_fmt.hpp"
#include "conv3d_wino_common.hpp"

struct WinoUp2Args {
    const unsigned char* x;    // low resolution, split-padded fp16 pairs [B][D+2][H+2][W+2][128 B]
    unsigned char* y;          // high resolution, split-padded fp16 pairs [B][2D+2][2H+2][2W+2][64 B]; holds the raw face corrections on entry
    const u32x4* wp;           // [ph 2][a 4][b 4][kd 3][pw 2][hi | lo][64 lanes] 16-byte fragments
    const float* unw;          // [ph 2][pw 2][16]: inverse of the Winograd weights' power-of-two pre-scaling
    const float* scale;        // [16] (carries the inverse of the plan's pre-scaling)
    const float* shift;        // [16]
    int B, D, H, W;            // low resolution; D == 8
    float neg_slope;
    int tiles_h, groups_w, total_units;      // units per role
    unsigned* sat;             // the range report's words (csrc/api.cpp)
};

namespace wu {
constexpr int HALF = 17, PITCH = 5, RPITCH = 9;   // as conv3d_wino.hip: plane sub-images | correction records
constexpr int NVOX = 4 * 2 * HALF;
constexpr int SUB = (NVOX * PITCH + 15) / 16 * 16;
constexpr int PLANE_SLOTS = 2 * SUB;
constexpr int NDMA = 22, DPW = 6;
constexpr int PLANE_LDS = NDMA * 1024;            // 22,528
constexpr int NBUF = 2;                           // low-resolution plane j lives in image j & 1
constexpr int ZB = NBUF * PLANE_LDS;              // the exchange: 2 x 16 KiB
constexpr int RB = ZB + 2 * 16384;                // correction records of one output plane of a unit: ring of three 12 KiB images
constexpr int RDPW = 3, RES_LDS = 4 * RDPW * 1024, NRES = 3;
constexpr int XB = RB + NRES * RES_LDS;           // X_prev: [b 4][channel pair 4][256 lanes][2 floats] = 32 KiB, lane-private
constexpr int DUMMY = XB + 32768;
constexpr int LDS_BYTES = DUMMY + 1024;           // 148,480
constexpr int DEPTH = 16;                         // upsampled planes of a unit (D == 8)
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
enum { PLAIN = 0, EVEN = 1, ODD = 2, COPY = 3 };
// step s (MFMAs on V_up[s]) builds V_up[s + 1]:  s even -> 2 j + 1 = 0.75 X_j + 0.25 X_(j+1) with X_(j+1) transformed in this step
// (EVEN), s odd -> 2 j + 2 = 0.25 X_j + 0.75 X_(j+1) (ODD; X_(j+1) becomes X_prev), s = 14 -> V_up[15] = X_7 (COPY), s = 15 -> the
// NEXT unit's V_up[0] = X'_0 (PLAIN)
__host__ __device__ constexpr int step_type(int s) { return s == 15 ? PLAIN : (s == 14 ? COPY : ((s & 1) ? ODD : EVEN)); }
// the step whose first slots request a low-resolution plane, and which: plane q goes to image q & 1, free one barrier after the
// raw patches of plane q - 2 were read from it (read at the end of steps 14' 15' 1 3 5 7 9 11 for planes 0 .. 7)
__host__ __device__ constexpr int step_request(int s) { return s == 0 ? 2 : (s == 1 ? 3 : (s == 3 ? 4 : (s == 5 ? 5 : (s == 7 ? 6 : (s == 9 ? 7 : (s == 11 ? 8 : (s == 13 ? 9 : -1))))))); }   // 8, 9: planes 0, 1 of the next unit
// raw patches read at the END of step s (for the transform in step s + 1): plane index, or -1
__host__ __device__ constexpr int step_read(int s) { return s == 14 ? 8 : (s == 15 ? 9 : ((s & 1) && s <= 11 ? (s + 3) / 2 : -1)); }
}  // namespace wu

// (lo-half value, hi-half value) -> the split's packed hi and lo dwords under MODE.FP16_OVFL: the conversions saturate by themselves
__device__ __forceinline__ void split_pair_ovfl(float v0, float v1, unsigned& hi, unsigned& lo, float& satm) {
    satm = sf_sat_acc(satm, v0, v1);         // range report (csrc/split_fmt.hpp): on the unclamped values
    hi = sf_cvt_pk<true>(v0, v1);
    lo = sf_cvt_pk<true>(mix_sub_lo(v0, hi), mix_sub_hi(v1, hi));
}

__global__ __launch_bounds__(256, 1) void conv3d_wino_up2_kernel(WinoUp2Args a) {
    using namespace wu;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // MODE.FP16_OVFL (hwreg 1, bit 23): an fp32 -> fp16 conversion whose result overflows gives +-65504 instead of +-inf
    // (tools/ubench/fp16_ovfl_trapsts.hip: v_cvt_pk_f16_f32(70000) = 0x7bff; true infinities stay infinities) -- the range clamp
    // of the fp16 split (one v_med3_f32 per value, 40 per step of this vector-issue-bound kernel) for free.  The range report
    // still sees every value in front of its conversion.
    __builtin_amdgcn_s_setreg((1 - 1) << 11 | 23 << 6 | 1, 1u);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);                 // this wave's row of the transform space
    const int n = lane & 15, kg = lane >> 4;
    const int Hp = a.H + 2, Wp = a.W + 2;
    const long long plane_bytes = (long long)Hp * Wp * 128;
    const long long frame_bytes = (a.D + 2) * plane_bytes;
    const long long total_bytes = frame_bytes * a.B;
    const int Hq = 2 * a.H + 2, Wq = 2 * a.W + 2;                            // rows / columns of the output tensor (border included)
    const long long oplane_bytes = (long long)Hq * Wq * 64;
    const long long oframe_bytes = (2 * a.D + 2) * oplane_bytes;
    const long long ototal_bytes = oframe_bytes * a.B;
    // B^T row a has two non-zeros: t = x[i0] + sgn * x[i1]
    const int i0 = wv == 0 ? 0 : (wv == 2 ? 2 : 1);
    const int i1 = wv == 0 ? 2 : (wv == 1 ? 2 : (wv == 2 ? 1 : 3));
    const float sgn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(wv == 1 ? 0x3f800000 : 0xbf800000));
    const int pa = wv & 1, q = wv >> 1;                                      // after the exchange: this wave's cell of every tile
    const float osg = pa ? -1.f : 1.f;

    // ---- the workgroup's role and walk: blockIdx = (walker * 2 + ph) * 8 + xcd.  XCD x owns a contiguous eighth of the unit space;
    // walker w of its `nwalk` takes every nwalk-th unit, for both row phases at about the same time (they read the same input) ----
    const int G = gridDim.x, xcd = (int)blockIdx.x & 7, iw = (int)blockIdx.x >> 3;
    const int ph = iw & 1, walker = iw >> 1, nwalk = G >> 4;
    const int T_ = a.total_units, q8 = T_ >> 3, r8 = T_ & 7;
    const int ulo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, ucnt = q8 + (xcd < r8 ? 1 : 0);
    const int ubase = ulo + walker, ustep = nwalk;
    const int nmine = walker < ucnt ? (ucnt - walker + nwalk - 1) / nwalk : 0;
    if (nmine == 0) return;         // (the whole workgroup, before anything was requested)

    // ---- weights: resident for the whole launch ----
    u32x4 wh[4][3][2], wl[4][3][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const u32x4* p_ = a.wp + ((long long)(((((ph * 4 + wv) * 4 + b) * 3 + kd) * 2 + c) * 2)) * 64 + lane;
                wh[b][kd][c] = p_[0];
                wl[b][kd][c] = p_[64];
            }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int c = 0; c < 2; ++c) asm volatile("" : "+a"(wh[b][kd][c]), "+a"(wl[b][kd][c]));
    // epilogue constants of this lane's couts 4 kg .. 4 kg + 3: result = act(main * (unw * scale) + (correction * scale + shift))
    const f32x4 esc_ = *reinterpret_cast<const f32x4*>(a.scale + kg * 4), esh_ = *reinterpret_cast<const f32x4*>(a.shift + kg * 4);
    const f32x4 ema_[2] = {*reinterpret_cast<const f32x4*>(a.unw + (ph * 2 + 0) * 16 + kg * 4) * esc_,
                           *reinterpret_cast<const f32x4*>(a.unw + (ph * 2 + 1) * 16 + kg * 4) * esc_};
    __builtin_amdgcn_sched_barrier(0);

    const int lane_out = (2 * n) * 128 + (kg >> 1) * 16 + (kg & 1) * 8;       // this lane's 8-byte hi piece of cell 2 n's voxel pair, pw = 0
    // fragment reads: this lane's hi piece (slice kg >> 1, channel half kg & 1) of column 2 n in patch rows i0 / i1, image 0
    const int rd0 = ((kg & 1) * SUB + (i0 * 2 * HALF + n) * PITCH + (kg >> 1) * 2) * 16;
    const int rd1 = ((kg & 1) * SUB + (i1 * 2 * HALF + n) * PITCH + (kg >> 1) * 2) * 16;
    constexpr int RD2 = 16;                           // from a lane's hi piece to its lo piece
    // ---- DMA plans ----
    unsigned voff[DPW], rvoff[RDPW];
    int rrow[RDPW], rcol[RDPW];                       // correction voxel of a request lane: low-res row (0 | 1) and column (0 .. 31) of the unit
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
        const int m = wv + 4 * k;
        const int sl = m * 64 + lane;
        const int sub = sl >= SUB ? 1 : 0, rem = sl - sub * SUB;
        const int vox = rem / PITCH, c = rem - vox * PITCH;
        const int row = vox / (2 * HALF), rr = vox - row * (2 * HALF);
        const int par = rr / HALF, idx = rr - par * HALF;
        const int piece = c * 2 + sub;                // fp16 pairs [slice][hi | lo][channels 0-7 | 8-15]: even groups read pieces 0, 2 | 4, 6, odd ones 1, 3 | 5, 7
        voff[k] = (m < NDMA && sl < PLANE_SLOTS && vox < NVOX && c < 4) ? (unsigned)((row * Wp + 2 * idx + par) * 128 + piece * 16)
                                                                        : 0xffffff00u;          // pads: beyond num_records, zero-filled
    }
#pragma unroll
    for (int k = 0; k < RDPW; ++k) {
        const int sl = (wv + 4 * k) * 64 + lane;
        const int vox = sl / RPITCH, piece = sl - vox * RPITCH;
        rrow[k] = vox >> 5;
        rcol[k] = vox & 31;
        // the unit's two low-res rows are hi-res rows 2 (2 r + row) + ph: two rows of the output tensor apart; a cell = 128 bytes
        rvoff[k] = (vox < 64 && piece < 8) ? (unsigned)((vox >> 5) * 2 * Wq * 64 + (vox & 31) * 128 + piece * 16) : 0xffffff00u;
    }
    // the epilogue's correction reads: cell (row pa, column 2 n + q), fp32 couts 4 kg .. 4 kg + 3 of the pw = 0 voxel (pw = 1: 64 B on)
    const int rrd = ((pa * 32 + 2 * n + q) * RPITCH + kg) * 16;
    const int xoff = XB + tid * 8;                    // this lane's X_prev slots: + (b * 4 + d) * 2048

    f32x4 Y[3][4][2];
    u32x4 vh[2][4], vl[2][4];      // V_up of the plane being multiplied | of the next one
    u32x4 raw[2][4][2];            // [patch row i0 | i1][column][hi | lo] of the low-resolution plane being transformed
    float T[2][4];                 // h-pass of one channel pair: [lo | hi half][column]
    f32x2 Xc[4][4];                // [b][channel pair]: the transformed low-resolution plane X_(j+1) (EVEN -> ODD)
    f32x2 Pq[4];                   // X_prev of one channel pair, all b, on its way from LDS
    float satm = 0.f;              // running maximum |clamped value| (range report, csrc/split_fmt.hpp)

// descriptor whose base is the patch origin of unit U (padded rows 2 r .., columns 32 c ..); no records past the workgroup's last unit
#define KU_DESC(U, LIVE)                                                                                    \
    ({                                                                                                      \
        const int c_ = (U) % a.groups_w, t_ = (U) / a.groups_w;                                             \
        const int r_ = t_ % a.tiles_h, b_ = t_ / a.tiles_h;                                                 \
        const long long off_ = (LIVE) ? b_ * frame_bytes + ((long long)(2 * r_) * Wp + 32 * c_) * 128 : 0;  \
        const long long left_ = total_bytes - off_;                                                         \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.x) + off_, 0,                        \
                                          (LIVE) ? (left_ > 0x7fffff00ll ? 0x7fffff00 : (int)left_) : 0, 0x00020000); \
    })
// low-resolution plane Q (0 .. 7 of this unit, 8 | 9 = planes 0 | 1 of the next) -> image Q & 1, pieces K0 .. K1 - 1 of this wave
#define KU_DMA_ISSUE(Q, K0, K1)                                                                             \
    _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_)                                                  \
        __builtin_amdgcn_raw_ptr_buffer_load_lds((Q) < 8 ? dsc : dsc_next,                                  \
                                                 (__attribute__((address_space(3))) void*)(lds + (wv + 4 * k_ < NDMA ? ((Q) & 1) * PLANE_LDS + (wv + 4 * k_) * 1024 : DUMMY)), \
                                                 16, voff[k_], (unsigned)(((Q) % 8 + 1) * plane_bytes), 0, 0);
// the raw corrections of output plane O of this unit -> image IMG of the ring (zeros where the cell is not on a face)
#define KU_RES_DMA(IMG, O)                                                                                  \
    {                                                                                                       \
        const unsigned so_ = (unsigned)(((O) + 1) * oplane_bytes);                                          \
        _Pragma("unroll") for (int k_ = 0; k_ < RDPW; ++k_)                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdsc, (__attribute__((address_space(3))) void*)(lds + RB + (IMG) * RES_LDS + (wv + 4 * k_) * 1024), \
                                                     16, rvu[k_], so_, 0, 0);                               \
    }
#define KU_READ(BUF_)                                                                                       \
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
// One piece of the work that turns raw / X_prev / Xc into V[VN] of type TY, sized to ride behind one MFMA: M = 0 .. 71 -> channel pair
// d = M / 18 and, within it: pieces 0 .. 7 the h-pass of one (column, half) each (PLAIN, EVEN: three dependent mixed-precision fmas)
// and, in pieces 0 .. 3, the request of X_prev[b = piece][d] from LDS (all but PLAIN); then per b one piece of w-pass + depth blend +
// range clamp and one of the split; piece 16 closes the last b.
#define KU_TPIECE(M, VN, TY)                                                                                \
    {                                                                                                       \
        constexpr int d_ = (M) / 18, r_ = (M) % 18;                                                         \
        if constexpr (r_ < 8) {                                                                             \
            constexpr int j_ = r_ >> 1;                                                                     \
            if constexpr ((TY) != PLAIN && r_ < 4)                                                          \
                Pq[r_] = *reinterpret_cast<const f32x2*>(lds + xoff + (r_ * 4 + d_) * 2048);                \
            if constexpr ((TY) == PLAIN || (TY) == EVEN) {                                                  \
                /* ONE asm statement per chain: hipcc pads every VGPR an asm statement defines against a use by the very next \
                   instruction, 4 cycles per pad, and only its own instructions count as distance */       \
                if constexpr ((r_ & 1) == 0) {                                                              \
                    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"                \
                        "v_fma_mix_f32 %0, %3, %5, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"                 \
                        "v_fma_mix_f32 %0, %4, %5, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]"                      \
                        : "=&v"(T[0][j_]) : "v"(raw[0][j_][0][d_]), "v"(raw[0][j_][1][d_]), "v"(raw[1][j_][0][d_]), "v"(raw[1][j_][1][d_]), "s"(sgn)); \
                } else {                                                                                    \
                    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n\t"                \
                        "v_fma_mix_f32 %0, %3, %5, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"                 \
                        "v_fma_mix_f32 %0, %4, %5, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]"                      \
                        : "=&v"(T[1][j_]) : "v"(raw[0][j_][0][d_]), "v"(raw[0][j_][1][d_]), "v"(raw[1][j_][0][d_]), "v"(raw[1][j_][1][d_]), "s"(sgn)); \
                }                                                                                           \
            }                                                                                               \
        } else if constexpr (r_ < 16) {                                                                     \
            constexpr int b_ = (r_ - 8) >> 1;                                                               \
            constexpr int x_ = b_ == 0 ? 0 : (b_ == 2 ? 2 : 1), y_ = b_ == 0 ? 2 : (b_ == 1 ? 2 : (b_ == 2 ? 1 : 3)); \
            if constexpr (((r_ - 8) & 1) == 0) {                                                            \
                if constexpr ((TY) == PLAIN || (TY) == EVEN) {                                              \
                    X0 = b_ == 1 ? T[0][x_] + T[0][y_] : T[0][x_] - T[0][y_];                               \
                    X1 = b_ == 1 ? T[1][x_] + T[1][y_] : T[1][x_] - T[1][y_];                               \
                }                                                                                           \
                if constexpr ((TY) == PLAIN) {            /* V_up[0] = X_0, which is also the next X_prev */ \
                    *reinterpret_cast<f32x2*>(lds + xoff + (b_ * 4 + d_) * 2048) = f32x2{X0, X1};           \
                } else if constexpr ((TY) == EVEN) {      /* 0.75 X_prev + 0.25 X */                       \
                    Xc[b_][d_] = f32x2{X0, X1};                                                             \
                    X0 = __builtin_fmaf(X0 - Pq[b_][0], 0.25f, Pq[b_][0]);                                  \
                    X1 = __builtin_fmaf(X1 - Pq[b_][1], 0.25f, Pq[b_][1]);                                  \
                } else if constexpr ((TY) == ODD) {       /* 0.25 X_prev + 0.75 X; X is the next X_prev */ \
                    X0 = __builtin_fmaf(Xc[b_][d_][0] - Pq[b_][0], 0.75f, Pq[b_][0]);                       \
                    X1 = __builtin_fmaf(Xc[b_][d_][1] - Pq[b_][1], 0.75f, Pq[b_][1]);                       \
                    *reinterpret_cast<f32x2*>(lds + xoff + (b_ * 4 + d_) * 2048) = Xc[b_][d_];              \
                } else {                                  /* V_up[2 D - 1] = X_(D-1) */                    \
                    X0 = Pq[b_][0];                                                                         \
                    X1 = Pq[b_][1];                                                                         \
                }                                                                                           \
                satm = sf_sat_acc(satm, X0, X1);          /* sums of four fp16-range values: saturated by the conversion (MODE.FP16_OVFL) */ \
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
// staging requests of step S: the low-resolution plane of wu::step_request in the step's first slots (its image was released by the
// barrier in front of the step), the corrections of output plane S behind them
#define KU_DPIECE(M, S)                                                                                     \
    if constexpr (step_request(S) >= 0 && (M) >= 1 && (M) <= 3) { KU_DMA_ISSUE(step_request(S), 2 * ((M) - 1), 2 * (M)) } \
    else if constexpr ((M) == 17) { KU_RES_DMA(ri, S) }
// the finished accumulators of slot SF leave through A^T: over b inside the wave, over a across the waves (LDS)
#define KU_OUT_WRITE(SF, ZIMG)                                                                              \
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
#define KU_EPI_READS(ZIMG)                                                                                  \
    f32x4 zz_[2][3], rf_[2];                                                                                \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_)                                                        \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_)                                                    \
            zz_[c_][k_] = *reinterpret_cast<const f32x4*>(lds + ZB + (ZIMG) * 16384 + (((pa + k_) * 2 + q) * 2 + c_) * 1024 + lane * 16); \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) rf_[c_] = *reinterpret_cast<const f32x4*>(rim_ + rrd + c_ * 64);
// wave (pa, q) finishes cell (2 r + pa, 2 n + q) of every tile: its two column phases are the hi-res voxel pair behind yb
#define KU_EPILOGUE(O)                                                                                      \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) {                                                      \
        f32x4 t_ = zz_[c_][0] + osg * (zz_[c_][1] + zz_[c_][2]);                                            \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                                    \
            t_[e_] = __builtin_fmaf(t_[e_], ema_[c_][e_], __builtin_fmaf(rf_[c_][e_], esc_[e_], esh_[e_])); \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) t_[e_] = __builtin_fmaxf(t_[e_], t_[e_] * a.neg_slope); \
        unsigned h0_, l0_, h1_, l1_;                                                                        \
        split_pair_ovfl(t_[0], t_[1], h0_, l0_, satm);                                                      \
        split_pair_ovfl(t_[2], t_[3], h1_, l1_, satm);                                                      \
        unsigned char* q_ = yb + (long long)((O) + 1) * oplane_bytes + lane_out + c_ * 64;                  \
        *reinterpret_cast<u32x2*>(q_) = u32x2{h0_, h1_};                                                    \
        *reinterpret_cast<u32x2*>(q_ + 32) = u32x2{l0_, l1_};                                               \
    }
// one MFMA and the pieces that ride behind it (conv3d_wino.hip's order: term-major inside a depth tap, the accumulators finished in
// this step first).  Depth taps on the zero border of the UPSAMPLED grid have no MFMAs.
#define KU_SLOT(M, S, SI, SM, SF, VB)                                                                       \
    {                                                                                                       \
        constexpr int g_ = (M) / 24, w_ = (M) % 24, tm_ = w_ / 8, b_ = (w_ % 8) / 2, c_ = w_ % 2;           \
        constexpr int kd_ = 2 - g_, s_ = g_ == 0 ? (SF) : (g_ == 1 ? (SM) : (SI));                          \
        constexpr bool live_ = !(g_ == 0 && (S) == 0) && !(g_ == 2 && (S) == DEPTH - 1);                    \
        constexpr bool first_ = tm_ == 0 && (g_ == 2 || (g_ == 1 && (S) == 0));                             \
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
        KU_TPIECE(M, (VB) ^ 1, step_type(S))                                                                \
        KU_DPIECE(M, S)                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
#define KU_S8(M, S, SI, SM, SF, VB)                                                                         \
    KU_SLOT(M, S, SI, SM, SF, VB) KU_SLOT(M + 1, S, SI, SM, SF, VB) KU_SLOT(M + 2, S, SI, SM, SF, VB) KU_SLOT(M + 3, S, SI, SM, SF, VB) \
    KU_SLOT(M + 4, S, SI, SM, SF, VB) KU_SLOT(M + 5, S, SI, SM, SF, VB) KU_SLOT(M + 6, S, SI, SM, SF, VB) KU_SLOT(M + 7, S, SI, SM, SF, VB)

// One step over upsampled plane S = 0 .. 15 (a compile-time constant: a unit is ONE basic block, conv3d_wino.hip).  V_up[S] in
// vh / vl[VB]; V_up[S + 1] goes to vh / vl[VB ^ 1] piece by piece behind this plane's MFMAs.  Open output planes: S + 1 in slot SI,
// S in slot SM, S - 1 in slot SF (leaves through the exchange at the end of the step).  Requests of the step, in issue order:
// [6: a low-resolution plane, in some steps] [3: the corrections of plane S, for the epilogue one step later]; then, behind the
// barrier, the 4 output stores of plane S - 1.
#define KU_STEP(S, SI, SM, SF, VB)                                                                          \
    {                                                                                                       \
        unsigned char* rim_ = lds + RB + rprev * RES_LDS;                                                   \
        asm volatile("s_nop 1");                                                                            \
        {                                                                                                   \
            float X0 = 0.f, X1 = 0.f, L0 = 0.f, L1 = 0.f;                                                   \
            KU_S8(0, S, SI, SM, SF, VB) KU_S8(8, S, SI, SM, SF, VB) KU_S8(16, S, SI, SM, SF, VB)            \
            KU_S8(24, S, SI, SM, SF, VB) KU_S8(32, S, SI, SM, SF, VB) KU_S8(40, S, SI, SM, SF, VB)          \
            KU_S8(48, S, SI, SM, SF, VB) KU_S8(56, S, SI, SM, SF, VB) KU_S8(64, S, SI, SM, SF, VB)          \
        }                                                                                                   \
        WN_PAD()                                                                                            \
        if constexpr ((S) > 0) { KU_OUT_WRITE(SF, ((S) - 1) & 1) }                                          \
        /* everything older than this step's requests has landed: the corrections asked for a step ago, the plane image read below */ \
        if constexpr (step_request(S) >= 0) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");     \
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");                                    \
        __builtin_amdgcn_s_barrier();                                                                       \
        if constexpr ((S) > 0) {                                                                            \
            KU_EPI_READS(((S) - 1) & 1)                                                                     \
            KU_EPILOGUE((S) - 1)                                                                            \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if constexpr (step_read(S) >= 0) { KU_READ(step_read(S) & 1) }                                      \
        rprev = ri;                                                                                         \
        ri = ri == NRES - 1 ? 0 : ri + 1;                                                                   \
    }
// the unit's last output plane (its slot got the last contribution in step 15: plane 16 is the zero border).  Its corrections were
// requested in step 15, in front of nothing else but that step's 4 output stores.
#define KU_FINISH(SF)                                                                                       \
    {                                                                                                       \
        unsigned char* rim_ = lds + RB + rprev * RES_LDS;                                                   \
        KU_OUT_WRITE(SF, (DEPTH - 1) & 1)                                                                   \
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                                         \
        __builtin_amdgcn_s_barrier();                                                                       \
        KU_EPI_READS((DEPTH - 1) & 1)                                                                       \
        KU_EPILOGUE(DEPTH - 1)                                                                              \
    }
#define KU_T8(M, TY) KU_TPIECE(M, 0, TY) KU_TPIECE(M + 1, 0, TY) KU_TPIECE(M + 2, 0, TY) KU_TPIECE(M + 3, 0, TY) KU_TPIECE(M + 4, 0, TY) KU_TPIECE(M + 5, 0, TY) KU_TPIECE(M + 6, 0, TY) KU_TPIECE(M + 7, 0, TY)

    int ri = 0, rprev = 0;                             // correction image to request into / requested a step ago
    auto dsc = KU_DESC(ubase, true);
    auto dsc_next = dsc;
    // what steps 14 and 15 of a unit do for the next one: planes 0 and 1 staged, V_up[0] = X_0 built (and stored as X_prev), the raw
    // patches of plane 1 read
    KU_DMA_ISSUE(0, 0, DPW)
    KU_DMA_ISSUE(1, 0, DPW)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    KU_READ(0)
    {
        float X0 = 0.f, X1 = 0.f, L0 = 0.f, L1 = 0.f;
        KU_T8(0, PLAIN) KU_T8(8, PLAIN) KU_T8(16, PLAIN) KU_T8(24, PLAIN) KU_T8(32, PLAIN) KU_T8(40, PLAIN) KU_T8(48, PLAIN) KU_T8(56, PLAIN) KU_T8(64, PLAIN)
    }
    KU_READ(1)
    __builtin_amdgcn_s_barrier();                      // (every wave has its patches of plane 0: image 0 may be requested into)
    for (int k = 0; k < nmine; ++k) {
        const int u = ubase + k * ustep;
        const int c = u % a.groups_w;
        const int t = u / a.groups_w;
        const int r = t % a.tiles_h;
        const int b = t / a.tiles_h;
        // this wave's cell origin in the output: hi-res row 2 (2 r + pa) + ph, voxel pair of low-res column 32 c + q (+ 1: the border)
        unsigned char* yb = a.y + b * oframe_bytes + ((long long)(2 * (2 * r + pa) + ph + 1) * Wq + 2 * (32 * c + q) + 1) * 64;
        // the unit's cells in the output tensor, row pa = 0, pw = 0 of column 32 c: where the raw corrections sit
        const long long roff = b * oframe_bytes + ((long long)(2 * (2 * r) + ph + 1) * Wq + 2 * (32 * c) + 1) * 64;
        const long long rleft = ototal_bytes - roff;
        const auto rdsc = __builtin_amdgcn_make_buffer_rsrc(a.y + roff, 0, rleft > 0x7fffff00ll ? 0x7fffff00 : (int)rleft, 0x00020000);
        // only cells on an H or W face of the volume hold a correction: everything else is requested out of range (zeros)
        unsigned rvu[RDPW];
#pragma unroll
        for (int k_ = 0; k_ < RDPW; ++k_) {
            const int ih_ = 2 * r + rrow[k_], iw_ = 32 * c + rcol[k_];
            const bool face_ = ih_ == 0 || ih_ == a.H - 1 || iw_ == 0 || iw_ == a.W - 1;
            rvu[k_] = face_ ? rvoff[k_] : 0xffffff00u;
        }
        dsc_next = KU_DESC(u + ustep, k + 1 < nmine);
        // slots follow the plane mod 3, the V pair the plane mod 2
#define KU_STEPP(S) KU_STEP(S, ((S) + 1) % 3, (S) % 3, ((S) + 2) % 3, (S) & 1)
        KU_STEPP(0) KU_STEPP(1) KU_STEPP(2) KU_STEPP(3) KU_STEPP(4) KU_STEPP(5) KU_STEPP(6) KU_STEPP(7)
        KU_STEPP(8) KU_STEPP(9) KU_STEPP(10) KU_STEPP(11) KU_STEPP(12) KU_STEPP(13) KU_STEPP(14) KU_STEPP(15)
        KU_FINISH((DEPTH - 1) % 3)
        dsc = dsc_next;
    }
    sf_sat_report(a.sat, kSatSplit, satm, kF16Max);
}

}  // namespace

namespace mvsgi {

// U = G g G^T per (cout, cin, kd) of the 32 -> 32 "row phase" layer [pw * 16 + co][ci][kd][th][tw] (folded along H and W, untouched
// along D), pre-scaled per cout by a power of two so that max |U| lies in (512, 1024], split into fp16 pairs, in the kernel's A-operand
// order [a][b][kd][pw][hi | lo][lane = (cin group) * 16 + cout][8 cins]; unscale[pw * 16 + co] = 2^-k.  Host code (the plan builder).
void wino_up2_pack_role(const float* w32, unsigned short* wp, float* unscale) {
    static const double Gm[4][3] = {{1., 0., 0.}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0., 0., 1.}};
    for (int co = 0; co < 32; ++co) {
        double amax = 0.;
        std::vector<double> Uc((size_t)32 * 48);
        for (int ci = 0; ci < 32; ++ci) {
            const float* g = w32 + ((size_t)co * 32 + ci) * 27;
            for (int ab = 0; ab < 16; ++ab)
                for (int kd = 0; kd < 3; ++kd) {
                    double acc = 0.;
                    for (int kh = 0; kh < 3; ++kh)
                        for (int kw = 0; kw < 3; ++kw) acc += Gm[ab >> 2][kh] * Gm[ab & 3][kw] * (double)g[kd * 9 + kh * 3 + kw];
                    Uc[(size_t)ci * 48 + ab * 3 + kd] = acc;
                    amax = std::fmax(amax, std::fabs(acc));
                }
        }
        int k = 0;
        if (amax > 0.) {
            int e;
            const double m = std::frexp(amax, &e);       // amax = m * 2^e, m in [0.5, 1)
            k = m == 0.5 ? 11 - e : 10 - e;               // amax * 2^k in (512, 1024]
            k = k < -100 ? -100 : (k > 100 ? 100 : k);
        }
        const double up = std::ldexp(1., k);
        unscale[co] = (float)std::ldexp(1., -k);
        for (int ci = 0; ci < 32; ++ci)
            for (int ab = 0; ab < 16; ++ab)
                for (int kd = 0; kd < 3; ++kd) {
                    unsigned short hi, lo;
                    sf_split_weight((float)(Uc[(size_t)ci * 48 + ab * 3 + kd] * up), true, hi, lo);
                    const size_t frag = (((size_t)ab * 3 + kd) * 2 + (co >> 4)) * 2;
                    const int lane = (ci >> 3) * 16 + (co & 15);
                    wp[((frag + 0) * 64 + lane) * 8 + (ci & 7)] = hi;
                    wp[((frag + 1) * 64 + lane) * 8 + (ci & 7)] = lo;
                }
    }
}

bool wino_up2_applies(int D, int H, int W) {
    return D == 8 && H > 0 && W > 0 && H % 2 == 0 && W % 32 == 0 && (long long)(D + 2) * (H + 2) * (W + 2) * 128 < (1ll << 31) &&
           (long long)(2 * D + 2) * (2 * H + 2) * (2 * W + 2) * 64 < (1ll << 31);
}

// the main kernel of the polyphase layer in this form: x_split low-res fp16 pairs, y_split the split-padded hi-res output holding the
// face / edge kernels' raw corrections; w_roles / unw from the plan (conv3d_up2poly.hip)
int wino_up2_launch(const void* x_split, const void* w_roles, const float* unw, const float* scale16, const float* shift16, void* y_split,
                    int B, int D, int H, int W, float neg_slope, hipStream_t st) {
    MVSGI_REQUIRE(wino_up2_applies(D, H, W), "mvsgi_conv3d_up2_poly(winograd): needs D == 8, H %% 2 == 0, W %% 32 == 0 (got %d, %d, %d)", D, H, W);
    WinoUp2Args a;
    memset(&a, 0, sizeof(a));
    a.x = static_cast<const unsigned char*>(x_split);
    a.y = static_cast<unsigned char*>(y_split);
    a.wp = static_cast<const u32x4*>(w_roles);
    a.unw = unw;
    a.scale = scale16;
    a.shift = shift16;
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.neg_slope = neg_slope;
    a.tiles_h = H / 2;
    a.groups_w = W / 32;
    const long long units = (long long)B * a.tiles_h * a.groups_w;
    MVSGI_REQUIRE(units < (1ll << 30), "mvsgi_conv3d_up2_poly(winograd): too many units");
    a.total_units = (int)units;
    MVSGI_SAT_WORDS(sat_words_);
    a.sat = sat_words_;
    static PersistentGeom geo_cache[kMaxDevices] = {};
    PersistentGeom geo;
    if (persistent_geometry(conv3d_wino_up2_kernel, 256, wu::LDS_BYTES, 1, geo_cache, "conv3d(winograd up2)", geo)) return 1;
    // grid = 8 XCDs x 2 row phases x walkers: one workgroup per CU, never more walkers than the largest XCD share has units
    long long nwalk = geo.cus / 16;
    const long long most = cdiv(units, 8);
    if (nwalk > most) nwalk = most;
    if (nwalk < 1) nwalk = 1;
    hipLaunchKernelGGL(conv3d_wino_up2_kernel, dim3((unsigned)(16 * nwalk)), dim3(256), wu::LDS_BYTES, st, a);
    return check_launch("mvsgi_conv3d_up2_poly(winograd main)");
}

}  // namespace mvsgi
