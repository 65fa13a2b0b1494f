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
// A unit is 16 tiles in a row (2 output rows x 32 columns) marched through all D planes: per plane a lane (tile n, channel group kg)
// reads the two patch rows its wave's `a` needs (B^T has two non-zeros per row), transforms 8 channels in registers straight into
// the MFMA's B-operand layout (no LDS round trip for V), and issues 72 MFMAs: plane p adds U_kd V_p to the open output planes
// p + 1 - kd.  A finished output plane leaves through A^T: the (b) half inside the wave, the (a) half across the waves through a
// 16 KiB LDS exchange, after which wave (pa, q) owns output voxel (2 r + pa, 2 n + q) of every tile: scale / shift, residual,
// LeakyReLU, split, store.  Tensors are split-padded (conv3d_rs.hip): the zero border is the convolution's padding.
#include "common.hpp"

#include <cstring>
#include <cstdlib>
#include <cstdio>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#include "split_fmt.hpp"

struct WinoArgs {
    const unsigned char* x;    // split-padded fp16 [B][D+2][H+2][W+2][32]
    unsigned char* y;          // the same geometry
    const unsigned char* res;  // split-padded residual or nullptr
    const u32x4* wp;           // [a 4][b 4][kd 3][cout tile 2][hi | lo][64 lanes] 16-byte fragments
    const float* scale;        // (carries the inverse of the weights' pre-scaling)
    const float* shift;
    int B, D, H, W;
    float neg_slope;
    int tiles_h, groups_w, total_units;
    unsigned long long* dbg;   // diagnostic (MVSGI_WINO_ABL bit 16): per wave of workgroup 0, cycles per step section
};

// fp32 <- f16 half of a dword through the mixed-precision fma (one instruction where widen + add are two or three)
__device__ __forceinline__ float mix_sum_lo(unsigned h, unsigned l) {       // f16lo(h) + f16lo(l)
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
    return r;
}
__device__ __forceinline__ float mix_sum_hi(unsigned h, unsigned l) {       // f16hi(h) + f16hi(l)
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
    return r;
}
__device__ __forceinline__ float mix_fma_lo(unsigned h, float s, float c) {  // f16lo(h) * s + c
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(s), "v"(c));
    return r;
}
__device__ __forceinline__ float mix_fma_hi(unsigned h, float s, float c) {  // f16hi(h) * s + c
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(s), "v"(c));
    return r;
}
__device__ __forceinline__ float mix_add_lo(unsigned h, float c) {           // f16lo(h) + c
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ float mix_add_hi(unsigned h, float c) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ float mix_sub_lo(float c, unsigned h) {           // c - f16lo(h)
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
    return r;
}
__device__ __forceinline__ float mix_sub_hi(float c, unsigned h) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(c));
    return r;
}
// (lo-half value, hi-half value) -> the split's packed hi and lo dwords (clamped to fp16's range first)
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
    v0 = sf_clamp<true>(v0);
    v1 = sf_clamp<true>(v1);
    hi = sf_cvt_pk<true>(v0, v1);
    lo = sf_cvt_pk<true>(mix_sub_lo(v0, hi), mix_sub_hi(v1, hi));
}

// The MFMAs are inline asm so that the weights can be pinned to the accumulator half of the register file ("a"); the output
// accumulators and the transformed activations are ordinary registers.  hipcc pads no hazards around asm: every group of MFMAs
// ends in WN_PAD before anything else may touch its accumulators.
#define WN_MF(ACC, WREG, XREG) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(WREG), "v"(XREG));
#define WN_MF0(ACC, WREG, XREG) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "+v"(ACC) : "a"(WREG), "v"(XREG));
#define WN_PAD() asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
// the 24 MFMAs of one depth tap: term-major, so that the three products of an accumulator are 8 MFMAs apart
#define WN_GROUP(S, KD, MF_FIRST)                                                                   \
    _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                                \
        _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) { MF_FIRST(Y[S][b_][c_], wl[b_][KD][c_], vh[b_]) } \
    _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                                \
        _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) { WN_MF(Y[S][b_][c_], wh[b_][KD][c_], vl[b_]) } \
    _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                                \
        _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) { WN_MF(Y[S][b_][c_], wh[b_][KD][c_], vh[b_]) }

namespace wn {
// LDS image of one input plane of a unit: 4 rows x 34 columns, even and odd columns apart, 9 sixteen-byte slots per voxel (its 8 pieces
// [slice][hi | lo][channels 0-7 | 8-15] as they lie in the tensor + 1 pad): the 16 lanes of a ds_read_b128 group -- one piece of the
// same-parity columns 2 n + j -- then walk the slots with stride 9, odd, i.e. all 16 sixteen-byte units of the 256-byte bank row;
// the staging side stays a copy of whole 128-byte voxel records (8 of every 9 consecutive lanes of a DMA instruction).
constexpr int HALF = 17, PITCH = 9;
constexpr int ROW_SLOTS = 2 * HALF * PITCH;       // 306
constexpr int PLANE_SLOTS = 4 * ROW_SLOTS;        // 1224
constexpr int NDMA = 20, DPW = NDMA / 4;          // 1 KiB pieces per plane, per wave
constexpr int PLANE_LDS = NDMA * 1024;            // 20,480
constexpr int NBUF = 3;
constexpr int ZB = NBUF * PLANE_LDS;              // the exchange: 2 x [a 4][q 2][cout tile 2][64 lanes][16 B]
// residual records of one output plane of a unit: 2 rows x 32 voxels at the same 9-slot pitch (the epilogue's 8-byte reads walk them
// with stride 18 slots: two-way conflicts at worst), 576 slots in 9 one-KiB pieces; two images (one barrier per step)
constexpr int RB = ZB + 2 * 16384, RDPW = 3, RES_LDS = 4 * RDPW * 1024;
constexpr int LDS_BYTES = RB + 2 * RES_LDS;       // 118,784
static_assert(PLANE_SLOTS <= NDMA * 64, "DMA pieces cover the image");
}  // namespace wn

template <int ABL>
__global__ __launch_bounds__(256, 1) void conv3d_wino32_kernel(WinoArgs a) {
    using namespace wn;
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
    __builtin_amdgcn_sched_barrier(0);

    const f32x4 esc[2] = {*reinterpret_cast<const f32x4*>(a.scale + kg * 4), *reinterpret_cast<const f32x4*>(a.scale + 16 + kg * 4)};
    const f32x4 esh[2] = {*reinterpret_cast<const f32x4*>(a.shift + kg * 4), *reinterpret_cast<const f32x4*>(a.shift + 16 + kg * 4)};
    const int lane_out = (2 * n) * 128 + (kg >> 1) * 16 + (kg & 1) * 8;       // this lane's 8-byte hi piece of an output voxel's slice 0
    // fragment reads: this lane's hi piece (slice kg >> 1, channel half kg & 1) of column 2 n in patch rows i0 / i1, image 0
    const int rd0 = ((i0 * 2 * HALF + n) * PITCH + (kg >> 1) * 4 + (kg & 1)) * 16;
    const int rd1 = ((i1 * 2 * HALF + n) * PITCH + (kg >> 1) * 4 + (kg & 1)) * 16;
    // ---- DMA plan: piece m = wv + 4 k fills slots [64 m, 64 m + 64) of an image ----
    unsigned voff[DPW];
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
        const int sl = (wv + 4 * k) * 64 + lane;
        const int vox = sl / PITCH, piece = sl - vox * PITCH;
        const int row = vox / (2 * HALF), rr = vox - row * (2 * HALF);
        const int par = rr / HALF, idx = rr - par * HALF;
        voff[k] = (sl < PLANE_SLOTS && piece < 8) ? (unsigned)((row * Wp + 2 * idx + par) * 128 + piece * 16)
                                                  : 0xffffff00u;          // pads: beyond num_records, zero-filled
    }

    unsigned rvoff[RDPW];
#pragma unroll
    for (int k = 0; k < RDPW; ++k) {
        const int sl = (wv + 4 * k) * 64 + lane;
        const int vox = sl / PITCH, piece = sl - vox * PITCH;
        rvoff[k] = (vox < 64 && piece < 8) ? (unsigned)(((vox >> 5) * Wp + (vox & 31)) * 128 + piece * 16) : 0xffffff00u;
    }
    // the epilogue's residual reads: voxel (row pa, column 2 n + q), this lane's 8 bytes of slice 0's hi piece
    const int rrd = ((pa * 32 + 2 * n + q) * PITCH + (kg >> 1)) * 16 + (kg & 1) * 8;

    f32x4 Y[3][4][2];
    u32x4 vh[4], vl[4];
    u32x4 raw[2][4][2];            // [patch row i0 | i1][column][hi | lo]

    const int G = gridDim.x;
    const int nmine = ((int)a.total_units - (int)blockIdx.x + G - 1) / G;     // units of this workgroup
    const int nstream = nmine * (a.D + 1);                                     // its plane stream: per unit the real planes 0 .. D - 1 and the zero border D

// descriptor whose base is the patch origin of unit U (padded rows 2 r .., columns 32 c ..)
#define WN_DESC(U)                                                                                          \
    ({                                                                                                      \
        const int c_ = (U) % a.groups_w, t_ = (U) / a.groups_w;                                             \
        const int r_ = t_ % a.tiles_h, b_ = t_ / a.tiles_h;                                                 \
        const long long off_ = b_ * frame_bytes + ((long long)(2 * r_) * Wp + 32 * c_) * 128;               \
        const long long left_ = total_bytes - off_;                                                         \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.x) + off_, 0,                        \
                                          left_ > 0x7fffff00ll ? 0x7fffff00 : (int)left_, 0x00020000);      \
    })
// the staging stream runs two planes ahead of the arithmetic
    int dk = 0, dp = 0, dg = 0, dbuf = 0;             // unit index, plane, stream position, image of the next plane to request
    auto dsc = WN_DESC((int)blockIdx.x);
#define WN_DMA()                                                                                            \
    {                                                                                                       \
        if (dg < nstream) {                                                                                 \
            const unsigned so_ = (unsigned)((dp + 1) * plane_bytes);                                        \
            unsigned char* dst_ = lds + dbuf * PLANE_LDS;                                                   \
            dbuf = dbuf == NBUF - 1 ? 0 : dbuf + 1;                                                         \
            _Pragma("unroll") for (int k_ = 0; k_ < DPW; ++k_)                                              \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(dsc, (__attribute__((address_space(3))) void*)(dst_ + (wv + 4 * k_) * 1024), \
                                                         16, voff[k_], so_, 0, 0);                          \
            ++dg;                                                                                           \
            if (++dp > a.D) {                                                                               \
                dp = 0;                                                                                     \
                ++dk;                                                                                       \
                if (dk < nmine) dsc = WN_DESC((int)blockIdx.x + dk * G);                                    \
            }                                                                                               \
        }                                                                                                   \
    }
#define WN_READ(BUF_)                                                                                       \
    {                                                                                                       \
        const unsigned char* im_ = lds + (BUF_) * PLANE_LDS;                                                \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                  \
            const int o_ = (((j_ & 1) * HALF + (j_ >> 1)) * PITCH) * 16;                                    \
            raw[0][j_][0] = *reinterpret_cast<const u32x4*>(im_ + rd0 + o_);                                \
            raw[0][j_][1] = *reinterpret_cast<const u32x4*>(im_ + rd0 + o_ + 32);                           \
            raw[1][j_][0] = *reinterpret_cast<const u32x4*>(im_ + rd1 + o_);                                \
            raw[1][j_][1] = *reinterpret_cast<const u32x4*>(im_ + rd1 + o_ + 32);                           \
        }                                                                                                   \
    }
// raw -> V (all four b of this wave's a), split, in the B-operand layout: lane (tile n, kg) holds channels 8 kg .. 8 kg + 7
#define WN_TRANSFORM()                                                                                      \
    _Pragma("unroll") for (int d_ = 0; d_ < 4; ++d_) {                                                      \
        float tl_[4], th_[4];                                                                               \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                  \
            tl_[j_] = mix_sum_lo(raw[0][j_][0][d_], raw[0][j_][1][d_]);                                     \
            tl_[j_] = mix_fma_lo(raw[1][j_][0][d_], sgn, tl_[j_]);                                          \
            tl_[j_] = mix_fma_lo(raw[1][j_][1][d_], sgn, tl_[j_]);                                          \
            th_[j_] = mix_sum_hi(raw[0][j_][0][d_], raw[0][j_][1][d_]);                                     \
            th_[j_] = mix_fma_hi(raw[1][j_][0][d_], sgn, th_[j_]);                                          \
            th_[j_] = mix_fma_hi(raw[1][j_][1][d_], sgn, th_[j_]);                                          \
        }                                                                                                   \
        unsigned h_, l_;                                                                                    \
        split_pair(tl_[0] - tl_[2], th_[0] - th_[2], h_, l_); vh[0][d_] = h_; vl[0][d_] = l_;               \
        split_pair(tl_[1] + tl_[2], th_[1] + th_[2], h_, l_); vh[1][d_] = h_; vl[1][d_] = l_;               \
        split_pair(tl_[2] - tl_[1], th_[2] - th_[1], h_, l_); vh[2][d_] = h_; vl[2][d_] = l_;               \
        split_pair(tl_[1] - tl_[3], th_[1] - th_[3], h_, l_); vh[3][d_] = h_; vl[3][d_] = l_;               \
    }
#define WN_STAMP(K)                                                                                         \
    if constexpr (ABL & 16) {                                                                               \
        unsigned long long t_;                                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
        tsum[K] += t_ - tlast;                                                                              \
        tlast = t_;                                                                                         \
    }

// One plane step.  P = real input plane 0 .. D (D = the zero border: it only finishes the last output plane).  Open output planes:
// P + 1 in slot SI (first contribution: C = 0), P in slot SM, P - 1 in slot SF (last contribution; finished here).
#define WN_STEP(P, SI, SM, SF)                                                                              \
    {                                                                                                       \
        const int p_ = (P);                                                                                 \
        const int o_ = p_ - 1;                          /* the output plane finished in this step (-1: none) */ \
        WN_STAMP(0)                                                                                         \
        /* residual records of the plane finished in this step: staged like the input (every global read of the kernel is an LDS-DMA \
           request waited for by hand: hipcc's vmcnt bookkeeping does not count them, its waits for ordinary loads would drain the stream) */ \
        unsigned char* rim_ = lds + RB + (g & 1) * RES_LDS;                                                 \
        if (a.res) {                                                                                        \
            const unsigned so_ = (unsigned)((o_ + 1) * plane_bytes);                                        \
            _Pragma("unroll") for (int k_ = 0; k_ < RDPW; ++k_)                                             \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rdsc, (__attribute__((address_space(3))) void*)(rim_ + (wv + 4 * k_) * 1024), \
                                                         16, rvoff[k_], so_, 0, 0);                         \
        }                                                                                                   \
        const bool dma_ = dg < nstream;                                                                     \
        if constexpr (!(ABL & 4)) { WN_DMA() }          /* the plane two steps ahead */                    \
        asm volatile("s_nop 1");                                                                            \
        WN_STAMP(1)                                                                                         \
        if constexpr (!(ABL & 1)) {                                                                         \
        WN_GROUP(SF, 2, WN_MF)                                                                              \
        WN_GROUP(SM, 1, WN_MF)                                                                              \
        WN_GROUP(SI, 0, WN_MF0)                                                                             \
        }                                                                                                   \
        WN_PAD()                                                                                            \
        WN_STAMP(2)                                                                                         \
        /* A^T over b inside the wave, then across the waves through LDS */                                 \
        unsigned char* zb_ = lds + ZB + (g & 1) * 16384;   /* alternating over ALL steps of the walk: one barrier per step is enough */ \
        _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) {                                                  \
            const f32x4 z0_ = (Y[SF][0][c_] + Y[SF][1][c_]) + Y[SF][2][c_];                                 \
            const f32x4 z1_ = (Y[SF][1][c_] - Y[SF][2][c_]) - Y[SF][3][c_];                                 \
            *reinterpret_cast<f32x4*>(zb_ + ((wv * 2 + 0) * 2 + c_) * 1024 + lane * 16) = z0_;              \
            *reinterpret_cast<f32x4*>(zb_ + ((wv * 2 + 1) * 2 + c_) * 1024 + lane * 16) = z1_;              \
        }                                                                                                   \
        /* everything older than this step's staging requests has landed: the residual, and the image of the NEXT plane */ \
        if (dma_) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");                               \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                    \
        WN_STAMP(3)                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                       \
        WN_STAMP(4)                                                                                         \
        f32x4 zz_[2][3];                                                                                    \
        u32x2 rh_[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}}, rl_[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};             \
        if (a.res) {                                                                                        \
            _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) {                                              \
                rh_[c_] = *reinterpret_cast<const u32x2*>(rim_ + rrd + c_ * 64);                            \
                rl_[c_] = *reinterpret_cast<const u32x2*>(rim_ + rrd + c_ * 64 + 32);                       \
            }                                                                                               \
        }                                                                                                   \
        _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_)                                                    \
            _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_)                                                \
                zz_[c_][k_] = *reinterpret_cast<const f32x4*>(zb_ + (((pa + k_) * 2 + q) * 2 + c_) * 1024 + lane * 16); \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (g + 1 < nstream) { WN_READ(nbuf) }          /* behind the exchange reads: the epilogue starts on those */ \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) {                                                  \
            f32x4 t_ = zz_[c_][0] + osg * (zz_[c_][1] + zz_[c_][2]);                                        \
            t_ = t_ * esc[c_] + esh[c_];                                                                    \
            if (a.res) {                                                                                    \
                t_[0] = mix_add_lo(rl_[c_][0], mix_add_lo(rh_[c_][0], t_[0]));                              \
                t_[1] = mix_add_hi(rl_[c_][0], mix_add_hi(rh_[c_][0], t_[1]));                              \
                t_[2] = mix_add_lo(rl_[c_][1], mix_add_lo(rh_[c_][1], t_[2]));                              \
                t_[3] = mix_add_hi(rl_[c_][1], mix_add_hi(rh_[c_][1], t_[3]));                              \
            }                                                                                               \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) t_[e_] = __builtin_fmaxf(t_[e_], t_[e_] * a.neg_slope); \
            unsigned h0_, l0_, h1_, l1_;                                                                    \
            split_pair(t_[0], t_[1], h0_, l0_);                                                             \
            split_pair(t_[2], t_[3], h1_, l1_);                                                             \
            if (o_ >= 0) {                                                                                  \
                unsigned char* q_ = yb + (long long)(o_ + 1) * plane_bytes + lane_out + c_ * 64;            \
                *reinterpret_cast<u32x2*>(q_) = u32x2{h0_, h1_};                                            \
                *reinterpret_cast<u32x2*>(q_ + 32) = u32x2{l0_, l1_};                                       \
            }                                                                                               \
        }                                                                                                   \
        WN_STAMP(5)                                                                                         \
        if constexpr (!(ABL & 2)) { WN_TRANSFORM() }                                                        \
        WN_STAMP(6)                                                                                         \
        ++g;                                                                                                \
        nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;                                                             \
    }

    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    if constexpr (ABL & 16) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
    int g = 0;                                         // stream position of the plane being multiplied
    // the first two planes of the stream; V of the first
    WN_DMA()
    WN_DMA()
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    WN_READ(0)
    WN_TRANSFORM()
    int nbuf = 1;                                      // image of stream position g + 1
    for (int k = 0; k < nmine; ++k) {
        const int u = (int)blockIdx.x + k * G;
        const int c = u % a.groups_w;
        const int t = u / a.groups_w;
        const int r = t % a.tiles_h;
        const int b = t / a.tiles_h;
        // this wave's output voxel origin (padded row 2 r + pa + 1, column 32 c + q + 1)
        unsigned char* yb = a.y + b * frame_bytes + ((long long)(2 * r + pa + 1) * Wp + 32 * c + q + 1) * 128;
        const long long roff = b * frame_bytes + ((long long)(2 * r + 1) * Wp + 32 * c + 1) * 128;      // the unit's output rows in the residual tensor
        const long long rleft = total_bytes - roff;
        const auto rdsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.res ? a.res : a.x) + roff, 0,
                                                            rleft > 0x7fffff00ll ? 0x7fffff00 : (int)rleft, 0x00020000);
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)         // slot 0 (output plane 0) must start at zero; the others only must not be undefined
#pragma unroll
            for (int b_ = 0; b_ < 4; ++b_)
#pragma unroll
                for (int c_ = 0; c_ < 2; ++c_) Y[s_][b_][c_] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p0 = 0; p0 <= a.D; p0 += 3) {
            WN_STEP(p0, 1, 0, 2)
            if (p0 + 1 <= a.D) WN_STEP(p0 + 1, 2, 1, 0)
            if (p0 + 2 <= a.D) WN_STEP(p0 + 2, 0, 2, 1)
        }
    }
    if constexpr (ABL & 16) {
        if (blockIdx.x == 0 && lane == 0 && a.dbg)
            for (int k = 0; k < 8; ++k) a.dbg[wv * 8 + k] = tsum[k];
    }
}

template <int ABL>
int wino_launch(const WinoArgs& a, long long units, hipStream_t st) {
    static mvsgi::PersistentGeom geo_cache[mvsgi::kMaxDevices] = {};
    mvsgi::PersistentGeom geo;
    if (mvsgi::persistent_geometry(conv3d_wino32_kernel<ABL>, 256, wn::LDS_BYTES, 1, geo_cache, "conv3d(winograd)", geo)) return 1;
    const unsigned grid = (unsigned)(units < geo.cus ? units : geo.cus);
    hipLaunchKernelGGL(conv3d_wino32_kernel<ABL>, dim3(grid), dim3(256), wn::LDS_BYTES, st, a);
    return 0;
}

}  // namespace

extern "C" {

size_t mvsgi_conv3d_wino32_packed_weight_bytes(void) { return (size_t)4 * 4 * 3 * 2 * 2 * 64 * 16; }

int mvsgi_conv3d_wino32_f16(const void* x_split, const void* w_packed, const float* scale, const float* shift, const void* res_split,
                            void* y_split, int B, int D, int H, int W, float neg_slope, void* stream) {
    MVSGI_REQUIRE(x_split && w_packed && scale && shift && y_split, "mvsgi_conv3d_wino32_f16: null pointer");
    MVSGI_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "mvsgi_conv3d_wino32_f16: bad dims");
    MVSGI_REQUIRE(H % 2 == 0 && W % 32 == 0, "mvsgi_conv3d_wino32_f16: needs H %% 2 == 0 and W %% 32 == 0 (got %d, %d)", H, W);
    MVSGI_REQUIRE(neg_slope >= 0.f && neg_slope <= 1.f, "mvsgi_conv3d_wino32_f16: neg_slope outside [0, 1]");
    MVSGI_REQUIRE((long long)B * (D + 2) * (H + 2) * (W + 2) * 128 < (1ll << 40), "mvsgi_conv3d_wino32_f16: tensor too large");
    MVSGI_REQUIRE((long long)(D + 2) * (H + 2) * (W + 2) * 128 < (1ll << 31), "mvsgi_conv3d_wino32_f16: frame too large for 32-bit plane offsets");
    WinoArgs a;
    memset(&a, 0, sizeof(a));
    a.x = reinterpret_cast<const unsigned char*>(x_split);
    a.y = reinterpret_cast<unsigned char*>(y_split);
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
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const char* e_ = getenv("MVSGI_WINO_ABL");
    const int abl = e_ ? atoi(e_) : 0;
    int rc = 0;
    switch (abl) {
        case 16: {
            static unsigned long long* dbg = nullptr;
            if (!dbg) (void)hipMalloc(&dbg, 32 * 8);
            a.dbg = dbg;
            rc = wino_launch<16>(a, units, st);
            (void)hipDeviceSynchronize();
            unsigned long long h[32];
            (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
            for (int w = 0; w < 4; ++w) {
                fprintf(stderr, "wave %d:", w);
                for (int k = 0; k < 7; ++k) fprintf(stderr, " %llu", h[w * 8 + k]);
                fprintf(stderr, "\n");
            }
            break;
        }
        case 1: rc = wino_launch<1>(a, units, st); break;
        case 2: rc = wino_launch<2>(a, units, st); break;
        case 4: rc = wino_launch<4>(a, units, st); break;
        case 6: rc = wino_launch<6>(a, units, st); break;
        case 7: rc = wino_launch<7>(a, units, st); break;
        default: rc = wino_launch<0>(a, units, st);
    }
    if (rc) return rc;
    return mvsgi::check_launch("mvsgi_conv3d_wino32_f16");
}

}  // extern "C"
