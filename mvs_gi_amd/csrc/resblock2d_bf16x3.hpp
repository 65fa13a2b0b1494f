// Fused residual block of the feature extractor: ResConvBlk2d.forward
// (dsta_mvs/model/common/common_modules.py:165-176) for 16 -> 16 channels, 3x3, stride 1,
//   r = LReLU(BN1(conv1(x)));  y = LReLU(BN2(conv2(r)) + x)
// in ONE launch, with r never leaving the CU.  Included by conv2d.hip (needs conv3d_bf16x3.hpp).
//
// Why: the 16 -> 16 layers of the extractor are HBM-bound (24 flop/B): as two launches a block moves
// x -> r -> y plus the residual, 5.5 x 64 B per pixel; fused it reads x once (18x18 halo per 14x14
// outputs) and writes y, 2.65 x 64 B per pixel (+ the residual from L2).
//
// A workgroup = 4 consumer waves + 4 producer waves, persistent over bricks of 14 x 30 output pixels:
//   producers  stage the 18 x 34 x 16-channel input window of the NEXT brick (fp32 -> bf16 hi / lo, the
//              80-byte-per-pixel LDS image of the 3-D kernels), two bricks ahead in registers, item by item;
//   consumers  phase A: conv1 on the 16 x 32 region around the brick (the outputs + 1 ring), split-bf16 MFMA
//              (5 tap pairs x 3 products), BN1 + LeakyReLU in registers, ZERO outside the image (it is conv2's
//              padding), split again and written to a second LDS image;                      -- barrier --
//              phase B: conv2 on the 14 x 30 brick from that image, BN2 + residual (x, from L2, requested at the
//              start of the brick) + LeakyReLU, 16-byte stores;                                -- barrier --
//              (a variant that skews the two convolutions by one brick -- one barrier per brick, the epilogue of one
//              under the MFMAs of the other -- measured the same; per-brick fixed costs are what a bigger brick buys.)
// Both weight sets (2 x 5 pairs x hi/lo = 20 fragments = 80 VGPRs) stay in registers for the whole launch:
// no weight traffic at all.  conv1 is evaluated on 512 pixels and conv2's tiles carry 420 live outputs of 512,
// so the kernel issues 2.4 MFMA-pixels per output instead of 2 -- the price of the halved traffic.
#pragma once

struct ResBlk2dArgs {
    const float* x;
    const f32x4* wp1;     // pack_weights_bf16x3_kernel(taps = 9): [5 pairs][hi|lo][64 lanes][8 bf16]
    const f32x4* wp2;
    const float* scale1;
    const float* shift1;
    const float* scale2;
    const float* shift2;
    float* y;
    int N, H, W;          // images, size; C == 16
    int tiles_h, tiles_w, total_units;
    float neg_slope;
};

__global__ __launch_bounds__(512, 2) void resblock2d_bf16x3_kernel(ResBlk2dArgs a) {
    constexpr int TOH = 14, TOW = 30;           // output brick
    constexpr int RH = TOH + 2, RW = TOW + 2;   // conv1 region 16 x 32 = 32 MFMA tiles of 16 pixels
    constexpr int IH = TOH + 4, IW = TOW + 4;   // input window 18 x 34
    constexpr int IVA = IH * IW;                // 612 staged pixels
    constexpr int NIT = (IVA * 4 + 255) / 256;  // 10 staging items per producer thread
    constexpr int BUFA = IVA * kVSB;            // 48960 B
    constexpr int RBW = RW + 2;                 // row pitch (pixels) of the conv1-result image: + 2 columns read by masked lanes
    constexpr int BUFB = (RH + 2) * RBW * kVSB; // 48960 B: 16 rows written + 2 rows only the unstored tile rows 14, 15 read
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];   // [imgA 0][imgA 1][imgB]
    unsigned char* imgB = ldsb + 2 * BUFA;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = a.total_units, G = gridDim.x;
    const int nmine = (total - (int)blockIdx.x + G - 1) / G;
    const long long img_elems = (long long)a.H * a.W * 16;

#define MVSGI_RB_DECODE(ID, N_, OH, OW)                          \
    {                                                            \
        int t_ = xcd_remap((ID), total);                         \
        OW = (t_ % a.tiles_w) * TOW;                             \
        t_ /= a.tiles_w;                                         \
        OH = (t_ % a.tiles_h) * TOH;                             \
        N_ = t_ / a.tiles_h;                                     \
    }

    if (wave >= 4) {
        // ============================ producers ============================
        const int ptid = tid - 256;
        int goff[NIT];
        unsigned okmask = 0, okA = 0, okB = 0;
        const float* xb = a.x;
        f32x4 preA[NIT], preB[NIT];
        int k2 = 0;                               // next brick (ordinal) to request
#define MVSGI_RB_PLAN(UNIT)                                                                             \
        {                                                                                               \
            int n_, oh_, ow_;                                                                           \
            MVSGI_RB_DECODE(UNIT, n_, oh_, ow_)                                                         \
            okmask = 0;                                                                                 \
            _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                        \
                const int e = ptid + it * 256;                                                          \
                const int v = e >> 2, q = e & 3;                                                        \
                const int gh = oh_ - 2 + v / IW, gw = ow_ - 2 + v % IW;                                 \
                const bool ok = e < IVA * 4 && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;              \
                goff[it] = ok ? (gh * a.W + gw) * 16 + q * 4 : 0;                                       \
                okmask |= ok ? (1u << it) : 0u;                                                         \
            }                                                                                           \
            xb = a.x + (long long)n_ * img_elems;                                                       \
        }
#define MVSGI_RB_ISSUE1(PRE, IT) PRE[IT] = *reinterpret_cast<const f32x4*>(xb + goff[IT]);
#define MVSGI_RB_PUT1(PRE, OK, DST, IT)                                                                 \
        {                                                                                               \
            const int e = ptid + (IT) * 256;                                                            \
            if (e < IVA * 4) {                                                                          \
                u32x2 hi, lo;                                                                           \
                split_bf16x4(PRE[IT], hi, lo);                                                          \
                if (!(((OK) >> (IT)) & 1u)) hi = lo = u32x2{0u, 0u};                                    \
                *reinterpret_cast<u32x2*>((DST) + (e >> 2) * kVSB + (e & 3) * 8) = hi;                  \
                *reinterpret_cast<u32x2*>((DST) + (e >> 2) * kVSB + 32 + (e & 3) * 8) = lo;             \
            }                                                                                           \
        }
        // request brick k2 into NEW (unconditionally: past the end the last brick again, see conv3d_bf16x3.hpp) while
        // the brick in OLD is split and written, item by item
#define MVSGI_RB_STEP(NEW, OKNEW, OLD, OKOLD, DST, DOPUT)                                               \
        {                                                                                               \
            MVSGI_RB_PLAN((int)blockIdx.x + k2 * G)                                                     \
            _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                        \
                MVSGI_RB_ISSUE1(NEW, it)                                                                \
                if (DOPUT) MVSGI_RB_PUT1(OLD, OKOLD, DST, it)                                           \
                __builtin_amdgcn_sched_barrier(0);                                                      \
            }                                                                                           \
            OKNEW = okmask;                                                                             \
            if (++k2 >= nmine) k2 = nmine - 1;                                                          \
        }
        MVSGI_RB_STEP(preA, okA, preA, okA, ldsb, false)          // brick 0 requested
#pragma unroll
        for (int it = 0; it < NIT; ++it) MVSGI_RB_PUT1(preA, okA, ldsb, it)
        MVSGI_RB_STEP(preA, okA, preA, okA, ldsb, false)          // brick 1 in flight
        __syncthreads();                                           // image 0 holds brick 0
        for (int u = 0; u < nmine; u += 2) {
            MVSGI_RB_STEP(preB, okB, preA, okA, ldsb + ((u + 1) & 1) * BUFA, u + 1 < nmine)
            __syncthreads();                                       // conv1 result of brick u written
            __syncthreads();                                       // brick u done, image of brick u+1 complete
            if (u + 1 < nmine) {
                MVSGI_RB_STEP(preA, okA, preB, okB, ldsb + ((u + 2) & 1) * BUFA, u + 2 < nmine)
                __syncthreads();
                __syncthreads();
            }
        }
#undef MVSGI_RB_PLAN
#undef MVSGI_RB_ISSUE1
#undef MVSGI_RB_PUT1
#undef MVSGI_RB_STEP
    } else {
        // ============================ consumers ============================
        const int col = lane & 15, kg = lane >> 4;
        const bool second = kg & 1;
        // both weight sets, resident
        bf16x8 w1h[5], w1l[5], w2h[5], w2l[5];
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const char* q1 = reinterpret_cast<const char*>(a.wp1) + p * 2048 + lane * 16;
            const char* q2 = reinterpret_cast<const char*>(a.wp2) + p * 2048 + lane * 16;
            w1h[p] = *reinterpret_cast<const bf16x8*>(q1);
            w1l[p] = *reinterpret_cast<const bf16x8*>(q1 + 1024);
            w2h[p] = *reinterpret_cast<const bf16x8*>(q2);
            w2l[p] = *reinterpret_cast<const bf16x8*>(q2 + 1024);
        }
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(a.scale1 + kg * 4), b1 = *reinterpret_cast<const f32x4*>(a.shift1 + kg * 4);
        const f32x4 s2 = *reinterpret_cast<const f32x4*>(a.scale2 + kg * 4), b2 = *reinterpret_cast<const f32x4*>(a.shift2 + kg * 4);
        // tile (i, hf) of this wave = region row 4*wave + i, columns 16*hf + col; fragment byte offsets (hi part) of half 0
        int baseA[4], baseB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * wave + i;
            baseA[i] = (r * IW + col) * kVSB + (kg >> 1) * 16;       // conv1: region (r, c) reads window rows r..r+2, cols c..c+2
            baseB[i] = (r * RBW + col) * kVSB + (kg >> 1) * 16;      // conv2: output (r, c) reads region rows r..r+2
        }
#define MVSGI_RB_READ(IMG, PITCH, BASE, HF, P, BUFI)                                                    \
            {                                                                                           \
                const int t0_ = 2 * (P), t1_ = 2 * (P) + 1 < 9 ? 2 * (P) + 1 : 2 * (P);                 \
                const int o0_ = ((t0_ / 3) * (PITCH) + t0_ % 3) * kVSB, o1_ = ((t1_ / 3) * (PITCH) + t1_ % 3) * kVSB; \
                const int off_ = (second ? o1_ : o0_) + (HF) * 16 * kVSB;                               \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                         \
                    xh[BUFI][i] = *reinterpret_cast<const bf16x8*>((IMG) + BASE[i] + off_);             \
                    xl[BUFI][i] = *reinterpret_cast<const bf16x8*>((IMG) + BASE[i] + off_ + 32);        \
                }                                                                                       \
            }
// one convolution = 10 steps (2 column halves x 5 tap pairs) of 4 tiles x 3 products; three fragment buffers,
// requests two steps ahead (a step is only 12 MFMAs = 192 cycles, less than an LDS round trip)
#define MVSGI_RB_CONV(IMG, PITCH, BASE, WH, WL, ACC)                                                    \
        {                                                                                               \
            bf16x8 xh[3][4], xl[3][4];                                                                  \
            MVSGI_RB_READ(IMG, PITCH, BASE, 0, 0, 0)                                                    \
            MVSGI_RB_READ(IMG, PITCH, BASE, 0, 1, 1)                                                    \
            _Pragma("unroll") for (int st = 0; st < 10; ++st) {                                         \
                const int hf = st / 5, p = st % 5;                                                      \
                if (st + 2 < 10) MVSGI_RB_READ(IMG, PITCH, BASE, (st + 2) / 5, (st + 2) % 5, (st + 2) % 3) \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
                    ACC[hf][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WL[p], xh[st % 3][i], ACC[hf][i], 0, 0, 0); \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
                    ACC[hf][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WH[p], xl[st % 3][i], ACC[hf][i], 0, 0, 0); \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
                    ACC[hf][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WH[p], xh[st % 3][i], ACC[hf][i], 0, 0, 0); \
            }                                                                                           \
        }
        f32x4 acc[2][4];
        __syncthreads();                                           // image 0 holds brick 0
        for (int u = 0; u < nmine; ++u) {
            int n_, oh0, ow0;
            MVSGI_RB_DECODE((int)blockIdx.x + u * G, n_, oh0, ow0)
            const unsigned char* imgA = ldsb + (u & 1) * BUFA;
            // residual pixels of this lane's eight output tiles: requested now, used after both convolutions
            const float* xres = a.x + (long long)n_ * img_elems;
            f32x4 rx[2][4];
            int oo[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * wave + i, c = 16 * hf + col, oh = oh0 + r, ow = ow0 + c;
                    oo[hf][i] = (r < TOH && c < TOW && oh < a.H && ow < a.W) ? (oh * a.W + ow) * 16 + kg * 4 : -1;
                    rx[hf][i] = *reinterpret_cast<const f32x4*>(xres + (oo[hf][i] >= 0 ? oo[hf][i] : 0));
                }
            // ---- phase A: conv1 on the 16 x 32 region, result -> imgB (zero outside the image) ----
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[hf][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            MVSGI_RB_CONV(imgA, IW, baseA, w1h, w1l, acc)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * wave + i, c = 16 * hf + col;
                    const int gh = oh0 - 1 + r, gw = ow0 - 1 + c;
                    f32x4 v = acc[hf][i] * s1 + b1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.neg_slope;
                    u32x2 hi, lo;
                    split_bf16x4(v, hi, lo);
                    if (!(gh >= 0 && gh < a.H && gw >= 0 && gw < a.W)) hi = lo = u32x2{0u, 0u};
                    // lane (col, kg): channels 4*kg .. 4*kg+3 of region pixel (r, c)
                    *reinterpret_cast<u32x2*>(imgB + (r * RBW + c) * kVSB + kg * 8) = hi;
                    *reinterpret_cast<u32x2*>(imgB + (r * RBW + c) * kVSB + 32 + kg * 8) = lo;
                }
            __syncthreads();                                       // conv1 result complete
            // ---- phase B: conv2 on the 14 x 30 brick (tile rows 14, 15 and columns 30, 31 are not stored) ----
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[hf][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            MVSGI_RB_CONV(imgB, RBW, baseB, w2h, w2l, acc)
            float* yb = a.y + (long long)n_ * img_elems;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (oo[hf][i] >= 0) {
                        f32x4 v = acc[hf][i] * s2 + b2 + rx[hf][i];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.neg_slope;
                        *reinterpret_cast<f32x4*>(yb + oo[hf][i]) = v;
                    }
                }
            __syncthreads();                                       // imgB free, image of brick u+1 complete
        }
#undef MVSGI_RB_CONV
#undef MVSGI_RB_READ
    }
#undef MVSGI_RB_DECODE
}
