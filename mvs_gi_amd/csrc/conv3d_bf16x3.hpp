// Split-bf16 ("bf16x3") implicit-GEMM 3x3x3 convolution on v_mfma_f32_16x16x32_bf16.
// Included by conv3d.hip (shares ConvArgs / f32x4 / kVS with the exact-fp32 kernels).
//
// Arithmetic: every fp32 operand is split as x = hi + lo, hi = bf16(x), lo = bf16(x - hi)
// (16 significant bits); a product is hi*hi + hi*lo + lo*hi with fp32 accumulation: 3 bf16
// MFMAs at 16x the fp32-MFMA rate = 5.3x the exact kernel, ~2^-16 relative error per product.
//
// Mapping (D[cout 16][voxel 16] += W[cout 16][k 32] * X[k 32][voxel 16]):
//   * one MFMA's K = 32 is a PAIR of taps x 16 channels: lane = (kg << 4) | col,
//     kg & 1 selects the tap of the pair, kg >> 1 the 8-channel half.  27 taps -> 14 pairs
//     (the last half empty: zero weights, 3.6 % padding work);
//   * a 16-voxel tile is 16 consecutive W positions of the brick (TW == 16), so the 16 lanes a
//     ds_read_b128 services together (8 lanes of one kg, 8 of its tap-partner) address 15
//     different consecutive voxels + 1 shared one at an 80-byte stride: bank-conflict free for the
//     in-row pairs (delta 1) and the row-wrap pairs (delta ITW - 2 = 16), see DESIGN.md;
//   * activations stay fp32 in HBM; the halo brick of one 16-channel slice is fetched into
//     registers one slice AHEAD (loads in flight under the MFMA loop), split on the way into LDS
//     as [hi0-7 | hi8-15 | lo0-7 | lo8-15 | pad] = 80 B per voxel;
//   * weights arrive pre-split and lane-ordered straight from L2 (one coalesced 16-byte load per
//     lane per fragment); every wave of every workgroup reads the same few hundred KB;
//   * the flat grid is re-mapped so that each XCD (= each private L2) owns a contiguous run of
//     bricks: neighbouring bricks' halos and the cout-blocks of one brick hit the same L2.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kVSB = 80;      // LDS bytes per staged voxel
constexpr int kPairs = 14;

// [Cout][Cin][27] -> [Cin/16][14 pairs][Cout/16][hi|lo][64 lanes][8 bf16]
//   lane = (kg << 4) | i holds W[cout = ct*16+i][cin = cc*16 + (kg>>1)*8 + j][tap = 2p + (kg&1)]
__global__ void pack_weights_bf16x3_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cout, int Cin) {
    const int CT = Cout / 16;
    const long long total = (long long)(Cin / 16) * kPairs * CT * 64;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int ct = (int)(r % CT);
    r /= CT;
    const int p = (int)(r % kPairs);
    const int cc = (int)(r / kPairs);
    const int kg = lane >> 4;
    const int co = ct * 16 + (lane & 15);
    const int ci = cc * 16 + (kg >> 1) * 8;
    const int tap = 2 * p + (kg & 1);
    bf16x8 hi, lo;
    for (int j = 0; j < 8; ++j) {
        const float v = tap < 27 ? w[((long long)co * Cin + ci + j) * 27 + tap] : 0.f;
        const __bf16 h = (__bf16)v;
        hi[j] = h;
        lo[j] = (__bf16)(v - (float)h);
    }
    const long long o = ((((long long)cc * kPairs + p) * CT + ct) * 2) * 64 + lane;
    wp[o] = hi;
    wp[o + 64] = lo;
}

// bijective XCD-aware remap of a flat block id (cdna_hip_programming.md T1): blocks b, b+8, ...
// share an XCD; give each XCD a contiguous run of the logical index space.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int S>
__global__ __launch_bounds__(256) void conv3d_bf16x3_kernel(ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(WM * MW * 16 == TD * TH * TW, "brick must be covered by the voxel tiles");
    constexpr int ITD = (TD - 1) * S + 3, ITH = (TH - 1) * S + 3, ITW = (TW - 1) * S + 3;
    constexpr int IV = ITD * ITH * ITW;
    constexpr int NIT = (IV * 4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int col = lane & 15, kg = lane >> 4;
    const bool second = kg & 1;       // this lane's k-range belongs to the pair's second tap

    // flat grid -> (brick, cout block), cout block fastest, XCD-contiguous
    const int CT = a.Cout / 16;
    const int ny = (CT + WN * NW - 1) / (WN * NW);
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = t % ny;
    t /= ny;
    const int tw_i = t % a.tiles_w;
    t /= a.tiles_w;
    const int th_i = t % a.tiles_h;
    t /= a.tiles_h;
    const int td_i = t % a.tiles_d;
    const int b = t / a.tiles_d;
    const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
    const int id0 = od0 * S - 1, ih0 = oh0 * S - 1, iw0 = ow0 * S - 1;
    const int ct0 = (cb * WN + wn) * NW;

    // per-lane LDS byte address of each voxel tile's B fragment (hi part, tap offset excluded)
    int base[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int v = (wm * MW + i) * 16 + col;
        const int w_ = v % TW, h_ = (v / TW) % TH, d_ = v / (TW * TH);
        base[i] = (((d_ * S) * ITH + h_ * S) * ITW + w_ * S) * kVSB + (kg >> 1) * 16;
    }

    // staging plan: item e = tid + it*256 -> (halo voxel e>>2, channel quad e&3); element offset
    // of the item within one (batch, slice-0) volume.  Padding / surplus items load a valid
    // dummy address and are zeroed by a select (no per-item branch: a branch per load makes
    // hipcc wait for each one in turn).
    int goff[NIT];
    unsigned okmask = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * 256;
        const int v = e >> 2, q = e & 3;
        const int iw = v % ITW, ih = (v / ITW) % ITH, id = v / (ITW * ITH);
        const int gd = id0 + id, gh = ih0 + ih, gw = iw0 + iw;
        const bool ok = (e < IV * 4) && gd >= 0 && gd < a.Din && gh >= 0 && gh < a.Hin && gw >= 0 && gw < a.Win;
        goff[it] = ok ? ((gd * a.Hin + gh) * a.Win + gw) * a.Cin + q * 4 : 0;
        okmask |= ok ? (1u << it) : 0u;
    }
    const float* xb_base = a.x + (long long)b * a.Din * a.Hin * a.Win * a.Cin;
    f32x4 pre[NIT];
    static_assert(NIT <= kPairs - 1, "the slice prefetch is spread over the pair loop");
#define MVSGI_PRELOAD(IT, CC) pre[IT] = *reinterpret_cast<const f32x4*>(xb_base + goff[IT] + (CC) * 16);

    f32x4 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.Cin / 16;
    const bf16x8* wpb = reinterpret_cast<const bf16x8*>(a.wp) + lane;
    int ctc[NW];                                   // clamped cout tile (surplus tiles are never stored)
#pragma unroll
    for (int j = 0; j < NW; ++j) ctc[j] = ct0 + j < CT ? ct0 + j : CT - 1;

    // software pipeline: the weight fragments of pair p+1 are requested before pair p is multiplied
    // (double-buffered registers); the activation fragments of pair p+1 are read from LDS half a
    // pair ahead, into the registers the first / second half of pair p's MFMAs have just consumed
    // (small tiles: fully double-buffered).  All indices are static after unrolling.
    constexpr int XB = MW <= 2 ? 2 : 1;
    constexpr int MH = XB == 2 ? MW : MW / 2;      // voxel tiles per half
    bf16x8 wh[2][NW], wl[2][NW], xh[XB][MW], xl[XB][MW];
#define MVSGI_LOADW(BUF, CC, P)                                                                   \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                              \
        const bf16x8* q_ = wpb + ((long long)((CC) * kPairs + (P)) * CT + ctc[j]) * 128;          \
        wh[BUF][j] = q_[0];                                                                       \
        wl[BUF][j] = q_[64];                                                                      \
    }
#define MVSGI_READX(BUF, P, I0, I1)                                                               \
    {                                                                                             \
        const int p_ = (P);                                                                       \
        const int t0_ = 2 * p_, t1_ = (2 * p_ + 1 < 27) ? 2 * p_ + 1 : 2 * p_;                    \
        const int o0_ = (((t0_ / 9) * ITH + (t0_ / 3) % 3) * ITW + t0_ % 3) * kVSB;               \
        const int o1_ = (((t1_ / 9) * ITH + (t1_ / 3) % 3) * ITW + t1_ % 3) * kVSB;               \
        const int off_ = second ? o1_ : o0_;                                                      \
        _Pragma("unroll") for (int i = (I0); i < (I1); ++i) {                                     \
            xh[BUF][i] = *reinterpret_cast<const bf16x8*>(ldsb + base[i] + off_);                 \
            xl[BUF][i] = *reinterpret_cast<const bf16x8*>(ldsb + base[i] + off_ + 32);            \
        }                                                                                         \
    }
#define MVSGI_MFMAS(WB, XBUF, I0, I1)                                                             \
    _Pragma("unroll") for (int i = (I0); i < (I1); ++i)                                           \
        _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                          \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[WB][j], xh[XBUF][i], acc[i][j], 0, 0, 0); \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[WB][j], xl[XBUF][i], acc[i][j], 0, 0, 0); \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[WB][j], xh[XBUF][i], acc[i][j], 0, 0, 0); \
        }

#pragma unroll
    for (int it = 0; it < NIT; ++it) { MVSGI_PRELOAD(it, 0) }
    MVSGI_LOADW(0, 0, 0)
    for (int cc = 0; cc < nchunks; ++cc) {
        __syncthreads();                       // every wave is done reading the previous slice
#pragma unroll
        for (int it = 0; it < NIT; ++it) {     // split fp32 -> (hi, lo) bf16 on the way into LDS
            const int e = tid + it * 256;
            if (e < IV * 4) {
                const int v = e >> 2, q = e & 3;
                const bool ok = (okmask >> it) & 1u;
                bf16x4 hi, lo;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xv = ok ? pre[it][k] : 0.f;
                    const __bf16 h = (__bf16)xv;
                    hi[k] = h;
                    lo[k] = (__bf16)(xv - (float)h);
                }
                *reinterpret_cast<bf16x4*>(ldsb + v * kVSB + q * 8) = hi;
                *reinterpret_cast<bf16x4*>(ldsb + v * kVSB + 32 + q * 8) = lo;
            }
        }
        __syncthreads();
        const bool more = cc + 1 < nchunks;
        MVSGI_READX(0, 0, 0, MW)
#pragma unroll
        for (int p = 0; p < kPairs; ++p) {
            const int cur = p & 1, nxt = cur ^ 1;
            const int xcur = XB == 2 ? cur : 0, xnxt = XB == 2 ? nxt : 0;
            if (p + 1 < kPairs) {
                MVSGI_LOADW(nxt, cc, p + 1)
            } else if (more) {
                MVSGI_LOADW(nxt, cc + 1, 0)
            }
            // one item of the next slice's halo brick per pair: global loads complete in order, so
            // a burst here would sit in front of every weight fragment of the following pairs
            if (more && p >= 1 && p - 1 < NIT) { MVSGI_PRELOAD(p - 1, cc + 1) }
            if (XB == 2) {
                if (p + 1 < kPairs) MVSGI_READX(xnxt, p + 1, 0, MW)
                __builtin_amdgcn_sched_barrier(0);
                MVSGI_MFMAS(cur, xcur, 0, MW)
            } else {
                __builtin_amdgcn_sched_barrier(0);
                MVSGI_MFMAS(cur, 0, 0, MH)
                __builtin_amdgcn_sched_barrier(0);
                if (p + 1 < kPairs) MVSGI_READX(0, p + 1, 0, MH)
                __builtin_amdgcn_sched_barrier(0);
                MVSGI_MFMAS(cur, 0, MH, MW)
                __builtin_amdgcn_sched_barrier(0);
                if (p + 1 < kPairs) MVSGI_READX(0, p + 1, MH, MW)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef MVSGI_MFMAS
#undef MVSGI_PRELOAD
#undef MVSGI_LOADW
#undef MVSGI_READX

    // epilogue: lane (col, kg) of tile (i, j) holds couts ct*16 + 4*kg + 0..3 of voxel i*16 + col
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int v = (wm * MW + i) * 16 + col;
        const int w_ = v % TW, h_ = (v / TW) % TH, d_ = v / (TW * TH);
        const int od = od0 + d_, oh = oh0 + h_, ow = ow0 + w_;
        if (od >= a.Do || oh >= a.Ho || ow >= a.Wo) continue;
        const long long vox = (((long long)b * a.Do + od) * a.Ho + oh) * a.Wo + ow;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int ct = ct0 + j;
            if (ct >= CT) continue;
            const int co = ct * 16 + kg * 4;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + co);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + co);
            f32x4 r = acc[i][j] * sc + sh;
            if (a.res) r += *reinterpret_cast<const f32x4*>(a.res + vox * a.Cout + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : r[e] * a.neg_slope;
            *reinterpret_cast<f32x4*>(a.y + vox * a.Cout + co) = r;
        }
    }
}

template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int S>
int launch_bf16x3(ConvArgs a, hipStream_t st) {
    constexpr int ITD = (TD - 1) * S + 3, ITH = (TH - 1) * S + 3, ITW = (TW - 1) * S + 3;
    constexpr size_t lds_bytes = (size_t)ITD * ITH * ITW * kVSB;
    static_assert(lds_bytes <= 160 * 1024, "LDS tile too large");
    auto kern = conv3d_bf16x3_kernel<NW, MW, WM, WN, TD, TH, TW, S>;
    static bool attr_done = false;   // benign race: idempotent
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return mvsgi::fail("conv3d: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    a.tiles_d = (int)mvsgi::cdiv(a.Do, TD);
    a.tiles_h = (int)mvsgi::cdiv(a.Ho, TH);
    a.tiles_w = (int)mvsgi::cdiv(a.Wo, TW);
    const int CT = a.Cout / 16;
    const long long nb = (long long)a.B * a.tiles_d * a.tiles_h * a.tiles_w * mvsgi::cdiv(CT, WN * NW);
    MVSGI_REQUIRE(nb < (1ll << 31), "conv3d: too many workgroups");
    MVSGI_REQUIRE((long long)a.Din * a.Hin * a.Win * a.Cin < (1ll << 31), "conv3d: volume too large for 32-bit offsets");
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), lds_bytes, st, a);
    return mvsgi::check_launch("mvsgi_conv3d_f32(bf16x3)");
}
