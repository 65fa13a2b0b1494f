// Exact-fp32 implicit-GEMM 3x3(x3) convolution on v_mfma_f32_16x16x4_f32 (see conv3d.hip for the
// design notes); shared by conv3d.hip (KD = 3) and conv2d.hip (KD = 1).
#pragma once

// ----------------------------------------------------------------------------------------
// weight packing: [Cout][Cin][taps] -> [Cin/16][taps][Cout/16][64 lanes][4]   (taps = 27, or 9 for 2-D)
//   lane = (kg << 4) | i  holds  W[cout = ct*16 + i][cin = cc*16 + 4*kg + e][tap], e = 0..3
// ----------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, f32x4* __restrict__ wp, int Cout, int Cin, int taps) {
    const int CT = Cout / 16;
    const long long total = (long long)(Cin / 16) * taps * CT * 64;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int ct = (int)(r % CT);
    r /= CT;
    const int tap = (int)(r % taps);
    const int cc = (int)(r / taps);
    const int co = ct * 16 + (lane & 15);
    const int ci = cc * 16 + 4 * (lane >> 4);
    f32x4 v;
    for (int e = 0; e < 4; ++e) v[e] = w[((long long)co * Cin + ci + e) * taps + tap];
    wp[idx] = v;
}

// stage the halo brick of one 16-channel slice into LDS (zero outside the volume = conv padding)
template <int ITD, int ITH, int ITW>
__device__ __forceinline__ void stage_slice(float* lds, const float* __restrict__ xb_base, int Cin, int c0,
                                            int id0, int ih0, int iw0, int Din, int Hin, int Win, int tid) {
    constexpr int IV = ITD * ITH * ITW;
    for (int e = tid; e < IV * 4; e += 256) {
        const int v = e >> 2, q = e & 3;
        const int iw = v % ITW, ih = (v / ITW) % ITH, id = v / (ITW * ITH);
        const int gd = id0 + id, gh = ih0 + ih, gw = iw0 + iw;
        f32x4 val = f32x4{0.f, 0.f, 0.f, 0.f};
        if (gd >= 0 && gd < Din && gh >= 0 && gh < Hin && gw >= 0 && gw < Win)
            val = *reinterpret_cast<const f32x4*>(xb_base + (((long long)gd * Hin + gh) * Win + gw) * Cin + c0 + q * 4);
        *reinterpret_cast<f32x4*>(&lds[v * kVS + q * 4]) = val;
    }
}

// ----------------------------------------------------------------------------------------
// MFMA implicit GEMM
//   NW x MW : 16-cout x 16-voxel accumulator tiles per wave
//   WM x WN : the 4 waves of the workgroup along voxels x couts
//   TD,TH,TW: output brick (TD*TH*TW == WM*MW*16);  S: stride
// ----------------------------------------------------------------------------------------
//   KD = 3: 3x3x3 on a volume; KD = 1: 3x3 on images (D = image index, never strided or padded)
template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int S, int KD = 3>
__global__ __launch_bounds__(256) void conv3d_mfma_kernel(ConvArgs a) {
    constexpr int kTaps = KD * 9, SD = KD == 1 ? 1 : S;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(WM * MW * 16 == TD * TH * TW, "brick must be covered by the voxel tiles");
    constexpr int ITD = (TD - 1) * SD + KD, ITH = (TH - 1) * S + 3, ITW = (TW - 1) * S + 3;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int col = lane & 15, kg = lane >> 4;

    int t = blockIdx.x;
    const int tw_i = t % a.tiles_w;
    t /= a.tiles_w;
    const int th_i = t % a.tiles_h;
    t /= a.tiles_h;
    const int td_i = t % a.tiles_d;
    const int b = t / a.tiles_d;
    const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
    const int id0 = od0 * SD - KD / 2, ih0 = oh0 * S - 1, iw0 = ow0 * S - 1;   // padding = 1 (none across images)

    const int CT = a.Cout / 16;
    const int ct0 = (blockIdx.y * WN + wn) * NW;

    int base[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int v = (wm * MW + i) * 16 + col;
        const int w_ = v % TW, h_ = (v / TW) % TH, d_ = v / (TW * TH);
        base[i] = (((d_ * SD) * ITH + h_ * S) * ITW + w_ * S) * kVS + kg * 4;
    }

    f32x4 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.Cin / 16;
    const float* xb_base = a.x + (long long)b * a.Din * a.Hin * a.Win * a.Cin;
    for (int cc = 0; cc < nchunks; ++cc) {
        __syncthreads();
        stage_slice<ITD, ITH, ITW>(lds, xb_base, a.Cin, cc * 16, id0, ih0, iw0, a.Din, a.Hin, a.Win, tid);
        __syncthreads();
        const f32x4* wp = a.wp + (long long)cc * kTaps * CT * 64 + lane;
#pragma unroll
        for (int tap = 0; tap < kTaps; ++tap) {
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const int off = ((kd * ITH + kh) * ITW + kw) * kVS;
            f32x4 wa[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int ct = ct0 + j;
                wa[j] = (ct < CT) ? wp[(long long)(tap * CT + ct) * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            f32x4 xv[MW];
#pragma unroll
            for (int i = 0; i < MW; ++i) xv[i] = *reinterpret_cast<const f32x4*>(&lds[base[i] + off]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int j = 0; j < NW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][k], xv[i][k], acc[i][j], 0, 0, 0);
        }
    }

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

template <int NW, int MW, int WM, int WN, int TD, int TH, int TW, int S, int KD = 3>
int launch_mfma(ConvArgs a, hipStream_t st) {
    constexpr int ITD = (TD - 1) * (KD == 1 ? 1 : S) + KD, ITH = (TH - 1) * S + 3, ITW = (TW - 1) * S + 3;
    constexpr size_t lds_bytes = (size_t)ITD * ITH * ITW * kVS * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS tile too large");
    auto kern = conv3d_mfma_kernel<NW, MW, WM, WN, TD, TH, TW, S, KD>;
    static mvsgi::PersistentGeom geo_cache[mvsgi::kMaxDevices] = {};      // per device: the LDS limit is a per-device attribute
    mvsgi::PersistentGeom geo;
    if (mvsgi::persistent_geometry(kern, 256, lds_bytes, 8, geo_cache, "conv3d(f32 mfma)", geo)) return 1;
    a.tiles_d = (int)mvsgi::cdiv(a.Do, TD);
    a.tiles_h = (int)mvsgi::cdiv(a.Ho, TH);
    a.tiles_w = (int)mvsgi::cdiv(a.Wo, TW);
    const long long nt = (long long)a.B * a.tiles_d * a.tiles_h * a.tiles_w;
    MVSGI_REQUIRE(nt < (1ll << 31), "conv3d: too many tiles");
    const int CT = a.Cout / 16;
    dim3 grid((unsigned)nt, (unsigned)mvsgi::cdiv(CT, WN * NW));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
    return mvsgi::check_launch("mvsgi_conv3d_f32(mfma)");
}

