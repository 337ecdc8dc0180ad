// Im2col-free implicit-GEMM convolution on the gfx950 fp32 matrix cores — kernel template.
//
// GEMM view (per image n):  D[co][pix] = sum_{ci,tap} Wp[ci][tap][co] * T(x)[ci][pix shifted by tap]
//   M = output channels  (MFMA A operand = weights,   A[i = lane&31][k = lane>>5])
//   N = output pixels    (MFMA B operand = input,     B[k = lane>>5][j = lane&31])
//   K = Cin * ks*ks, walked as (chunk of CK input channels) x (tap) x (2 channels per MFMA)
// With v_mfma_f32_32x32x2_f32 the D fragment is D[row = (reg&3)+8*(reg>>2)+4*(lane>>5)][col = lane&31],
// so lanes 0..31 of one accumulator register hold 32 consecutive pixels of ONE output channel:
// the NCHW epilogue store is a coalesced 128-byte row segment.
//
// Per workgroup (4 wave64): MT = 32*WM*RM output channels x (WN*RN) N-tiles of 32 pixels.  An N-tile
// is (32/TW) rows x TW columns, N-tiles are stacked vertically, so the block's pixel tile is
// PH = WN*RN*32/TW rows x TW columns.  For each chunk of CK input channels the block stages
//   xs[CK][IH_T][IW_P]  the input patch INCLUDING the halo (loaded once, reused by all ks*ks taps) with
//                       pad / reflect / nearest-upsample / subsample folded into the index map and the
//                       InstanceNorm affine (+PReLU) folded into the value, and
//   ws[CK][ks*ks][MT]   the weight slice (co contiguous -> conflict-free A reads)
// in LDS.  The chunk loop is software-pipelined: the global loads of chunk c+1 are issued into
// registers before the MFMAs of chunk c and committed to LDS after them.  For the common case
// (stride 1, dilation a template constant) the whole tile geometry is compile-time, so every LDS read
// of the fully unrolled ks*ks*CK/2-step MFMA loop is `base VGPR + immediate` and the compiler hoists
// reads far ahead of the MFMAs that consume them.  GEN=true keeps stride/dilation/geometry at run time
// (used for the single stride-2 layer, NonlocalNet.py:370).
//
// Accumulation: each chunk's ks*ks*CK-term fma chain starts from zero and is added to a running
// total (blocked summation, the shape a CPU GEMM has) — ~6x less rounding than one 4608-term chain.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
// (4 for cfg 3, which would fit LDS-wise, was measured: no gain, 128-VGPR cap costs spills)
#ifndef CONV_DMA_OCC
#define CONV_DMA_OCC 3   // workgroups per CU the one-tile-per-wave LDS-DMA variants are register-allocated for
#endif
#define CONV_AS1 __attribute__((address_space(1)))
#define CONV_AS3 __attribute__((address_space(3)))

#define CONV_EPT_GEN 12  // staged input elements per thread in the run-time-geometry variant
#define CONV_MAX_AFFINE_CIN 512  // static LDS affine table of the compile-time-geometry variants

struct ConvKArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* in_scale;
    const float* in_shift;
    const float* in_slope_ptr;
    const float* act_slope_ptr;
    const float* res;
    float* y;
    int N, Cin, H, W;   // stored input
    int VH, VW;         // virtual input (after up/sub-sampling)
    int Cout, OH, OW;
    int ks, stride, dil, pad, pad_mode, in_up, in_sub;
    int act, in_prelu;
    float act_slope;
    long x_bs, y_bs, res_bs;
    long w_bs;             // filter batch stride (0: one filter set for the batch; != 0: image n uses w + n * w_bs, a batched GEMM)
    int IH_T, IW_T, IW_P;  // LDS input-patch geometry (authoritative only for GEN kernels)
    int cin_pad;           // Cin rounded up to a multiple of 4
    int split;             // split-K factor S (1 = off): blockIdx.z = n*S + s, raw partial sums go to `part`
    int chunks_per_split;
    float* part;           // [S][N][Cout][OH][OW]
    long long* dbg_buf;    // timing experiments only: per-workgroup {start, end, HW_ID, XCC_ID} (dvc_debug_conv_trace)
    int gray;              // DVC_CONV_GRAY_INPUT (conv_image_kernel only): the stored input is ONE plane of centred luminance
    int dbg;               // timing experiments only (dvc_debug_conv_variant): 1 = no DMA after the first chunk,
                           // 2 = no patch DMA, 3 = no weight DMA after the first chunk (results are wrong)
};

struct ConvCfg {
    int wm, wn, rm, rn;
};
// ordered from fewest staged bytes per MFMA (largest tile) to most workgroups (smallest tile)
static const ConvCfg kConvCfgs[5] = {
    {1, 4, 2, 2},  // 0: 64 co x 8 N-tiles
    {1, 4, 1, 2},  // 1: 32 co x 8 N-tiles
    {1, 4, 2, 1},  // 2: 64 co x 4 N-tiles
    {1, 4, 1, 1},  // 3: 32 co x 4 N-tiles
    {2, 2, 1, 1},  // 4: 64 co x 2 N-tiles
};

// input channels per LDS chunk (8 keeps the double-buffered LDS footprint small enough for >= 2
// workgroups per CU; occupancy, not chunk length, hides the global-load latency)
__host__ __device__ constexpr int conv_ck(int ks, int tiles_per_wave, bool gen) {
    return ks == 3 ? 8 : 16;
}
// LDS buffer sizes (floats).  Both are padded so that the staging stores are unconditional (no exec-mask
// branches inside the chunk loop): threads without an input element write a dummy slot at the end of the
// patch buffer, and every weight float4 slot q < WPT*256 exists.
__host__ __device__ constexpr int conv_xs_floats(int ck, int ih_t, int iw_p) {
    return ((ck * ih_t * iw_p + 3) & ~3) + 4;
}
__host__ __device__ constexpr int conv_ws_floats(int ck, int kk, int mt) {
    return ((ck * kk * (mt / 4) + 255) / 256) * 256 * 4;
}
// LDS row pitch: rows of one N-tile must land on disjoint bank ranges for ds_read_b32 (32 banks):
// pitch == tw (mod 32) for tw in {16, 8}; any pitch >= width for tw == 32.
// The LDS-DMA variants take pitch == width instead: a 2-way conflict on the B reads (LDS is ~25 % busy)
// is cheaper than the 2.7x larger patch, which costs a resident workgroup per CU.
__host__ __device__ constexpr int conv_pitch(int tw, int iw_t, int stride, bool dma = false) {
    if (tw == 32 || stride != 1 || dma) return iw_t;
    int p = iw_t;
    while (p % 32 != tw) ++p;
    return p;
}

// virtual coordinate -> source coordinate, or -1 when the tap reads a zero
__device__ __forceinline__ int map_virtual(int v, int V, int pad_mode) {
    if (v < 0) {
        if (pad_mode != DVC_PAD_REFLECT) return -1;
        v = -v;
    } else if (v >= V) {
        if (pad_mode != DVC_PAD_REFLECT) return -1;
        v = 2 * (V - 1) - v;
    }
    return (v >= 0 && v < V) ? v : -1;  // far outside only happens for discarded partial-tile outputs
}

// (A = ConvKArgs in whatever address space the caller holds it: generic, or the kernel-argument segment)
template <class A>
__device__ __forceinline__ int stored_offset(const A& a, int vy, int vx) {
    int sy = map_virtual(vy, a.VH, a.pad_mode);
    int sx = map_virtual(vx, a.VW, a.pad_mode);
    if (sy < 0 || sx < 0) return -1;
    if (a.in_up == 2) {
        sy >>= 1;
        sx >>= 1;
    } else if (a.in_sub == 2) {
        sy <<= 1;
        sx <<= 1;
    }
    return sy * a.W + sx;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case DVC_ACT_RELU: return v > 0.f ? v : 0.f;
        case DVC_ACT_PRELU:
        case DVC_ACT_LEAKY: return v >= 0.f ? v : v * slope;
        case DVC_ACT_TANH128: return tanhf(v) * 128.f;
        default: return v;
    }
}

// DMA = true ("plain" layers: no input affine / PReLU, Cin % CK == 0, Cout % MT == 0, compile-time
// geometry): the patch and the weight slice go global -> LDS by LDS-DMA (global_load_lds), no staging
// registers, no commit phase and (almost) no VALU in the chunk loop.  That matters more than it would on
// other matrix pipes: the fp32 MFMA shares the SIMD's fp32 lanes with ordinary VALU work (vector and
// matrix fp32 peaks are the same number), so every staging VALU instruction is taken from the MFMA rate.
// The LDS patch image is then linear in the staged element index (pitch padding included); cells that
// read padding zeros are zeroed once and never written again (their lanes are masked off in the DMA).
// Waves per SIMD the variant is register-allocated for: 3 workgroups/CU for the one-tile-per-wave LDS-DMA variants
// (2 for the dilated 32-wide one, whose patch is larger), 1 for the register-staging variants with several tiles per wave
// and a narrow pixel tile (their staging registers do not fit twice), 2 otherwise — what each variant actually reaches.
__host__ __device__ constexpr int conv_min_waves(int rmrn, int tw, int dil, bool gen, bool dma) {
    if (dma && rmrn == 1) return (dil == 2 && tw == 32) ? 2 : CONV_DMA_OCC;
    if (!dma && !gen && rmrn > 1 && tw < 32) return 1;
    return 2;
}

template <int WM, int WN, int RM, int RN, int TW, int KS, int DIL, bool GEN, bool DMA = false>
__global__ __launch_bounds__(64 * WM * WN, conv_min_waves(RM * RN, TW, DIL, GEN, DMA)) void conv_mfma_kernel(ConvKArgs a) {
    static_assert(!(GEN && DMA), "LDS-DMA staging needs compile-time geometry");
    constexpr int NT = 64 * WM * WN;
    constexpr int MT = 32 * WM * RM;
    constexpr int RPT = 32 / TW;  // rows per 32-pixel N-tile
    constexpr int PH = WN * RN * RPT;
    constexpr int KK = KS * KS;
    constexpr int CK = conv_ck(KS, RM * RN, GEN);
    constexpr int ROW4 = MT / 4;
    constexpr int WPT = (CK * KK * ROW4 + NT - 1) / NT;  // float4 weight loads per thread per chunk
    // compile-time geometry (ignored by GEN kernels)
    constexpr int C_IH = PH + DIL * (KS - 1);
    constexpr int C_IW = TW + DIL * (KS - 1);
    constexpr int C_IWP = conv_pitch(TW, C_IW, 1, DMA);
    constexpr int EPT = GEN ? CONV_EPT_GEN : DMA ? (CK * C_IH * C_IWP + NT - 1) / NT : (CK * C_IH * C_IW + NT - 1) / NT;

    const int stride = GEN ? a.stride : 1;
    const int dil = GEN ? a.dil : DIL;
    const int IH_T = GEN ? a.IH_T : C_IH;
    const int IW_T = GEN ? a.IW_T : C_IW;
    const int IW_P = GEN ? a.IW_P : C_IWP;
    const int plane = IH_T * IW_P;
    const int tile_elems = IH_T * IW_T;
    const int total = CK * tile_elems;
    const int xs_floats = conv_xs_floats(CK, IH_T, IW_P);

    // LDS: two input-patch buffers and two weight-slice buffers (double buffering: chunk c+1 is written
    // while chunk c is read, one barrier per chunk) + the per-channel affine table.  For the
    // compile-time-geometry variants these are FOUR DISTINCT static arrays: only then can alias analysis
    // prove that the staging stores into one buffer do not touch the MFMA chain's reads of the other,
    // which is what allows the scheduler to interleave the two streams.
    constexpr int ws_floats = conv_ws_floats(CK, KK, MT);
    constexpr int C_XS = GEN ? 4 : conv_xs_floats(CK, C_IH, C_IWP);
    constexpr int C_WS = GEN ? 4 : ws_floats;
    constexpr int AFF_MAX = (GEN || DMA) ? 4 : 2 * (CONV_MAX_AFFINE_CIN + 16);
    __shared__ __attribute__((aligned(16))) float s_xs0[C_XS];
    __shared__ __attribute__((aligned(16))) float s_xs1[C_XS];
    __shared__ __attribute__((aligned(16))) float s_ws0[C_WS];
    __shared__ __attribute__((aligned(16))) float s_ws1[C_WS];
    __shared__ __attribute__((aligned(16))) float s_aff[AFF_MAX];
    extern __shared__ __attribute__((aligned(16))) float smem[];  // GEN only
    float* const xsb0 = GEN ? smem : s_xs0;
    float* const xsb1 = GEN ? smem + xs_floats : s_xs1;
    float* const wsb0 = GEN ? smem + 2 * xs_floats : s_ws0;
    float* const wsb1 = GEN ? smem + 2 * xs_floats + ws_floats : s_ws1;
    float* const aff = GEN ? smem + 2 * xs_floats + 2 * ws_floats : s_aff;  // [2][cin_pad + CK]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long* dbgp = nullptr;
    if (kDvcDebug && DMA && a.dbg_buf && tid == 0) {
        dbgp = a.dbg_buf + 4L * (blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z));
        dbgp[0] = __builtin_amdgcn_s_memtime();
        dbgp[2] = __builtin_amdgcn_s_getreg(63492);   // HW_ID
        dbgp[3] = __builtin_amdgcn_s_getreg(63508);   // XCC_ID
    }
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_x = (a.OW + TW - 1) / TW;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int ox0 = bx * TW, oy0 = by * PH;
    const int m0 = blockIdx.y * MT;
    const int n = blockIdx.z / a.split, ksplit = blockIdx.z % a.split;
    const int HWi = a.H * a.W;  // tensors are < 2^31 elements
    const float* xn = a.x + (long)n * a.x_bs;
    const float* wn_ = a.w + (long)n * a.w_bs;
    const bool affine = a.in_scale != nullptr;
    const float in_slope = a.in_prelu ? *a.in_slope_ptr : 0.f;
    const int vy0 = oy0 * stride - a.pad, vx0 = ox0 * stride - a.pad;

    // ---- per-thread staging plan, computed once (identical for every channel chunk).
    // x element e = tid + t*NT of the [CK][IH_T][IW_T] patch:
    //   gofs[t]  offset from the chunk's first channel plane (c*HW + y*W + x), -1 = reads zero,
    //            -2 = no such element
    //   lofs[t]  (channel-in-chunk << 20) | LDS float offset
    // weight float4 q = tid + i*NT of the [CK*KK][MT] slice:  wofs[i] = row*Cout + m0 + col (or -1)
    int gofs[EPT], lofs[EPT];
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        int e = tid + t * NT;
        if (DMA) {  // element e of the PITCHED image [CK][IH_T][IW_P]; LDS float offset == e
            int c = e / plane;
            int rem = e - c * plane;
            int iy = rem / IW_P, ix = rem - iy * IW_P;
            int g = (e < CK * plane && ix < IW_T) ? stored_offset(a, vy0 + iy, vx0 + ix) : -1;
            gofs[t] = g >= 0 ? c * HWi + g : -1;
            lofs[t] = 0;
        } else if (e < total) {
            int c = e / tile_elems;
            int rem = e - c * tile_elems;
            int iy = rem / IW_T, ix = rem - iy * IW_T;
            int g = stored_offset(a, vy0 + iy, vx0 + ix);
            gofs[t] = g >= 0 ? c * HWi + g : -1;
            lofs[t] = (c << 20) | (c * plane + iy * IW_P + ix);
        } else {
            gofs[t] = -2;
            lofs[t] = xs_floats - 1;  // dummy slot
        }
    }
    constexpr int nq = CK * KK * ROW4;
    int wofs[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        int q = tid + i * NT;
        int row = q / ROW4, col = (q % ROW4) * 4;
        wofs[i] = (q < nq && m0 + col < a.Cout) ? row * a.Cout + m0 + col : -1;
    }
    const int nchunks_all = (a.Cin + CK - 1) / CK;
    const int c_begin = ksplit * a.chunks_per_split;
    const int nchunks = min(nchunks_all, c_begin + a.chunks_per_split);  // this block walks [c_begin, nchunks)
    const bool ragged = (a.Cin % CK) != 0;  // only then can a staged channel lie beyond Cin

    float xr[EPT];
    float4 wr[WPT];
    // issue the global loads of chunk `ci` into registers (nothing consumes them until commit)
    auto issue = [&](int ci) {
        const int c0 = ci * CK;
        const int xbase = c0 * HWi;
        const int wbase = c0 * KK * a.Cout;
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            bool ok = gofs[t] >= 0 && (!ragged || c0 + (lofs[t] >> 20) < a.Cin);
            xr[t] = xn[ok ? (unsigned)(xbase + gofs[t]) : 0u];
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            bool ok = wofs[i] >= 0 && (!ragged || c0 + (tid + i * NT) / (KK * ROW4) < a.Cin);
            wr[i] = *reinterpret_cast<const float4*>(wn_ + (ok ? (unsigned)(wbase + wofs[i]) : 0u));
        }
    };
    // transform + write the prefetched chunk `ci` into LDS buffer `buf`
    auto commit = [&](int ci, float* xs, float* ws) {
        const int c0 = ci * CK;
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            const int ch = c0 + (lofs[t] >> 20);
            const bool ok = gofs[t] >= 0 && (!ragged || ch < a.Cin);
            float v = xr[t];
            if (affine) v = v * aff[ch] + aff[a.cin_pad + CK + ch];   // table is zero-padded past Cin
            if (a.in_prelu) v = v >= 0.f ? v : v * in_slope;
            xs[lofs[t] & 0xFFFFF] = ok ? v : 0.f;   // unconditional store (dummy slot if no element)
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int q = tid + i * NT;
            const bool ok = wofs[i] >= 0 && (!ragged || c0 + q / (KK * ROW4) < a.Cin);
            *reinterpret_cast<float4*>(ws + q * 4) = ok ? wr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    // LDS-DMA staging of chunk `ci` (DMA variants): every lane supplies its own global address, the LDS
    // destination is wave-uniform base + lane * size.
    const bool dbg_nox = kDvcDebug && (a.dbg == 1 || a.dbg == 2), dbg_now = kDvcDebug && (a.dbg == 1 || a.dbg == 3);
    auto issue_dma = [&](int ci, float* xs, float* ws) {
        const float* xc = xn + (long)ci * CK * HWi;
        const float* wc = wn_ + (long)ci * CK * KK * a.Cout;
#pragma unroll
        for (int t = 0; t < EPT; ++t)
            if (gofs[t] >= 0 && !(dbg_nox && ci != c_begin))
                __builtin_amdgcn_global_load_lds((const CONV_AS1 void*)(xc + (unsigned)gofs[t]),
                                                 (CONV_AS3 void*)(xs + t * NT + wave * 64), 4, 0, 0);
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (wofs[i] >= 0 && !(dbg_now && ci != c_begin))
                __builtin_amdgcn_global_load_lds((const CONV_AS1 void*)(wc + (unsigned)wofs[i]),
                                                 (CONV_AS3 void*)(ws + (i * NT + wave * 64) * 4), 16, 0, 0);
    };

    if (DMA) {
        for (int i = tid; i < C_XS / 4; i += NT) {
            reinterpret_cast<float4*>(xsb0)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            reinterpret_cast<float4*>(xsb1)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        issue_dma(c_begin, xsb0, wsb0);
        __syncthreads();  // (drains the DMA: vmcnt(0) before the barrier)
    } else {
    issue(c_begin);
    if (affine) {
        const float* scn = a.in_scale + (long)n * a.Cin;
        const float* shn = a.in_shift + (long)n * a.Cin;
        for (int i = tid; i < a.cin_pad + CK; i += NT) {   // (+CK: a ragged last chunk indexes past Cin)
            aff[i] = i < a.Cin ? scn[i] : 0.f;
            aff[a.cin_pad + CK + i] = i < a.Cin ? shn[i] : 0.f;
        }
        __syncthreads();
    }
    commit(c_begin, xsb0, wsb0);
    issue(min(c_begin + 1, nchunks - 1));
    __syncthreads();
    }

    // tot: running sum; acc: one chunk's MFMA chain.  Flushing per chunk keeps every fp32 chain short
    // (CK*ks*ks terms) and makes the total a sum of Cin/CK partials — blocked summation.
    f32x16 tot[RM][RN], acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;

    int xoff[RN];  // this lane's pixel in the LDS patch (tap 0, channel `hi`), float offset
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        int t = wn * RN + j;
        int r = l31 / TW, c = l31 % TW;
        xoff[j] = hi * plane + ((t * RPT + r) * stride) * IW_P + c * stride;
    }
    const int woff = hi * KK * MT + wm * RM * 32 + l31;

    // One chunk: write chunk ci+1 (already in registers) into the OTHER LDS buffer, refill the
    // registers with chunk ci+2, run chunk ci's MFMA chain.  Single basic block; the scheduler is
    // asked (sched_group_barrier) to spread the staging instructions between the MFMAs.  Past the
    // last chunk the clamped indices re-stage the last chunk: harmless.
    auto chunk = [&](int ci, const float* xs, const float* ws, float* xs_next, float* ws_next) {
        if (DMA) {
            if (ci + 1 < nchunks) issue_dma(ci + 1, xs_next, ws_next);
        } else {
            commit(min(ci + 1, nchunks - 1), xs_next, ws_next);
            issue(min(ci + 2, nchunks - 1));
        }
        const float* wb = ws + woff;
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            const int toff = ky * dil * IW_P + kx * dil;
#pragma unroll
            for (int kk = 0; kk < CK; kk += 2) {
                float av[RM], bv[RN];
#pragma unroll
                for (int i = 0; i < RM; ++i) av[i] = wb[(kk * KK + tap) * MT + i * 32];
#pragma unroll
                for (int j = 0; j < RN; ++j) bv[j] = xs[xoff[j] + kk * plane + toff];
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        // interleave request: per MFMA two LDS reads and a few VALU; every 4th MFMA one global load and
        // one LDS write of the staging stream
#pragma unroll
        for (int i = 0; i < KK * (CK / 2) * RM * RN; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
            if (DMA) continue;
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
            if ((i & 3) == 0) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
            }
        }
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) tot[i][j] += acc[i][j];
        __syncthreads();  // chunk ci+1 is visible in LDS; everyone is done reading chunk ci
    };
    for (int ci = c_begin; ci < nchunks; ci += 2) {
        chunk(ci, xsb0, wsb0, xsb1, wsb1);
        if (ci + 1 < nchunks) chunk(ci + 1, xsb1, wsb1, xsb0, wsb0);
    }

    if (kDvcDebug && dbgp) dbgp[1] = __builtin_amdgcn_s_memtime();
    // ---- epilogue
    if (a.split > 1) {  // split-K: raw partial sums; bias / residual / activation happen in the reduce kernel
        const long OHW = (long)a.OH * a.OW;
        float* pn = a.part + ((long)ksplit * a.N + n) * a.Cout * OHW;
        const int pr = l31 / TW, pc = l31 % TW;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            int t = wn * RN + j;
            int oy = oy0 + t * RPT + pr, ox = ox0 + pc;
            if (oy >= a.OH || ox >= a.OW) continue;
            long pix = (long)oy * a.OW + ox;
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int co = m0 + (wm * RM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (co < a.Cout) pn[(long)co * OHW + pix] = tot[i][j][r];
                }
        }
        return;
    }
    // bias + residual + activation, coalesced NCHW store
    const float slope = a.act_slope_ptr ? *a.act_slope_ptr : a.act_slope;
    const long OHW = (long)a.OH * a.OW;
    float* yn = a.y + (long)n * a.y_bs;
    const float* rn_ = a.res ? a.res + (long)n * a.res_bs : nullptr;
    const int pr = l31 / TW, pc = l31 % TW;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        int t = wn * RN + j;
        int oy = oy0 + t * RPT + pr;
        int ox = ox0 + pc;
        if (oy >= a.OH || ox >= a.OW) continue;
        long pix = (long)oy * a.OW + ox;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = m0 + (wm * RM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (co < a.Cout) {
                    float v = tot[i][j][r];
                    if (a.bias) v += a.bias[co];
                    if (rn_) v += rn_[(long)co * OHW + pix];
                    yn[(long)co * OHW + pix] = apply_act(v, a.act, slope);
                }
            }
        }
    }
}

// ---- per-variant launchers (one translation unit each, so they compile in parallel)
template <int KS, int DIL, bool GEN, bool DMA, int WM, int WN, int RM, int RN>
static void conv_launch_tw(int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a) {
    constexpr int NT = 64 * WM * WN;
    switch (tw) {
        case 32: hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, RM, RN, 32, KS, DIL, GEN, DMA>), grid, dim3(NT), lds, s, a); break;
        case 16: hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, RM, RN, 16, KS, DIL, GEN, DMA>), grid, dim3(NT), lds, s, a); break;
        default: hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, RM, RN, 8, KS, DIL, GEN, DMA>), grid, dim3(NT), lds, s, a); break;
    }
}

template <int KS, int DIL, bool GEN, bool DMA = false>
static void conv_launch_variant(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a) {
    switch (cfg) {
        case 0: conv_launch_tw<KS, DIL, GEN, DMA, 1, 4, 2, 2>(tw, grid, lds, s, a); break;
        case 1: conv_launch_tw<KS, DIL, GEN, DMA, 1, 4, 1, 2>(tw, grid, lds, s, a); break;
        case 2: conv_launch_tw<KS, DIL, GEN, DMA, 1, 4, 2, 1>(tw, grid, lds, s, a); break;
        case 3: conv_launch_tw<KS, DIL, GEN, DMA, 1, 4, 1, 1>(tw, grid, lds, s, a); break;
        default: conv_launch_tw<KS, DIL, GEN, DMA, 2, 2, 1, 1>(tw, grid, lds, s, a); break;
    }
}

void conv_launch_k3d1(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a);
void conv_launch_k3d2(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a);
void conv_launch_k1(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a);
void conv_launch_gen(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a);
void conv_launch_k3d1_dma(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a);
void conv_launch_k3d2_dma(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a);
void conv_launch_k1_dma(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a);
