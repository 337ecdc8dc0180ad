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

#define CONV_EPT_GEN 12  // staged input elements per thread in the run-time-geometry variant

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
    int IH_T, IW_T, IW_P;  // LDS input-patch geometry (authoritative only for GEN kernels)
    int cin_pad;           // Cin rounded up to a multiple of 4
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

// input channels per LDS chunk: one-tile-per-wave configurations take 16 so that a chunk's MFMA chain
// (72 x 64 cycles) is long enough to cover the global-load latency of the next chunk's prefetch
__host__ __device__ constexpr int conv_ck(int ks, int tiles_per_wave, bool gen) {
    return ks == 3 ? ((tiles_per_wave == 1 && !gen) ? 16 : 8) : 16;
}
// LDS row pitch: rows of one N-tile must land on disjoint bank ranges for ds_read_b32 (32 banks):
// pitch == tw (mod 32) for tw in {16, 8}; any pitch >= width for tw == 32.
__host__ __device__ constexpr int conv_pitch(int tw, int iw_t, int stride) {
    if (tw == 32 || stride != 1) return iw_t;
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

__device__ __forceinline__ int stored_offset(const ConvKArgs& a, int vy, int vx) {
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

template <int WM, int WN, int RM, int RN, int TW, int KS, int DIL, bool GEN>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(ConvKArgs a) {
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
    constexpr int C_IWP = conv_pitch(TW, C_IW, 1);
    constexpr int EPT = GEN ? CONV_EPT_GEN : (CK * C_IH * C_IW + NT - 1) / NT;

    const int stride = GEN ? a.stride : 1;
    const int dil = GEN ? a.dil : DIL;
    const int IH_T = GEN ? a.IH_T : C_IH;
    const int IW_T = GEN ? a.IW_T : C_IW;
    const int IW_P = GEN ? a.IW_P : C_IWP;
    const int plane = IH_T * IW_P;
    const int tile_elems = IH_T * IW_T;
    const int total = CK * tile_elems;
    const int xs_floats = (CK * plane + 3) & ~3;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ws = smem + xs_floats;
    float* aff = ws + CK * KK * MT;  // [2][cin_pad] per-channel scale / shift of this image

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_x = (a.OW + TW - 1) / TW;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int ox0 = bx * TW, oy0 = by * PH;
    const int m0 = blockIdx.y * MT;
    const int n = blockIdx.z;
    const int HWi = a.H * a.W;  // tensors are < 2^31 elements
    const float* xn = a.x + (long)n * a.x_bs;
    const bool affine = a.in_scale != nullptr;
    const float in_slope = a.in_prelu ? *a.in_slope_ptr : 0.f;
    const int vy0 = oy0 * stride - a.pad, vx0 = ox0 * stride - a.pad;

    // ---- per-thread staging plan (identical for every channel chunk): element e = tid + t*NT of the
    // [CK][IH_T][IW_T] patch -> offset inside one stored channel plane (-1: reads zero, -2: no element)
    int goff[EPT];
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        int e = tid + t * NT;
        if (e < total) {
            int rem = e % tile_elems;
            int iy = rem / IW_T, ix = rem - iy * IW_T;
            goff[t] = stored_offset(a, vy0 + iy, vx0 + ix);
        } else {
            goff[t] = -2;
        }
    }
    constexpr int nq = CK * KK * ROW4;
    const int grow_end = a.Cin * KK;

    float xr[EPT];
    float4 wr[WPT];
    // issue the global loads of one channel chunk into registers (no dependent use -> all in flight)
    auto issue = [&](int c0) {
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            int ch = c0 + (tid + t * NT) / tile_elems;
            bool ok = goff[t] >= 0 && ch < a.Cin;
            xr[t] = xn[ok ? (unsigned)(ch * HWi + goff[t]) : 0u];
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            int q = tid + i * NT;
            int row = q / ROW4, col = (q % ROW4) * 4;
            int grow = c0 * KK + row;
            bool ok = q < nq && grow < grow_end && m0 + col < a.Cout;
            wr[i] = *reinterpret_cast<const float4*>(a.w + (ok ? (unsigned)(grow * a.Cout + m0 + col) : 0u));
        }
    };
    // transform + write the prefetched chunk into LDS
    auto commit = [&](int c0) {
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            if (goff[t] != -2) {
                int e = tid + t * NT;
                int c = e / tile_elems;
                int rem = e - c * tile_elems;
                int iy = rem / IW_T, ix = rem - iy * IW_T;
                int ch = c0 + c;
                bool ok = goff[t] >= 0 && ch < a.Cin;
                float v = 0.f;
                if (ok) {
                    v = xr[t];
                    if (affine) v = v * aff[ch] + aff[a.cin_pad + ch];
                    if (a.in_prelu) v = v >= 0.f ? v : v * in_slope;
                }
                xs[c * plane + iy * IW_P + ix] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            int q = tid + i * NT;
            if (q < nq) {
                int row = q / ROW4, col = (q % ROW4) * 4;
                int grow = c0 * KK + row;
                bool ok = grow < grow_end && m0 + col < a.Cout;
                *reinterpret_cast<float4*>(ws + row * MT + col) = ok ? wr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    issue(0);
    if (affine) {
        const float* scn = a.in_scale + (long)n * a.Cin;
        const float* shn = a.in_shift + (long)n * a.Cin;
        for (int i = tid; i < a.Cin; i += NT) {
            aff[i] = scn[i];
            aff[a.cin_pad + i] = shn[i];
        }
        __syncthreads();
    }
    commit(0);
    __syncthreads();

    f32x16 tot[RM][RN], acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;

    const float* xb[RN];  // this lane's pixel in the LDS patch (tap 0, channel `hi`)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        int t = wn * RN + j;
        int r = l31 / TW, c = l31 % TW;
        xb[j] = xs + hi * plane + ((t * RPT + r) * stride) * IW_P + c * stride;
    }
    const float* wb = ws + hi * KK * MT + wm * RM * 32 + l31;

    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        const bool has_next = c0 + CK < a.Cin;
        if (has_next) issue(c0 + CK);  // global latency hides under this chunk's MFMAs
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
                for (int j = 0; j < RN; ++j) bv[j] = xb[j][kk * plane + toff];
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) tot[i][j] += acc[i][j];
        if (has_next) {
            __syncthreads();  // every wave finished reading this chunk from LDS
            commit(c0 + CK);
            __syncthreads();
        }
    }

    // ---- epilogue: bias + residual + activation, coalesced NCHW store
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
template <int KS, int DIL, bool GEN, int WM, int WN, int RM, int RN>
static void conv_launch_tw(int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a) {
    constexpr int NT = 64 * WM * WN;
    switch (tw) {
        case 32: hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, RM, RN, 32, KS, DIL, GEN>), grid, dim3(NT), lds, s, a); break;
        case 16: hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, RM, RN, 16, KS, DIL, GEN>), grid, dim3(NT), lds, s, a); break;
        default: hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, RM, RN, 8, KS, DIL, GEN>), grid, dim3(NT), lds, s, a); break;
    }
}

template <int KS, int DIL, bool GEN>
static void conv_launch_variant(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a) {
    switch (cfg) {
        case 0: conv_launch_tw<KS, DIL, GEN, 1, 4, 2, 2>(tw, grid, lds, s, a); break;
        case 1: conv_launch_tw<KS, DIL, GEN, 1, 4, 1, 2>(tw, grid, lds, s, a); break;
        case 2: conv_launch_tw<KS, DIL, GEN, 1, 4, 2, 1>(tw, grid, lds, s, a); break;
        case 3: conv_launch_tw<KS, DIL, GEN, 1, 4, 1, 1>(tw, grid, lds, s, a); break;
        default: conv_launch_tw<KS, DIL, GEN, 2, 2, 1, 1>(tw, grid, lds, s, a); break;
    }
}

void conv_launch_k3d1(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a);
void conv_launch_k3d2(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a);
void conv_launch_k1(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a);
void conv_launch_gen(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a);
