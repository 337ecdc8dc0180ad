// Im2col-free implicit-GEMM convolution on the gfx950 fp32 matrix cores.
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
// into LDS, then runs ks*ks*CK/2 MFMA steps per register tile.  fp32 MFMA is 64 cycles per
// instruction per SIMD, so LDS bandwidth (2 ds_read_b32 per 1..4 MFMAs) is never the limiter; the
// design goal is enough workgroups (>= 1-2 waves per SIMD) and few staged bytes per MFMA.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CONV_EPT 12  // staged input elements per thread kept as precomputed offsets

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
    int ck;             // input channels per chunk (even)
    int IH_T, IW_T, IW_P;
    int xs_floats;      // CK*IH_T*IW_P rounded up to a multiple of 4
    int ws_floats;      // CK*ks*ks*MT
    int cin_pad;        // Cin rounded up to a multiple of 4
};

// virtual coordinate -> stored offset component, or -1 when the tap reads a zero
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

template <int WM, int WN, int RM, int RN, int TW, int KS>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(ConvKArgs a) {
    constexpr int NT = 64 * WM * WN;
    constexpr int KK = KS * KS;
    constexpr int CK = KS == 3 ? 8 : 16;  // input channels per chunk (compile-time: the MFMA loop fully unrolls)
    constexpr int MT = 32 * WM * RM;
    constexpr int RPT = 32 / TW;  // rows per 32-pixel N-tile
    constexpr int PH = WN * RN * RPT;
    constexpr int ROW4 = MT / 4;
    constexpr int WPT = (CK * KK * ROW4 + NT - 1) / NT;  // float4 weight loads per thread per chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ws = smem + a.xs_floats;
    float* aff = ws + a.ws_floats;  // [2][cin_pad] per-channel scale / shift of this image

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_x = (a.OW + TW - 1) / TW;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int ox0 = bx * TW, oy0 = by * PH;
    const int m0 = blockIdx.y * MT;
    const int n = blockIdx.z;
    const int HWi = a.H * a.W;
    const float* xn = a.x + (long)n * a.x_bs;
    const bool affine = a.in_scale != nullptr;
    const float in_slope = a.in_prelu ? *a.in_slope_ptr : 0.f;

    const int plane = a.IH_T * a.IW_P;
    const int tile_elems = a.IH_T * a.IW_T;
    const int total = CK * tile_elems;  // host guarantees total <= CONV_EPT * NT
    const int vy0 = oy0 * a.stride - a.pad, vx0 = ox0 * a.stride - a.pad;

    // ---- per-thread staging plan (identical for every channel chunk)
    int goff[CONV_EPT];   // offset inside one channel plane, -1 = zero, -2 = nothing to do
    int lpack[CONV_EPT];  // (channel-in-chunk << 24) | LDS float offset
#pragma unroll
    for (int t = 0; t < CONV_EPT; ++t) {
        int e = tid + t * NT;
        if (e < total) {
            int c = e / tile_elems;
            int rem = e - c * tile_elems;
            int iy = rem / a.IW_T;
            int ix = rem - iy * a.IW_T;
            goff[t] = stored_offset(a, vy0 + iy, vx0 + ix);
            lpack[t] = (c << 24) | (c * plane + iy * a.IW_P + ix);
        } else {
            goff[t] = -2;
            lpack[t] = 0;
        }
    }
    const int nq = CK * KK * ROW4;
    const int grow_end = a.Cin * KK;

    float xr[CONV_EPT];
    float4 wr[WPT];
    // issue the global loads of one channel chunk into registers (no dependent use -> all in flight)
    auto issue = [&](int c0) {
#pragma unroll
        for (int t = 0; t < CONV_EPT; ++t) {
            int ch = c0 + (lpack[t] >> 24);
            bool ok = goff[t] >= 0 && ch < a.Cin;
            xr[t] = xn[ok ? (unsigned)(ch * HWi + goff[t]) : 0u];  // tensors are < 2^31 elements
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
        for (int t = 0; t < CONV_EPT; ++t) {
            if (goff[t] != -2) {
                int ch = c0 + (lpack[t] >> 24);
                bool ok = goff[t] >= 0 && ch < a.Cin;
                float v = 0.f;
                if (ok) {
                    v = xr[t];
                    if (affine) v = v * aff[ch] + aff[a.cin_pad + ch];
                    if (a.in_prelu) v = v >= 0.f ? v : v * in_slope;
                }
                xs[lpack[t] & 0xFFFFFF] = v;
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

    // tot: running sum; acc: one chunk's MFMA chain.  Flushing per chunk keeps every fp32 chain short
    // (CK*ks*ks terms) and makes the total a sum of Cin/CK partials — the same blocked summation
    // shape as a CPU GEMM, ~6x less rounding than one 4608-term fma chain.
    f32x16 tot[RM][RN], acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;

    int xbase[RN];  // per-lane LDS offset of this lane's pixel (tap 0, channel `hi`)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        int t = wn * RN + j;
        int r = l31 / TW, c = l31 % TW;
        xbase[j] = hi * plane + ((t * RPT + r) * a.stride) * a.IW_P + c * a.stride;
    }
    const float* wbase = ws + hi * KK * MT + wm * RM * 32 + l31;
    const int dIW = a.dil * a.IW_P;

    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        const bool has_next = c0 + CK < a.Cin;
        if (has_next) issue(c0 + CK);  // global latency hides under this chunk's MFMAs
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // fully unrolled: KK*CK/2 k-steps; weight offsets are immediates, the compiler hoists the
        // ds_reads of later steps above the MFMAs of earlier ones (no LDS latency between MFMAs)
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            const int toff = ky * dIW + kx * a.dil;
#pragma unroll
            for (int kk = 0; kk < CK; kk += 2) {
                float av[RM], bv[RN];
#pragma unroll
                for (int i = 0; i < RM; ++i) av[i] = wbase[(kk * KK + tap) * MT + i * 32];
#pragma unroll
                for (int j = 0; j < RN; ++j) bv[j] = xs[xbase[j] + kk * plane + toff];
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

// ------------------------------------------------------------------------------------------------
struct ConvCfg {
    int wm, wn, rm, rn;
};
// ordered from fewest staged bytes per MFMA (largest tile) to most workgroups (smallest tile)
static const ConvCfg kCfgs[5] = {
    {1, 4, 2, 2},  // 0: 64 co x 8 N-tiles
    {1, 4, 1, 2},  // 1: 32 co x 8 N-tiles
    {1, 4, 2, 1},  // 2: 64 co x 4 N-tiles
    {1, 4, 1, 1},  // 3: 32 co x 4 N-tiles
    {2, 2, 1, 1},  // 4: 64 co x 2 N-tiles
};

template <int WM, int WN, int RM, int RN, int KS>
static void launch_tw(int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a) {
    constexpr int NT = 64 * WM * WN;
    switch (tw) {
        case 32: hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, RM, RN, 32, KS>), grid, dim3(NT), lds, s, a); break;
        case 16: hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, RM, RN, 16, KS>), grid, dim3(NT), lds, s, a); break;
        default: hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, RM, RN, 8, KS>), grid, dim3(NT), lds, s, a); break;
    }
}

template <int KS>
static void launch_cfg(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a) {
    switch (cfg) {
        case 0: launch_tw<1, 4, 2, 2, KS>(tw, grid, lds, s, a); break;
        case 1: launch_tw<1, 4, 1, 2, KS>(tw, grid, lds, s, a); break;
        case 2: launch_tw<1, 4, 2, 1, KS>(tw, grid, lds, s, a); break;
        case 3: launch_tw<1, 4, 1, 1, KS>(tw, grid, lds, s, a); break;
        default: launch_tw<2, 2, 1, 1, KS>(tw, grid, lds, s, a); break;
    }
}

static int virt_dim(int S, int up, int sub) {
    if (up == 2) return S * 2;
    if (sub == 2) return (S + 1) / 2;
    return S;
}

extern "C" int dvc_conv2d_out_hw(const DvcConvDesc* d, int32_t* OH, int32_t* OW) {
    DVC_REQUIRE(d, "dvc_conv2d_out_hw: null descriptor");
    int VH = virt_dim(d->H, d->in_up, d->in_sub), VW = virt_dim(d->W, d->in_up, d->in_sub);
    int ext = d->dil * (d->ksize - 1) + 1;
    *OH = (VH + 2 * d->pad - ext) / d->stride + 1;
    *OW = (VW + 2 * d->pad - ext) / d->stride + 1;
    return 0;
}

static int pick_tw(int OW) {
    int best = 32, best_w = cdiv(OW, 32) * 32;
    for (int tw : {16, 8}) {
        int wpad = cdiv(OW, tw) * tw;
        if (wpad < best_w) {
            best = tw;
            best_w = wpad;
        }
    }
    return best;
}

extern "C" int dvc_conv2d(const DvcConvDesc* d, const float* x, const float* w_packed,
                          const float* bias, const float* in_scale, const float* in_shift,
                          const float* in_slope_ptr, const float* act_slope_ptr,
                          const float* residual, float* y, dvcStream stream) {
    DVC_REQUIRE(d && x && w_packed && y, "dvc_conv2d: null argument");
    DVC_REQUIRE(d->ksize == 1 || d->ksize == 3, "dvc_conv2d: ksize must be 1 or 3 (got %d)", d->ksize);
    DVC_REQUIRE(d->stride == 1 || d->stride == 2, "dvc_conv2d: stride must be 1 or 2");
    DVC_REQUIRE(d->dil == 1 || d->dil == 2, "dvc_conv2d: dilation must be 1 or 2");
    DVC_REQUIRE(d->pad >= 0 && d->pad <= 2, "dvc_conv2d: pad must be 0..2");
    DVC_REQUIRE(d->Cout % 4 == 0, "dvc_conv2d: Cout must be a multiple of 4 (got %d)", d->Cout);
    DVC_REQUIRE((d->in_up == 1 || d->in_up == 2) && (d->in_sub == 1 || d->in_sub == 2) &&
                    !(d->in_up == 2 && d->in_sub == 2),
                "dvc_conv2d: bad in_up/in_sub");
    DVC_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dvc_conv2d: scale/shift must come together");
    DVC_REQUIRE(!d->in_prelu || in_slope_ptr, "dvc_conv2d: in_prelu needs in_slope_ptr");
    DVC_REQUIRE(d->N > 0 && d->Cin > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "dvc_conv2d: bad shape");
    DVC_REQUIRE((reinterpret_cast<uintptr_t>(w_packed) & 15) == 0, "dvc_conv2d: weights must be 16-byte aligned");

    ConvKArgs a;
    a.x = x; a.w = w_packed; a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift;
    a.in_slope_ptr = in_slope_ptr; a.act_slope_ptr = act_slope_ptr; a.res = residual; a.y = y;
    a.N = d->N; a.Cin = d->Cin; a.H = d->H; a.W = d->W;
    a.VH = virt_dim(d->H, d->in_up, d->in_sub);
    a.VW = virt_dim(d->W, d->in_up, d->in_sub);
    int32_t OH, OW;
    dvc_conv2d_out_hw(d, &OH, &OW);
    DVC_REQUIRE(OH > 0 && OW > 0, "dvc_conv2d: empty output");
    if (d->pad_mode == DVC_PAD_REFLECT)
        DVC_REQUIRE(d->pad < a.VH && d->pad < a.VW, "dvc_conv2d: reflect pad needs pad < input size");
    a.Cout = d->Cout; a.OH = OH; a.OW = OW;
    a.ks = d->ksize; a.stride = d->stride; a.dil = d->dil; a.pad = d->pad; a.pad_mode = d->pad_mode;
    a.in_up = d->in_up; a.in_sub = d->in_sub; a.act = d->act; a.in_prelu = d->in_prelu;
    a.act_slope = d->act_slope;
    a.x_bs = d->x_batch_stride ? d->x_batch_stride : (long)d->Cin * d->H * d->W;
    a.y_bs = d->y_batch_stride ? d->y_batch_stride : (long)d->Cout * OH * OW;
    a.res_bs = d->res_batch_stride ? d->res_batch_stride : (long)d->Cout * OH * OW;

    const int tw = pick_tw(OW);
    const int rpt = 32 / tw;
    const int ck = d->ksize == 3 ? 8 : 16;
    // a configuration is usable when one chunk of its input patch fits the per-thread staging plan
    auto fits = [&](int i) {
        const ConvCfg& c = kCfgs[i];
        int ph = c.wn * c.rn * rpt;
        int ih = (ph - 1) * d->stride + d->dil * (d->ksize - 1) + 1;
        int iw = (tw - 1) * d->stride + d->dil * (d->ksize - 1) + 1;
        return ck * ih * iw <= CONV_EPT * 256;
    };
    int cfg = d->cfg;
    if (cfg < 0) {
        // largest tile that still gives >= 2 waves per SIMD (2048 waves); else >= 1; else most waves
        int first1 = -1, first2 = -1, most = -1;
        long most_waves = -1;
        for (int i = 0; i < 5; ++i) {
            if (!fits(i)) continue;
            const ConvCfg& c = kCfgs[i];
            int mt = 32 * c.wm * c.rm, ph = c.wn * c.rn * rpt;
            long waves = 4L * cdiv(OW, tw) * cdiv(OH, ph) * cdiv(d->Cout, mt) * d->N;
            if (d->Cout < mt && i != 1 && i != 3) continue;  // don't waste half the M tile
            if (waves >= 2048 && first2 < 0) first2 = i;
            if (waves >= 1024 && first1 < 0) first1 = i;
            if (waves > most_waves) { most_waves = waves; most = i; }
        }
        cfg = first2 >= 0 ? first2 : (first1 >= 0 ? first1 : most);
        DVC_REQUIRE(cfg >= 0, "dvc_conv2d: no tile configuration fits this geometry");
    }
    DVC_REQUIRE(cfg >= 0 && cfg < 5, "dvc_conv2d: cfg out of range");
    DVC_REQUIRE(fits(cfg), "dvc_conv2d: tile configuration %d does not fit this geometry", cfg);
    const ConvCfg& c = kCfgs[cfg];
    const int mt = 32 * c.wm * c.rm, ph = c.wn * c.rn * rpt;
    a.ck = ck;
    a.IH_T = (ph - 1) * d->stride + d->dil * (d->ksize - 1) + 1;
    a.IW_T = (tw - 1) * d->stride + d->dil * (d->ksize - 1) + 1;
    // row pitch: rows of one N-tile must land on disjoint bank ranges for ds_read_b32 (32 banks):
    // pitch == tw (mod 32) for tw in {16, 8}; anything >= IW_T for tw == 32.
    int pitch = a.IW_T;
    if (tw < 32 && d->stride == 1) {
        while (pitch % 32 != tw) ++pitch;
    }
    a.IW_P = pitch;
    a.xs_floats = (a.ck * a.IH_T * a.IW_P + 3) & ~3;
    a.ws_floats = a.ck * a.ks * a.ks * mt;
    a.cin_pad = (d->Cin + 3) & ~3;
    size_t lds = sizeof(float) * ((size_t)a.xs_floats + (size_t)a.ws_floats + (in_scale ? 2 * (size_t)a.cin_pad : 0));
    DVC_REQUIRE(lds <= 160 * 1024, "dvc_conv2d: LDS tile too large (%zu bytes)", lds);
    dim3 grid(cdiv(OW, tw) * cdiv(OH, ph), cdiv(d->Cout, mt), d->N);
    hipStream_t s = (hipStream_t)stream;
    if (d->ksize == 3) launch_cfg<3>(cfg, tw, grid, lds, s, a);
    else launch_cfg<1>(cfg, tw, grid, lds, s, a);
    DVC_CHECK_LAUNCH("dvc_conv2d");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// 1x1 conv with Cout <= 4 (ColorVidNet.conv10_ab: 128 -> 2, then tanh*128).  Bandwidth-trivial:
// one thread per pixel, channel loop with coalesced reads along the pixel axis.
template <int COUT>
__global__ void conv1x1_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                     const float* __restrict__ bias, int Cin, long HW, int act,
                                     float* __restrict__ y) {
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int n = blockIdx.y;
    if (p >= HW) return;
    const float* xn = x + (long)n * Cin * HW + p;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    for (int c = 0; c < Cin; ++c) {
        float v = xn[(long)c * HW];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = fmaf(v, w[o * Cin + c], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
        float v = acc[o] + (bias ? bias[o] : 0.f);
        y[((long)n * COUT + o) * HW + p] = apply_act(v, act, 0.f);
    }
}

extern "C" int dvc_conv1x1_small(const float* x, const float* w, const float* bias, int32_t N,
                                 int32_t Cin, int32_t HW, int32_t Cout, int32_t act, float* y,
                                 dvcStream stream) {
    DVC_REQUIRE(x && w && y, "dvc_conv1x1_small: null argument");
    DVC_REQUIRE(Cout >= 1 && Cout <= 4, "dvc_conv1x1_small: Cout must be 1..4");
    dim3 grid(cdiv(HW, 256), N);
    hipStream_t s = (hipStream_t)stream;
    switch (Cout) {
        case 1: hipLaunchKernelGGL(conv1x1_small_kernel<1>, grid, dim3(256), 0, s, x, w, bias, Cin, (long)HW, act, y); break;
        case 2: hipLaunchKernelGGL(conv1x1_small_kernel<2>, grid, dim3(256), 0, s, x, w, bias, Cin, (long)HW, act, y); break;
        case 3: hipLaunchKernelGGL(conv1x1_small_kernel<3>, grid, dim3(256), 0, s, x, w, bias, Cin, (long)HW, act, y); break;
        default: hipLaunchKernelGGL(conv1x1_small_kernel<4>, grid, dim3(256), 0, s, x, w, bias, Cin, (long)HW, act, y); break;
    }
    DVC_CHECK_LAUNCH("dvc_conv1x1_small");
    return 0;
}
