// Host side of the convolution engine: geometry, tile-configuration choice, launch.
// The kernel template lives in conv_kernel.h; its instantiations are split over conv_k3d1.hip,
// conv_k3d2.hip, conv_k1.hip and conv_gen.hip.
#include "conv_kernel.h"
#include "conv_sk_kernel.h"
#include <algorithm>

#include "conv_wino_kernel.h"

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

// split-K second stage: y = act(sum_s part[s] + bias + residual), fixed summation order (deterministic).
// VEC = 4: one float4 per thread (OHW % 4 == 0 and 16-byte aligned bases), all S partial loads in flight.
template <int VEC>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ part, int S, long slab,
                                                                 int Cout, long OHW, const float* __restrict__ bias,
                                                                 const float* __restrict__ res, long res_bs,
                                                                 int act, float act_slope,
                                                                 const float* __restrict__ act_slope_ptr,
                                                                 float* __restrict__ y, long y_bs) {
    const float slope = act_slope_ptr ? *act_slope_ptr : act_slope;
    const long per_img = (long)Cout * OHW;
    const int n = blockIdx.y;
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (i >= per_img) return;
    float v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = 0.f;
    const float* p0 = part + (long)n * per_img + i;
    if (VEC == 4) {
        float4 t[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) t[s] = s < S ? *reinterpret_cast<const float4*>(p0 + (long)s * slab) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < 8; ++s) {   // S <= 8, fixed order
            v[0] += t[s].x; v[1] += t[s].y; v[2] += t[s].z; v[3] += t[s].w;
        }
    } else {
        for (int s = 0; s < S; ++s) v[0] += p0[(long)s * slab];
    }
    const float b = bias ? bias[i / OHW] : 0.f;   // a float4 never straddles channels (OHW % 4 == 0)
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        float o = v[k] + b;
        if (res) o += res[(long)n * res_bs + i + k];
        y[(long)n * y_bs + i + k] = apply_act(o, act, slope);
    }
}

// split-K second stage with the 2x2 / stride-2 max pool fused (dvc_conv2d_winograd_pool): thread = one pooling window of one
// channel = 2 rows x float2 of every slice; the sums are conv_splitk_reduce_kernel's (same order, same arithmetic), y (when not
// NULL) gets them, `pool` their maximum when the whole window is inside the image (floor mode, as nn.MaxPool2d(2, 2)).
__global__ __launch_bounds__(256) void conv_splitk_reduce_pool_kernel(const float* __restrict__ part, int S, long slab, int Cout, int OH,
                                                                      int OW, const float* __restrict__ bias, int act, float act_slope,
                                                                      const float* __restrict__ act_slope_ptr, float* __restrict__ y,
                                                                      long y_bs, float* __restrict__ pool, long pool_bs) {
    const float slope = act_slope_ptr ? *act_slope_ptr : act_slope;
    const int WH = (OH + 1) >> 1, WW = (OW + 1) >> 1;      // windows incl. the partial ones on an odd edge (they only feed y)
    const int n = blockIdx.y;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)Cout * WH * WW) return;
    const int wx = (int)(t % WW), wy = (int)((t / WW) % WH), co = (int)(t / ((long)WW * WH));
    const long OHW = (long)OH * OW, per_img = (long)Cout * OHW;
    const float b = bias ? bias[co] : 0.f;
    const bool pair = (OW & 1) == 0;                        // the window's two columns are one aligned float2
    float v[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = 2 * wy + r;
        v[r][0] = v[r][1] = 0.f;
        if (oy >= OH) continue;
        const float* p0 = part + (long)n * per_img + (long)co * OHW + (long)oy * OW + 2 * wx;
        float2 tt[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            tt[s] = make_float2(0.f, 0.f);
            if (s < S) {
                if (pair) tt[s] = *reinterpret_cast<const float2*>(p0 + (long)s * slab);
                else {
                    tt[s].x = p0[(long)s * slab];
                    if (2 * wx + 1 < OW) tt[s].y = p0[(long)s * slab + 1];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {       // S <= 8, fixed order
            v[r][0] += tt[s].x;
            v[r][1] += tt[s].y;
        }
        v[r][0] = apply_act(v[r][0] + b, act, slope);
        v[r][1] = apply_act(v[r][1] + b, act, slope);
        if (y) {
            float* yo = y + (long)n * y_bs + (long)co * OHW + (long)oy * OW + 2 * wx;
            if (pair) *reinterpret_cast<float2*>(yo) = make_float2(v[r][0], v[r][1]);
            else {
                yo[0] = v[r][0];
                if (2 * wx + 1 < OW) yo[1] = v[r][1];
            }
        }
    }
    if (2 * wy + 1 < OH && 2 * wx + 1 < OW)
        pool[(long)n * pool_bs + ((long)co * (OH >> 1) + wy) * (OW >> 1) + wx] = fmaxf(fmaxf(v[0][0], v[0][1]), fmaxf(v[1][0], v[1][1]));
}

// compute units of the current device (256 on MI355X), queried once
static int conv_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
            n = p.multiProcessorCount;
        else
            n = 256;
    }
    return n;
}

#ifdef DVC_DEBUG
static int g_conv_dbg = 0;
static long long* g_conv_dbg_buf = nullptr;
extern "C" void dvc_debug_conv_trace(long long* buf) { g_conv_dbg_buf = buf; }   // diagnostics (dvc_hip.h, last section)
extern "C" void dvc_debug_conv_variant(int v) { g_conv_dbg = v; }
#else
static constexpr int g_conv_dbg = 0;
static constexpr long long* g_conv_dbg_buf = nullptr;
#endif

bool conv_image_launch(const ConvKArgs& a, hipStream_t st);      // conv_image.hip

// One launch (plus its reduce / fixup) for the d->N images at x / y.  The PLAN — tile configuration, split over input
// channels, stream-K ranges — is always the single-image plan: an image's result never depends on what else is in the
// batch, and a batch of N is bit-identical to N single-image calls (the clip driver batches look-ahead front ends on that).
// `*group` (out): images this call covered (the split-K workspace may hold fewer than d->N at the single-image split).
static int conv2d_images(const DvcConvDesc* d, const float* x, const float* w_packed,
                         const float* bias, const float* in_scale, const float* in_shift,
                         const float* in_slope_ptr, const float* act_slope_ptr,
                         const float* residual, float* y, void* workspace, size_t workspace_bytes,
                         dvcStream stream, int* group) {
    DVC_REQUIRE(d && x && w_packed && y, "dvc_conv2d: null argument");
    *group = d->N;
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
    DVC_REQUIRE((long)d->Cin * d->H * d->W < (1L << 31) && (long)d->Cin * d->ksize * d->ksize * d->Cout < (1L << 31),
                "dvc_conv2d: tensor too large for 32-bit indexing");

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
    a.gray = (d->flags & DVC_CONV_GRAY_INPUT) ? 1 : 0;
    a.x_bs = d->x_batch_stride ? d->x_batch_stride : (long)(a.gray ? 1 : d->Cin) * d->H * d->W;
    a.y_bs = d->y_batch_stride ? d->y_batch_stride : (long)d->Cout * OH * OW;
    a.res_bs = d->res_batch_stride ? d->res_batch_stride : (long)d->Cout * OH * OW;
    a.cin_pad = (d->Cin + 3) & ~3;
    a.dbg = g_conv_dbg;
    a.dbg_buf = g_conv_dbg_buf;
    a.w_bs = d->w_batch_stride;
    DVC_REQUIRE(d->w_batch_stride >= 0 && d->w_batch_stride % 4 == 0, "dvc_conv2d: w_batch_stride must be a multiple of 4 (got %ld)",
                (long)d->w_batch_stride);
    DVC_REQUIRE(d->w_batch_stride == 0 || d->cfg < 32, "dvc_conv2d: per-image filters run on the general engine only (cfg < 32)");

    // the two image-input layers (3 -> 64, 7 -> 32) have a kernel of their own (conv_image.hip); an explicit cfg / split_k keeps
    // the general engine reachable (tests, tools/conv_algo_sweep.py)
    if (d->cfg < 0 && d->split_k == 0 && !d->w_batch_stride && !(g_conv_dbg & 1024) && conv_image_launch(a, (hipStream_t)stream)) {
        DVC_CHECK_LAUNCH("dvc_conv2d(image-input layer)");
        return 0;
    }
    DVC_REQUIRE(!a.gray, "dvc_conv2d: DVC_CONV_GRAY_INPUT is for the 3 -> 64 image-input layer under the automatic plan only");
    const int tw = pick_tw(OW);
    const int rpt = 32 / tw;
    // stride 1: geometry is baked into the kernel (one variant per ksize/dilation); stride 2 (and a
    // dilated 1x1, which nothing uses) takes the run-time-geometry variant
    const bool gen = d->stride != 1 || (d->ksize == 1 && d->dil != 1);
    auto geom = [&](int i, int* ih, int* iw) {
        const ConvCfg& c = kConvCfgs[i];
        int ph = c.wn * c.rn * rpt;
        *ih = (ph - 1) * d->stride + d->dil * (d->ksize - 1) + 1;
        *iw = (tw - 1) * d->stride + d->dil * (d->ksize - 1) + 1;
    };
    auto fits = [&](int i) -> bool {
        if (!gen) return true;
        int ih, iw;
        geom(i, &ih, &iw);
        return conv_ck(d->ksize, kConvCfgs[i].rm * kConvCfgs[i].rn, gen) * ih * iw <= CONV_EPT_GEN * 256;
    };
    if (gen) DVC_REQUIRE(d->ksize == 3, "dvc_conv2d: stride-2 / dilated variant supports ksize 3 only");
    int cfg = d->cfg;
    bool allow_dma = true;
    // Automatic choice, 3x3 layers with >= 256 input channels and no fused input transform (the 54x96 / 27x48 trunks of
    // WarpNet and ColorVidNet and the two widest decoder layers): the stream-K decomposition with 64x64 tiles and two
    // workgroups per CU wins on every one of them (profiles/r02_conv_layer_sweep.json: 62-64 us against 67-75 us).
    int sk_per_cu = d->split_k;
    if (cfg < 0 && d->split_k == 0 && !d->w_batch_stride && workspace && !gen && !in_scale && !d->in_prelu && d->ksize == 3 && d->Cin >= 256 &&
        d->Cin % 8 == 0 && d->Cout % 64 == 0 &&
        (size_t)2 * conv_num_cus() * 2 * 64 * 64 * sizeof(float) <= workspace_bytes &&
        (long)d->Cin * d->H * d->W * 4 < (1L << 31) && (long)d->Cin * 9 * d->Cout * 4 < (1L << 31)) {
        cfg = 36;
        sk_per_cu = 2;
    }
    if (cfg >= 32) {
        // 32 + tile configuration (2, 3 or 4): stream-K decomposition (conv_sk_kernel.h); split_k = workgroups per CU (0 -> 2)
        cfg -= 32;
        DVC_REQUIRE(cfg >= 2 && cfg <= 4, "dvc_conv2d: the stream-K path has tile configurations 2, 3 and 4 (cfg 34..36)");
        const ConvCfg& c = kConvCfgs[cfg];
        const int mt = 32 * c.wm * c.rm, ph = c.wn * c.rn * rpt, ck = conv_ck(d->ksize, c.rm * c.rn, false);
        DVC_REQUIRE(!gen && !in_scale && !d->in_prelu && d->Cin % ck == 0 && d->Cout % mt == 0,
                    "dvc_conv2d: the stream-K path needs a plain stride-1 layer (no fused input transform, Cin %% %d == 0, "
                    "Cout %% %d == 0)", ck, mt);
        DVC_REQUIRE(workspace, "dvc_conv2d: the stream-K path needs a workspace");
        DVC_REQUIRE((long)d->Cin * d->H * d->W * 4 < (1L << 31) && (long)d->Cin * d->ksize * d->ksize * d->Cout * 4 < (1L << 31),
                    "dvc_conv2d: tensor too large for buffer-descriptor staging");
        ConvSkArgs sk;
        sk.k = a;
        sk.k.split = 1; sk.k.chunks_per_split = 0; sk.k.part = nullptr;
        sk.tiles_x = cdiv(OW, tw);
        sk.px_tiles = sk.tiles_x * cdiv(OH, ph);
        sk.co_blocks = d->Cout / mt;
        sk.NC = d->Cin / ck;
        const long tiles = (long)sk.co_blocks * sk.px_tiles;       // of ONE image: the unit ranges are the single-image ones
        sk.U = tiles * sk.NC;
        DVC_REQUIRE(tiles < (1L << 30), "dvc_conv2d: too many tiles");
        int per_cu = sk_per_cu > 0 ? sk_per_cu : 2;
        if (per_cu > 2) per_cu = 2;     // the kernel is register-allocated for 2 workgroups per CU
        long G = (long)per_cu * conv_num_cus();
        if (G > sk.U) G = sk.U;
        const size_t slot = (size_t)mt * (c.wn * c.rn * 32) * sizeof(float);
        DVC_REQUIRE((size_t)G * 2 * slot <= workspace_bytes, "dvc_conv2d: workspace too small for the stream-K slots");
        sk.part = reinterpret_cast<float*>(workspace);
        ConvSkLaunch L;
        L.grid = dim3((unsigned)G);
        L.fix_grid = dim3((unsigned)tiles, SK_FIX_SPLIT);
        L.need_fixup = !(sk.U % G == 0 && (sk.U / G) % sk.NC == 0);
        hipStream_t st = (hipStream_t)stream;
        sk.k.N = 1;
        for (int n = 0; n < d->N; ++n) {        // one launch per image (the slots are reused: same stream)
            sk.k.x = x + (long)n * a.x_bs;
            sk.k.y = y + (long)n * a.y_bs;
            sk.k.res = residual ? residual + (long)n * a.res_bs : nullptr;
            if (d->ksize == 1) conv_sk_launch_k1(cfg, tw, L, st, sk);
            else if (d->dil == 1) conv_sk_launch_k3d1(cfg, tw, L, st, sk);
            else conv_sk_launch_k3d2(cfg, tw, L, st, sk);
            DVC_CHECK_LAUNCH("dvc_conv2d(stream-K)");
        }
        return 0;
    }
    if (cfg >= 16) {  // 16 + tile configuration: force register staging (autotuner / A-B measurements)
        cfg -= 16;
        allow_dma = false;
    }
    if (cfg < 0 && allow_dma && !gen && !in_scale && !d->in_prelu && d->Cin % conv_ck(d->ksize, 1, false) == 0 &&
        d->Cout % 32 == 0) {
        // LDS-DMA layers (profiles/r01_conv_layer_sweep.json): the one-tile-per-wave configurations win on
        // every shape of the network (3 resident workgroups per CU), the 64-channel one when Cout allows it
        cfg = d->Cout % 64 == 0 ? 4 : 3;
    }
    if (cfg < 0) {
        // Cost model fitted to the per-layer sweep (profiles/r01_conv_layer_sweep.json): a layer takes
        // (32x32 MFMA tiles per wave, padding included) x (waves per SIMD, at least one round) x a
        // staging penalty that grows as the tile shrinks (more staged bytes per MFMA).
        static const double pen[5] = {1.00, 1.06, 1.06, 1.12, 1.35};
        double best = 1e30;
        for (int i = 0; i < 5; ++i) {
            if (!fits(i)) continue;
            const ConvCfg& c = kConvCfgs[i];
            int mt = 32 * c.wm * c.rm, ph = c.wn * c.rn * rpt;
            if (d->Cout < mt && i != 1 && i != 3) continue;  // don't waste half the M tile
            double waves = 4.0 * cdiv(OW, tw) * cdiv(OH, ph) * cdiv(d->Cout, mt);     // of one image
            double rounds = waves / 1024.0;
            double p = (i == 4 && rounds <= 1.0) ? 1.10 : pen[i];
            double cost = c.rm * c.rn * (rounds < 1.0 ? 1.0 : rounds) * p;
            if (cost < best) { best = cost; cfg = i; }
        }
        DVC_REQUIRE(cfg >= 0, "dvc_conv2d: no tile configuration fits this geometry");
    }
    DVC_REQUIRE(cfg >= 0 && cfg < 5, "dvc_conv2d: cfg out of range");
    DVC_REQUIRE(fits(cfg), "dvc_conv2d: tile configuration %d does not fit this geometry", cfg);
    const ConvCfg& c = kConvCfgs[cfg];
    const int mt = 32 * c.wm * c.rm, ph = c.wn * c.rn * rpt;
    const int ck = conv_ck(d->ksize, c.rm * c.rn, gen);
    geom(cfg, &a.IH_T, &a.IW_T);
    a.IW_P = conv_pitch(tw, a.IW_T, d->stride);
    const int xs_floats = conv_xs_floats(ck, a.IH_T, a.IW_P);
    // run-time-geometry kernels carve everything from dynamic LDS; the others use static arrays
    size_t lds = gen ? sizeof(float) * (2 * (size_t)xs_floats + 2 * (size_t)conv_ws_floats(ck, a.ks * a.ks, mt) +
                                        (in_scale ? 2 * (size_t)(a.cin_pad + ck) : 0))
                     : 0;
    if (!gen && in_scale) DVC_REQUIRE(d->Cin <= CONV_MAX_AFFINE_CIN, "dvc_conv2d: fused input affine supports Cin <= %d", CONV_MAX_AFFINE_CIN);
    DVC_REQUIRE(lds <= 160 * 1024, "dvc_conv2d: LDS tile too large (%zu bytes)", lds);
    // split-K over input-channel chunks when the layer cannot put ~2 waves on every SIMD by itself
    const int nchunks = cdiv(d->Cin, ck);
    const long waves = 4L * cdiv(OW, tw) * cdiv(OH, ph) * cdiv(d->Cout, mt);      // of one image
    int S = 1;
    if (workspace && d->split_k != 1 && waves < 1536 && nchunks >= 4) {
        S = d->split_k > 1 ? d->split_k : (int)((2048 + waves - 1) / waves);
        if (d->split_k <= 1 && S > 4) S = 4;   // the static heuristic stays conservative; deeper splits are for the tuner
        if (S > 8) S = 8;
        if (S > nchunks / 2) S = nchunks / 2;
        while (S > 1 && (size_t)S * d->Cout * OH * OW * sizeof(float) > workspace_bytes) --S;
    }
    a.chunks_per_split = cdiv(nchunks, S);
    S = cdiv(nchunks, a.chunks_per_split);
    a.split = S;
    if (S > 1) {     // as many images per launch as the workspace holds partial sums for
        const size_t per_image = (size_t)S * d->Cout * OH * OW * sizeof(float);
        if ((size_t)d->N * per_image > workspace_bytes) *group = (int)(workspace_bytes / per_image);
        a.N = *group;
    }
    const int NB = a.N;
    a.part = reinterpret_cast<float*>(workspace);
    dim3 grid(cdiv(OW, tw) * cdiv(OH, ph), cdiv(d->Cout, mt), NB * S);
    hipStream_t s = (hipStream_t)stream;
    // "plain" layers stage through LDS-DMA (see conv_kernel.h)
    const bool dma = allow_dma && !gen && !in_scale && !d->in_prelu && d->Cin % ck == 0 && d->Cout % mt == 0;
    if (dma) {
        if (d->ksize == 1) conv_launch_k1_dma(cfg, tw, grid, lds, s, a);
        else if (d->dil == 1) conv_launch_k3d1_dma(cfg, tw, grid, lds, s, a);
        else conv_launch_k3d2_dma(cfg, tw, grid, lds, s, a);
    } else if (gen) conv_launch_gen(cfg, tw, grid, lds, s, a);
    else if (d->ksize == 1) conv_launch_k1(cfg, tw, grid, lds, s, a);
    else if (d->dil == 1) conv_launch_k3d1(cfg, tw, grid, lds, s, a);
    else conv_launch_k3d2(cfg, tw, grid, lds, s, a);
    DVC_CHECK_LAUNCH("dvc_conv2d");
    if (S > 1) {
        const long OHW = (long)OH * OW, per_img = (long)d->Cout * OHW;
        const bool v4 = (OHW % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.part) & 15) == 0) && (((long)NB * per_img) % 4 == 0);
        if (v4)
            hipLaunchKernelGGL(conv_splitk_reduce_kernel<4>, dim3((unsigned)((per_img / 4 + 255) / 256), NB), dim3(256), 0, s,
                               a.part, S, (long)NB * per_img, d->Cout, OHW, bias, residual, a.res_bs, d->act,
                               d->act_slope, act_slope_ptr, y, a.y_bs);
        else
            hipLaunchKernelGGL(conv_splitk_reduce_kernel<1>, dim3((unsigned)((per_img + 255) / 256), NB), dim3(256), 0, s,
                               a.part, S, (long)NB * per_img, d->Cout, OHW, bias, residual, a.res_bs, d->act,
                               d->act_slope, act_slope_ptr, y, a.y_bs);
        DVC_CHECK_LAUNCH("dvc_conv2d(split-K reduce)");
    }
    return 0;
}

extern "C" int dvc_conv2d(const DvcConvDesc* d, const float* x, const float* w_packed,
                          const float* bias, const float* in_scale, const float* in_shift,
                          const float* in_slope_ptr, const float* act_slope_ptr,
                          const float* residual, float* y, void* workspace, size_t workspace_bytes,
                          dvcStream stream) {
    DVC_REQUIRE(d, "dvc_conv2d: null descriptor");
    DvcConvDesc g = *d;
    int32_t OH = 0, OW = 0;
    dvc_conv2d_out_hw(d, &OH, &OW);
    const long x_bs = d->x_batch_stride ? d->x_batch_stride : (long)((d->flags & DVC_CONV_GRAY_INPUT) ? 1 : d->Cin) * d->H * d->W;
    const long y_bs = d->y_batch_stride ? d->y_batch_stride : (long)d->Cout * OH * OW;
    const long r_bs = d->res_batch_stride ? d->res_batch_stride : (long)d->Cout * OH * OW;
    g.x_batch_stride = x_bs; g.y_batch_stride = y_bs; g.res_batch_stride = r_bs;
    for (int n0 = 0; n0 < d->N;) {       // (one pass unless the split-K workspace is smaller than the batch needs)
        g.N = d->N - n0;
        int done = 0;
        const int rc = conv2d_images(&g, x ? x + (long)n0 * x_bs : x, w_packed ? w_packed + (long)n0 * d->w_batch_stride : w_packed, bias,
                                     in_scale ? in_scale + (long)n0 * d->Cin : nullptr,
                                     in_shift ? in_shift + (long)n0 * d->Cin : nullptr, in_slope_ptr, act_slope_ptr,
                                     residual ? residual + (long)n0 * r_bs : nullptr, y ? y + (long)n0 * y_bs : y, workspace,
                                     workspace_bytes, stream, &done);
        if (rc != 0) return rc;
        DVC_REQUIRE(done > 0, "dvc_conv2d: split-K workspace too small for one image");
        n0 += done;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) path (conv_wino_kernel.h) for 3x3 stride-1 layers without a fused input transform.
extern "C" size_t dvc_winograd_weight_floats(int32_t Cout, int32_t Cin) { return (size_t)Cout * Cin * 16; }

// U = G g G^T per (output channel, input channel), G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], evaluated in double and rounded
// once; thread = one (co, ci) pair, output layout [co / 32][ci][i][co % 32][j]
__global__ __launch_bounds__(256) void wino_pack_weight_kernel(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ u) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)Cout * Cin) return;
    const int ci = (int)(e / Cout), co = (int)(e % Cout);     // consecutive threads: consecutive co (16-byte stores side by side)
    const float* g = w + ((long)co * Cin + ci) * 9;
    double t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const double g0 = g[b], g1 = g[3 + b], g2 = g[6 + b];
        t[0][b] = g0;
        t[1][b] = 0.5 * (g0 + g1 + g2);
        t[2][b] = 0.5 * (g0 - g1 + g2);
        t[3][b] = g2;
    }
    float* dst = u + (((long)(co / 32) * Cin + ci) * 4 * 32 + (co % 32)) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double a = t[i][0], b = t[i][1], c = t[i][2];
        *reinterpret_cast<float4*>(dst + (long)i * 32 * 4) =
            make_float4((float)a, (float)(0.5 * (a + b + c)), (float)(0.5 * (a - b + c)), (float)c);
    }
}

extern "C" int dvc_winograd_pack_weight(const float* w, int32_t Cout, int32_t Cin, float* u_packed, dvcStream stream) {
    DVC_REQUIRE(w && u_packed, "dvc_winograd_pack_weight: null argument");
    DVC_REQUIRE(Cout > 0 && Cin > 0 && Cout % 32 == 0, "dvc_winograd_pack_weight: needs Cout %% 32 == 0 (got %d)", Cout);
    DVC_REQUIRE((reinterpret_cast<uintptr_t>(u_packed) & 15) == 0, "dvc_winograd_pack_weight: destination must be 16-byte aligned");
    const long n = (long)Cout * Cin;
    hipLaunchKernelGGL(wino_pack_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin,
                       u_packed);
    DVC_CHECK_LAUNCH("dvc_winograd_pack_weight");
    return 0;
}

// workgroup shapes (cfg / 4): channels = 32 wm, tile blocks = wn, input channels per LDS chunk = kc, workgroups resident per CU
// (shape 3, one wave pair per workgroup, is reachable through cfg 12..15 only: tools/conv_wino_ab.py, measured in
// profiles/r03_conv_wino_small_wg.txt; the cost model below keeps choosing among the first three)
static const struct { int wm, wn, kc, per_cu; } kWinoShapes[4] = {{4, 1, 4, 1}, {2, 2, 8, 1}, {2, 1, 4, 2}, {1, 1, 4, 4}};

// The plan of a Winograd launch: workgroup shape m, tile-block shape TR x (32/TR), split S over input-channel chunks.
// Cost in MFMA-times per SIMD: rounds of workgroups x (chunks x MFMAs per chunk + a per-workgroup prologue/epilogue) + a
// reduce launch.  A pure function of the descriptor and the workspace size (dvc_conv2d_winograd_split exposes S).
static void wino_plan(const DvcConvDesc* d, int OH, int OW, bool have_workspace, size_t workspace_bytes, int* best_m_out,
                      int* best_tr_out, int* best_S_out, int force_m = -1) {
    const int ss = d->dil;
    const int TY = cdiv(cdiv(OH, ss), 2), TX = cdiv(cdiv(OW, ss), 2);
    const int ncu = conv_num_cus();
    static const int kTR[4] = {1, 2, 4, 8};
    int best_m = -1, best_tr = 1, best_S = 1;
    double best_cost = 1e30;
    const int shape_cfg = d->cfg >= 0 ? d->cfg : -1;
    for (int m = 0; m < 4; ++m) {
        const int wm = kWinoShapes[m].wm, wn = kWinoShapes[m].wn, kc = kWinoShapes[m].kc;
        if (d->Cout % (32 * wm) != 0) continue;
        if (shape_cfg >= 0 && shape_cfg / 4 != m) continue;
        if (shape_cfg < 0 && m == 3) continue;
        if (force_m >= 0 && m != force_m) continue;
        const int nch = d->Cin / kc;
        for (int ti = 0; ti < 4; ++ti) {
            const int tr = kTR[ti];
            if (shape_cfg >= 0 && shape_cfg % 4 != ti) continue;
            // (workgroups of ONE image: the plan never depends on the batch size, see conv2d_images — unless the caller asks
            // for a plan of the whole batch, DVC_CONV_BATCH_PLAN: the R references of a clip run in lock step anyway)
            const long wgs = (long)ss * ss * cdiv(TY, tr * wn) * cdiv(TX, 32 / tr) * (d->Cout / (32 * wm)) *
                             ((d->flags & DVC_CONV_BATCH_PLAN) ? d->N : 1);
            for (int S = 1; S <= 8; ++S) {
                if (d->split_k > 0 && S != d->split_k) continue;
                if (S > 1 && (!have_workspace || S > nch / 2 ||
                              (size_t)S * d->Cout * OH * OW * sizeof(float) * ((d->flags & DVC_CONV_BATCH_PLAN) ? d->N : 1) >
                                  workspace_bytes)) continue;
                const int cps = cdiv(nch, S);
                if (cdiv(nch, cps) != S) continue;
                const double rounds = (double)cdivl(wgs * S, (long)ncu * kWinoShapes[m].per_cu);
                // (per-workgroup prologue / epilogue: partly hidden behind the sibling workgroup where two share a CU)
                double cost = rounds * (cps * (kc / 2) * 16.0 + (kWinoShapes[m].per_cu == 2 ? 130.0 : 160.0)) + (S > 1 ? 200.0 : 0.0);
                cost *= 1.0 + 0.02 * ti;     // wider tile rows store better
                // tile slots that hang over the edge of the image are computed all the same, and every extra workgroup
                // streams its filter slice again: e.g. 1x32-tile blocks on a 7x12-tile map (13x24 outputs) are 37 % full
                const double fill = (double)TY * TX / ((double)cdiv(TY, tr * wn) * (tr * wn) * cdiv(TX, 32 / tr) * (32 / tr));
                cost *= 1.0 + 0.3 * (1.0 - fill);
                if (cost < best_cost) { best_cost = cost; best_m = m; best_tr = tr; best_S = S; }
            }
        }
    }
#ifdef DVC_DEBUG
    // dvc_debug_conv_variant(64): twice the split (shorter-lived workgroups) — tools/bg_split_probe.py asks whether front-end
    // launches that run as background work delay the high-priority chain less that way
    if ((g_conv_dbg & 64) && d->split_k == 0 && have_workspace && best_m >= 0) {
        const int kc = kWinoShapes[best_m].kc, nch = d->Cin / kc;
        int S2 = best_S * 2;
        while (S2 > best_S && (S2 > 8 || S2 > nch / 2 || (size_t)S2 * d->Cout * OH * OW * sizeof(float) > workspace_bytes ||
                               cdiv(nch, cdiv(nch, S2)) != S2)) --S2;
        best_S = S2;
    }
    // dvc_debug_conv_variant(128 / 256): half the split / no split (fewer partial sums, longer workgroups, chip under-filled by
    // this launch alone) — the same probe asks whether the multi-stream driver prefers that
    // (8192: half the split only for launches that carry a batch — a stand-in for batch-aware planning, tools/bench_variant.py
    // --front-batch 2)
    if (((g_conv_dbg & (128 | 256)) || ((g_conv_dbg & 8192) && d->N >= 2)) && d->split_k == 0 && best_m >= 0 && best_S > 1) {
        const int kc = kWinoShapes[best_m].kc, nch = d->Cin / kc;
        int S2 = (g_conv_dbg & 256) ? 1 : best_S / 2;
        while (S2 > 1 && cdiv(nch, cdiv(nch, S2)) != S2) --S2;
        best_S = S2 < 1 ? 1 : S2;
    }
#endif
    *best_m_out = best_m; *best_tr_out = best_tr; *best_S_out = best_S;
}

static int wino_check_desc(const DvcConvDesc* d) {
    DVC_REQUIRE(d->ksize == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) && d->pad == d->dil,
                "dvc_conv2d_winograd: needs a 3x3 stride-1 layer with pad == dilation (1 or 2)");
    DVC_REQUIRE(!d->in_prelu, "dvc_conv2d_winograd: no fused input transform on this path");
    DVC_REQUIRE((d->in_up == 1 || d->in_up == 2) && (d->in_sub == 1 || d->in_sub == 2) && !(d->in_up == 2 && d->in_sub == 2),
                "dvc_conv2d_winograd: bad in_up/in_sub");
    DVC_REQUIRE(d->N > 0 && d->Cin > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "dvc_conv2d_winograd: bad shape");
    DVC_REQUIRE(d->Cin % 8 == 0 && d->Cout % 64 == 0, "dvc_conv2d_winograd: needs Cin %% 8 == 0 and Cout %% 64 == 0 (got %d, %d)",
                d->Cin, d->Cout);
    return 0;
}

// Images one Winograd launch covers: all of them, unless the workspace holds the partial outputs of fewer at this
// (single-image) split, or the 1-D grid would pass the 65535 workgroups the kernel's 16-bit reciprocal index decode
// handles.  -1: a single image already needs more workgroups than that.
static int wino_images_per_launch(const DvcConvDesc* d, int OH, int OW, int m, int tr, int split, size_t workspace_bytes) {
    const int ss = d->dil, wm = kWinoShapes[m].wm, wn = kWinoShapes[m].wn;
    const int TY = cdiv(cdiv(OH, ss), 2), TX = cdiv(cdiv(OW, ss), 2);
    const long gx = (long)ss * ss * cdiv(TY, tr * wn) * cdiv(TX, 32 / tr), gy = d->Cout / (32 * wm);
    const size_t per_img = (size_t)d->Cout * OH * OW * sizeof(float);
    long group = d->N;
    if (split > 1 && (size_t)d->N * split * per_img > workspace_bytes) group = (long)(workspace_bytes / ((size_t)split * per_img));
    const long per_image = gx * gy * split;
    if (per_image >= 65536) return -1;
    if (group * per_image >= 65536) group = 65535 / per_image;
    return (int)group;
}

extern "C" int dvc_conv2d_winograd_split(const DvcConvDesc* d, size_t workspace_bytes, int32_t* split,
                                         int32_t* images_per_launch) {
    DVC_REQUIRE(d && split && images_per_launch, "dvc_conv2d_winograd_split: null argument");
    if (int rc = wino_check_desc(d)) return rc;
    int32_t OH, OW;
    dvc_conv2d_out_hw(d, &OH, &OW);
    DVC_REQUIRE(OH > 0 && OW > 0, "dvc_conv2d_winograd_split: empty output");
    int m, tr, S;
    wino_plan(d, OH, OW, workspace_bytes > 0, workspace_bytes, &m, &tr, &S);
    DVC_REQUIRE(m >= 0, "dvc_conv2d_winograd_split: no configuration for cfg %d / split_k %d on this layer", d->cfg, d->split_k);
    // (the launch recomputes the split from the chunk count: same arithmetic as dvc_conv2d_winograd)
    const int kc = kWinoShapes[m].kc, nch = d->Cin / kc;
    *split = cdiv(nch, cdiv(nch, S));
    *images_per_launch = wino_images_per_launch(d, OH, OW, m, tr, *split, workspace_bytes);
    return 0;
}

// d2 / x2 != NULL: the two-input form (dvc_conv2d_winograd_dual) — `d` then carries the TOTAL channel count and the first
// input's geometry, `d2` the second input's (Cin, H, W, in_up, in_sub).
// Everything of a Winograd launch but the launch: argument checks, the kernel's argument block, the plan.  `*m_out` /
// `*tr_out`: workgroup shape and tile-block shape; `*group_out`: images one launch covers.
static int wino_setup(const DvcConvDesc* d, const DvcConvDesc* d2, const float* x, const float* x2, const float* u_packed,
                      const float* bias, const float* act_slope_ptr, const float* residual, float* y, void* workspace,
                      size_t workspace_bytes, float* pool, long pool_batch_stride, ConvWinoArgs& s, int* m_out, int* tr_out,
                      int* group_out) {
    DVC_REQUIRE(d && x && u_packed && (y || pool), "dvc_conv2d_winograd: null argument");
    DVC_REQUIRE(d->ksize == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) && d->pad == d->dil,
                "dvc_conv2d_winograd: needs a 3x3 stride-1 layer with pad == dilation (1 or 2)");
    DVC_REQUIRE(!d->in_prelu, "dvc_conv2d_winograd: no fused input transform on this path");
    DVC_REQUIRE((d->in_up == 1 || d->in_up == 2) && (d->in_sub == 1 || d->in_sub == 2) && !(d->in_up == 2 && d->in_sub == 2),
                "dvc_conv2d_winograd: bad in_up/in_sub");
    DVC_REQUIRE(d->N > 0 && d->Cin > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "dvc_conv2d_winograd: bad shape");
    DVC_REQUIRE(d->Cin % 8 == 0 && d->Cout % 64 == 0, "dvc_conv2d_winograd: needs Cin %% 8 == 0 and Cout %% 64 == 0 (got %d, %d)",
                d->Cin, d->Cout);
    DVC_REQUIRE((reinterpret_cast<uintptr_t>(u_packed) & 15) == 0, "dvc_conv2d_winograd: weights must be 16-byte aligned");
    DVC_REQUIRE((long)d->Cin * d->H * d->W * 4 < (1L << 31) && (long)d->Cin * 2048 * 4 < (1L << 31),
                "dvc_conv2d_winograd: tensor too large for buffer-descriptor staging");
    s.x2 = nullptr; s.x2_bs = 0; s.cinA = d->Cin; s.H2 = s.W2 = s.VH2 = s.VW2 = 0; s.in_up2 = s.in_sub2 = 1;
    ConvKArgs& a = s.k;
    a.x = x; a.w = u_packed; a.bias = bias; a.in_scale = nullptr; a.in_shift = nullptr;
    a.in_slope_ptr = nullptr; a.act_slope_ptr = act_slope_ptr; a.res = residual; a.y = y;
    a.N = d->N; a.Cin = d->Cin; a.H = d->H; a.W = d->W;
    a.VH = virt_dim(d->H, d->in_up, d->in_sub);
    a.VW = virt_dim(d->W, d->in_up, d->in_sub);
    int32_t OH, OW;
    dvc_conv2d_out_hw(d, &OH, &OW);
    DVC_REQUIRE(OH > 0 && OW > 0, "dvc_conv2d_winograd: empty output");
    if (d->pad_mode == DVC_PAD_REFLECT)
        DVC_REQUIRE(d->pad < a.VH && d->pad < a.VW, "dvc_conv2d_winograd: reflect pad needs pad < input size");
    a.Cout = d->Cout; a.OH = OH; a.OW = OW;
    a.ks = 3; a.stride = 1; a.dil = d->dil; a.pad = d->pad; a.pad_mode = d->pad_mode;
    a.in_up = d->in_up; a.in_sub = d->in_sub; a.act = d->act; a.in_prelu = 0;
    a.act_slope = d->act_slope;
    a.x_bs = d->x_batch_stride ? d->x_batch_stride : (long)d->Cin * d->H * d->W;
    a.y_bs = d->y_batch_stride ? d->y_batch_stride : (long)d->Cout * OH * OW;
    a.res_bs = d->res_batch_stride ? d->res_batch_stride : (long)d->Cout * OH * OW;
    a.cin_pad = d->Cin; a.IH_T = a.IW_T = a.IW_P = 0;
    a.dbg = g_conv_dbg; a.dbg_buf = nullptr; a.w_bs = 0; a.gray = 0;
    s.ss = d->dil;
    const int TY = cdiv(cdiv(OH, s.ss), 2), TX = cdiv(cdiv(OW, s.ss), 2);   // 2x2 tiles of one parity class
    int best_m = -1, best_tr = 1, best_S = 1;
    if (d2) {
        s.x2 = x2;
        s.cinA = d->Cin - d2->Cin;
        s.H2 = d2->H; s.W2 = d2->W;
        s.VH2 = virt_dim(d2->H, d2->in_up, d2->in_sub);
        s.VW2 = virt_dim(d2->W, d2->in_up, d2->in_sub);
        s.in_up2 = d2->in_up; s.in_sub2 = d2->in_sub;
        s.x2_bs = d2->x_batch_stride ? d2->x_batch_stride : (long)d2->Cin * d2->H * d2->W;
        DVC_REQUIRE(s.VH2 == a.VH && s.VW2 == a.VW, "dvc_conv2d_winograd_dual: the two inputs' virtual sizes differ (%d x %d vs %d x %d)",
                    a.VH, a.VW, s.VH2, s.VW2);
        DVC_REQUIRE((long)d2->Cin * d2->H * d2->W * 4 < (1L << 31), "dvc_conv2d_winograd_dual: second input too large");
        // the first input's rsrc covers only its own channels
        a.x_bs = d->x_batch_stride ? d->x_batch_stride : (long)s.cinA * d->H * d->W;
    }
    wino_plan(d, OH, OW, workspace != nullptr, workspace_bytes, &best_m, &best_tr, &best_S, d2 ? 2 : -1);
    DVC_REQUIRE(best_m >= 0, "dvc_conv2d_winograd: no configuration for cfg %d / split_k %d on this layer", d->cfg, d->split_k);
    const int wm = kWinoShapes[best_m].wm, wn = kWinoShapes[best_m].wn, kc = kWinoShapes[best_m].kc;
    const int nch = d->Cin / kc;
    s.blk_y = cdiv(TY, best_tr * wn);
    s.blk_x = cdiv(TX, 32 / best_tr);
    a.chunks_per_split = cdiv(nch, best_S);
    a.split = cdiv(nch, a.chunks_per_split);
    a.part = reinterpret_cast<float*>(workspace);
    s.gx = s.ss * s.ss * s.blk_y * s.blk_x;
    s.gy = d->Cout / (32 * wm);
    // images per launch: all of them, unless the workspace holds the partial outputs of fewer at this (single-image) split
    const int group = wino_images_per_launch(d, OH, OW, best_m, best_tr, a.split, workspace_bytes);
    DVC_REQUIRE(group >= 0, "dvc_conv2d_winograd: %ld workgroups per image (feature map too large for this path)",
                (long)s.gx * s.gy * a.split);
    DVC_REQUIRE(group > 0, "dvc_conv2d_winograd: split-K workspace too small for one image");
    DVC_REQUIRE(!(d->flags & DVC_CONV_DEFER_REDUCE) || a.split == 1 || group >= d->N,
                "dvc_conv2d_winograd: DVC_CONV_DEFER_REDUCE needs a workspace that holds the partial sums of the whole batch");
    DVC_REQUIRE(!(d->flags & DVC_CONV_DEFER_REDUCE) || a.split == 1 || !residual,
                "dvc_conv2d_winograd: DVC_CONV_DEFER_REDUCE does not carry a residual");
    s.pool_bs = pool_batch_stride ? pool_batch_stride : (long)d->Cout * (OH / 2) * (OW / 2);
    s.pool = pool;
    *m_out = best_m; *tr_out = best_tr; *group_out = group;
    return 0;
}

// the decode reciprocals and the grid of a launch over `NB` images (s.k.N etc. already set)
static dim3 wino_finish_grid(ConvWinoArgs& s, int NB) {
    s.gz = NB * s.k.split;
    s.m_gxy = wino_magic((long)s.gx * s.gy);
    s.m_gx = wino_magic(s.gx);
    s.m_cls = wino_magic((long)s.blk_y * s.blk_x);
    s.m_blkx = wino_magic(s.blk_x);
    s.m_split = wino_magic(s.k.split);
    return dim3((unsigned)(s.gx * s.gy * s.gz));      // 1-D: the kernel maps it XCD-aware onto (gx, gy, gz)
}

// the launch(es) that follow a split Winograd launch over `NB` images: split-K reduce (+ pool), unless deferred
static int wino_reduce(const DvcConvDesc* d, const ConvWinoArgs& s, int NB, int OH, int OW, const float* bias,
                       const float* act_slope_ptr, hipStream_t st) {
    const ConvKArgs& a = s.k;
    const long OHW = (long)OH * OW, per_img = (long)d->Cout * OHW;
    if (a.split > 1 && s.pool) {
        const long windows = (long)d->Cout * ((OH + 1) / 2) * ((OW + 1) / 2);
        hipLaunchKernelGGL(conv_splitk_reduce_pool_kernel, dim3((unsigned)((windows + 255) / 256), NB), dim3(256), 0, st, a.part,
                           a.split, (long)NB * per_img, d->Cout, OH, OW, bias, d->act, d->act_slope, act_slope_ptr, a.y, a.y_bs,
                           s.pool, s.pool_bs);
        DVC_CHECK_LAUNCH("dvc_conv2d_winograd_pool(split-K reduce)");
    } else if (a.split > 1 && !(d->flags & DVC_CONV_DEFER_REDUCE)) {
        const bool v4 = (OHW % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.part) & 15) == 0) && (((long)NB * per_img) % 4 == 0);
        if (v4)
            hipLaunchKernelGGL(conv_splitk_reduce_kernel<4>, dim3((unsigned)((per_img / 4 + 255) / 256), NB), dim3(256), 0, st,
                               a.part, a.split, (long)NB * per_img, d->Cout, OHW, bias, a.res, a.res_bs, d->act,
                               d->act_slope, act_slope_ptr, a.y, a.y_bs);
        else
            hipLaunchKernelGGL(conv_splitk_reduce_kernel<1>, dim3((unsigned)((per_img + 255) / 256), NB), dim3(256), 0, st,
                               a.part, a.split, (long)NB * per_img, d->Cout, OHW, bias, a.res, a.res_bs, d->act,
                               d->act_slope, act_slope_ptr, a.y, a.y_bs);
        DVC_CHECK_LAUNCH("dvc_conv2d_winograd(split-K reduce)");
    }
    return 0;
}

static int wino_run(const DvcConvDesc* d, const DvcConvDesc* d2, const float* x, const float* x2, const float* u_packed,
                    const float* bias, const float* act_slope_ptr, const float* residual, float* y, void* workspace,
                    size_t workspace_bytes, dvcStream stream, float* pool = nullptr, long pool_batch_stride = 0) {
    ConvWinoArgs s;
    int best_m = -1, best_tr = 1, group = 0;
    if (int rc = wino_setup(d, d2, x, x2, u_packed, bias, act_slope_ptr, residual, y, workspace, workspace_bytes, pool,
                            pool_batch_stride, s, &best_m, &best_tr, &group))
        return rc;
    ConvKArgs& a = s.k;
    int32_t OH, OW;
    dvc_conv2d_out_hw(d, &OH, &OW);
    hipStream_t st = (hipStream_t)stream;
    for (int n0 = 0; n0 < d->N; n0 += group) {
        const int NB = std::min(group, d->N - n0);
        a.N = NB;
        a.x = x + (long)n0 * a.x_bs;
        if (d2) s.x2 = x2 + (long)n0 * s.x2_bs;
        a.y = y ? y + (long)n0 * a.y_bs : nullptr;
        s.pool = pool ? pool + (long)n0 * s.pool_bs : nullptr;
        a.res = residual ? residual + (long)n0 * a.res_bs : nullptr;
        DVC_REQUIRE((long)s.gx * s.gy * NB * a.split < (1L << 31), "dvc_conv2d_winograd: grid too large");
        const dim3 grid = wino_finish_grid(s, NB);
        if (d2) conv_wino_launch_m1_dual(best_tr, grid, st, s);
        else if (best_m == 0) conv_wino_launch_m4(best_tr, grid, st, s);
        else if (best_m == 1) conv_wino_launch_m2(best_tr, grid, st, s);
        else if (best_m == 2) conv_wino_launch_m1(best_tr, grid, st, s);
        else conv_wino_launch_m0(best_tr, grid, st, s);
        DVC_CHECK_LAUNCH("dvc_conv2d_winograd");
        if (int rc = wino_reduce(d, s, NB, OH, OW, bias, act_slope_ptr, st)) return rc;
    }
    return 0;
}

// Several independent Winograd layers as ONE launch (conv_wino_group_kernel).  Every item is planned exactly as
// dvc_conv2d_winograd plans it alone; when all of them get the 64-channel x 32-tile workgroup shape and one launch each, their
// workgroups form one grid (longest workgroups first), followed by the items' own reduce launches where a split item does not
// defer its reduce.  Otherwise (another shape, a batch that needs several launches): one launch per item, as if called apart.
extern "C" int dvc_conv2d_winograd_group(const DvcConvGroupItem* items, int32_t n_items, dvcStream stream) {
    DVC_REQUIRE(items && n_items >= 1 && n_items <= WINO_GROUP_MAX, "dvc_conv2d_winograd_group: 1..%d items", WINO_GROUP_MAX);
    hipStream_t st = (hipStream_t)stream;
    ConvWinoGroupArgs g;
    int OHs[WINO_GROUP_MAX], OWs[WINO_GROUP_MAX], order[WINO_GROUP_MAX];
    ConvWinoArgs tmp[WINO_GROUP_MAX];
    int trs[WINO_GROUP_MAX];
    bool together = n_items > 1;
    for (int i = 0; i < n_items; ++i) {
        const DvcConvGroupItem& it = items[i];
        DVC_REQUIRE(it.x && it.u_packed && it.y, "dvc_conv2d_winograd_group: null argument in item %d", i);
        for (int j = 0; j < i; ++j) DVC_REQUIRE(it.y != items[j].y, "dvc_conv2d_winograd_group: items %d and %d share an output", j, i);
        int m = -1, grp = 0;
        if (int rc = wino_setup(&it.d, nullptr, it.x, nullptr, it.u_packed, it.bias, it.act_slope_ptr, it.residual, it.y, it.workspace,
                                it.workspace_bytes, nullptr, 0, tmp[i], &m, &trs[i], &grp))
            return rc;
        int32_t oh, ow;
        dvc_conv2d_out_hw(&it.d, &oh, &ow);
        OHs[i] = oh; OWs[i] = ow;
        if (m != 2 || grp < it.d.N) together = false;
        for (int j = 0; j < i; ++j)     // (only split items touch their workspace)
            DVC_REQUIRE(!(tmp[i].k.split > 1 && tmp[j].k.split > 1 && tmp[i].k.part == tmp[j].k.part),
                        "dvc_conv2d_winograd_group: the split items %d and %d share a workspace", j, i);
        order[i] = i;
    }
    if (!together) {
        for (int i = 0; i < n_items; ++i) {
            const DvcConvGroupItem& it = items[i];
            if (int rc = wino_run(&it.d, nullptr, it.x, nullptr, it.u_packed, it.bias, it.act_slope_ptr, it.residual, it.y, it.workspace,
                                  it.workspace_bytes, stream))
                return rc;
        }
        return 0;
    }
    // longest workgroups first (chunks per split): the short ones fill the tail
    std::sort(order, order + n_items, [&](int a, int b) {
        return tmp[a].k.chunks_per_split != tmp[b].k.chunks_per_split ? tmp[a].k.chunks_per_split > tmp[b].k.chunks_per_split : a < b;
    });
    g.n = n_items;
    g.per_xcd[0] = 0;
    for (int k = 0; k < WINO_GROUP_MAX; ++k) {
        const int i = order[k < n_items ? k : n_items - 1];
        g.item[k] = tmp[i];
        g.tr[k] = trs[i];
        const dim3 grid = wino_finish_grid(g.item[k], items[i].d.N);
        g.per_xcd[k + 1] = g.per_xcd[k] + (k < n_items ? cdiv((int)grid.x, 8) : 0);
    }
    DVC_REQUIRE((long)g.per_xcd[n_items] * 8 < (1L << 31), "dvc_conv2d_winograd_group: grid too large");
    conv_wino_launch_group_m1(dim3((unsigned)(g.per_xcd[n_items] * 8)), st, g);
    DVC_CHECK_LAUNCH("dvc_conv2d_winograd_group");
    for (int k = 0; k < n_items; ++k) {
        const int i = order[k];
        if (int rc = wino_reduce(&items[i].d, g.item[k], items[i].d.N, OHs[i], OWs[i], items[i].bias, items[i].act_slope_ptr, st)) return rc;
    }
    return 0;
}

extern "C" int dvc_conv2d_winograd(const DvcConvDesc* d, const float* x, const float* u_packed, const float* bias,
                                   const float* act_slope_ptr, const float* residual, float* y, void* workspace,
                                   size_t workspace_bytes, dvcStream stream) {
    return wino_run(d, nullptr, x, nullptr, u_packed, bias, act_slope_ptr, residual, y, workspace, workspace_bytes, stream);
}

extern "C" int dvc_conv2d_winograd_pool(const DvcConvDesc* d, const float* x, const float* u_packed, const float* bias,
                                        const float* act_slope_ptr, float* y, float* y_pool, int64_t pool_batch_stride,
                                        void* workspace, size_t workspace_bytes, dvcStream stream) {
    DVC_REQUIRE(d && y_pool, "dvc_conv2d_winograd_pool: null argument");
    DVC_REQUIRE(d->dil == 1 && !(d->flags & DVC_CONV_DEFER_REDUCE), "dvc_conv2d_winograd_pool: dilation 1, no deferred reduce");
    int32_t OH = 0, OW = 0;
    dvc_conv2d_out_hw(d, &OH, &OW);
    DVC_REQUIRE(OH >= 2 && OW >= 2, "dvc_conv2d_winograd_pool: output smaller than a pooling window");
    return wino_run(d, nullptr, x, nullptr, u_packed, bias, act_slope_ptr, nullptr, y, workspace, workspace_bytes, stream, y_pool,
                    pool_batch_stride);
}

extern "C" int dvc_conv2d_winograd_dual(const DvcConvDesc* dA, const DvcConvDesc* dB, const float* xA, const float* xB,
                                        const float* u_packed_cat, const float* bias, const float* act_slope_ptr, float* y,
                                        void* workspace, size_t workspace_bytes, dvcStream stream) {
    DVC_REQUIRE(dA && dB && xA && xB && u_packed_cat && y, "dvc_conv2d_winograd_dual: null argument");
    DVC_REQUIRE(dA->N == dB->N && dA->Cout == dB->Cout && dA->dil == dB->dil && dA->pad_mode == dB->pad_mode && dB->ksize == 3 &&
                    dB->stride == 1 && dB->pad == dB->dil,
                "dvc_conv2d_winograd_dual: the two convolutions must agree in batch, output channels, dilation and padding");
    DVC_REQUIRE(dA->Cin % 8 == 0 && dB->Cin % 8 == 0, "dvc_conv2d_winograd_dual: both channel counts must be multiples of 8");
    DVC_REQUIRE((dB->in_up == 1 || dB->in_up == 2) && (dB->in_sub == 1 || dB->in_sub == 2) && !(dB->in_up == 2 && dB->in_sub == 2),
                "dvc_conv2d_winograd_dual: bad in_up/in_sub of the second input");
    DVC_REQUIRE(!(dA->flags & DVC_CONV_DEFER_REDUCE), "dvc_conv2d_winograd_dual: no deferred reduce on this entry");
    DvcConvDesc d = *dA;
    d.Cin = dA->Cin + dB->Cin;
    d.x_batch_stride = dA->x_batch_stride ? dA->x_batch_stride : (int64_t)dA->Cin * dA->H * dA->W;
    return wino_run(&d, dB, xA, xB, u_packed_cat, bias, act_slope_ptr, nullptr, y, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// 1x1 conv with Cout <= 4 (ColorVidNet.conv10_ab: 128 -> 2, then tanh*128).  Pure bandwidth (42 MB in, 0.7 MB
// out at 216x384): a workgroup = 64 consecutive pixels x 4 channel groups, so that 4x more coalesced loads are
// in flight than with one thread per pixel; the four partial sums are added in a fixed order (deterministic).
template <int COUT>
__global__ __launch_bounds__(256) void conv1x1_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, int Cin, long HW, int act,
                                                            float* __restrict__ y) {
    __shared__ float part[4][COUT][64];
    const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long p = (long)blockIdx.x * 64 + px;
    const int n = blockIdx.y;
    const bool ok = p < HW;
    const float* xn = x + (long)n * Cin * HW + (ok ? p : 0);
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    const int c0 = (Cin * g) / 4, c1 = (Cin * (g + 1)) / 4;
    for (int c = c0; c < c1; ++c) {
        const float v = xn[(long)c * HW];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = fmaf(v, w[o * Cin + c], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) part[g][o][px] = acc[o];
    __syncthreads();
    if (g != 0 || !ok) return;
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
        float v = ((part[0][o][px] + part[1][o][px]) + (part[2][o][px] + part[3][o][px])) + (bias ? bias[o] : 0.f);
        y[((long)n * COUT + o) * HW + p] = apply_act(v, act, 0.f);
    }
}

extern "C" int dvc_conv1x1_small(const float* x, const float* w, const float* bias, int32_t N,
                                 int32_t Cin, int32_t HW, int32_t Cout, int32_t act, float* y,
                                 dvcStream stream) {
    DVC_REQUIRE(x && w && y, "dvc_conv1x1_small: null argument");
    DVC_REQUIRE(Cout >= 1 && Cout <= 4, "dvc_conv1x1_small: Cout must be 1..4");
    dim3 grid(cdiv(HW, 64), N);
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
