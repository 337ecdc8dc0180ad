/*
 * dvc_hip.h — C-ABI of libdvc_hip.so: hand-written gfx950 (MI355X / CDNA4) HIP kernels for the
 * inference hot path of Deep-Exemplar-based-Video-Colorization
 *   VGG19 features -> WarpNet dense exemplar<->frame correlation -> ColorVidNet generator.
 *
 * The reference (/root/reference) has NO native / FFI layer: every op on this path is a stock ATen
 * op reached through torch.nn.  The drop-in boundary is therefore the Python module
 * surface that /root/reference/test.py:17-19 imports; this header is the thin native layer underneath
 * it, and each entry point cites the reference call site(s) whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C: raw device pointers (fp32, NCHW, contiguous unless a batch stride is given), sizes,
 *     and a hipStream_t passed as void*.  No torch types.  No allocation, no synchronisation: every
 *     call only enqueues kernels on `stream` (so a caller may capture calls into a hipGraph).
 *   - return 0 on success; non-zero on error, with a message retrievable via dvc_last_error().
 *   - scratch memory is supplied by the caller (sizes via the *_workspace_bytes helpers).
 */
#ifndef DVC_HIP_H
#define DVC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the entry points declared in this header are exported. */
#pragma GCC visibility push(default)

typedef void* dvcStream; /* hipStream_t */

#define DVC_ABI_VERSION 19

int dvc_abi_version(void);
/* Thread-local description of the last failure (empty string if none). */
const char* dvc_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Convolution engine: im2col-free implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).
 * Replaces every nn.Conv2d (+ the padding / upsample / norm / activation modules wrapped around it):
 *   VGG19_pytorch.forward            models/NonlocalNet.py:235-254  (conv3x3 p1 + ReLU)
 *   WarpNet heads / ResidualBlock    models/NonlocalNet.py:364-410, 341-352
 *                                    (ReflectionPad2d + conv3x3 s1/s2; IN+PReLU folded into the
 *                                     consumer's load; Upsample folded into the index map)
 *   WarpNet.theta / .phi             models/NonlocalNet.py:418-423  (1x1)
 *   ColorVidNet.forward              models/ColorVidNet.py:98-141   (conv3x3 d1/d2, InstanceNorm and
 *                                     the depthwise stride-2 `*_ss` scale folded into the load,
 *                                     nearest-up folded into the index map, skip-add + ReLU epilogue)
 *
 * y[n,co,oy,ox] = act( bias[co] + residual[n,co,oy,ox]
 *                      + sum_{ci,ky,kx} w[co,ci,ky,kx] * T(x)[n,ci, oy*stride+ky*dil-pad, ox*stride+kx*dil-pad] )
 * where T(x) is the *virtual* input: the stored tensor x[N,Cin,H,W], optionally subsampled
 * (x[:, :, ::2, ::2]) or nearest-upsampled x2, with the per-(n,ci) affine v*in_scale+in_shift and an
 * optional PReLU applied to in-bounds values; out-of-bounds taps read 0 (zero pad) or the mirrored
 * position (reflect pad, applied in the virtual domain exactly like ReflectionPad2d after Upsample).
 *
 * Weights are pre-packed as w_packed[ci][ky*ks+kx][co] (co contiguous), Cout % 4 == 0.
 *
 * Layers too small to fill the 1024 SIMDs are split over input-channel chunks (split-K): partial sums
 * go to `workspace` ([S][N][Cout][OH][OW] floats) and a second tiny kernel adds them in a fixed order and
 * applies bias / residual / activation.  Without a workspace split-K is off.
 * Alternatively (cfg 32 + k) the layer is decomposed stream-K style: the (tile, channel-chunk) units are dealt in
 * equal contiguous ranges to a grid that exactly fills the chip; tiles that a range boundary splits go through
 * per-workgroup slots in `workspace` and a fixed-order fixup kernel (csrc/conv_sk_kernel.h).
 */
enum { DVC_ACT_NONE = 0, DVC_ACT_RELU = 1, DVC_ACT_PRELU = 2, DVC_ACT_LEAKY = 3, DVC_ACT_TANH128 = 4 };
enum { DVC_PAD_ZERO = 0, DVC_PAD_REFLECT = 1 };

typedef struct DvcConvDesc {
    int32_t N, Cin, H, W;       /* stored input tensor */
    int32_t Cout;
    int32_t ksize;              /* 1 or 3 */
    int32_t stride;             /* 1 or 2 */
    int32_t dil;                /* 1 or 2 */
    int32_t pad;                /* 0..2 */
    int32_t pad_mode;           /* DVC_PAD_* */
    int32_t in_up;              /* 1 | 2 : nearest upsample of the stored input */
    int32_t in_sub;             /* 1 | 2 : keep every 2nd row/col of the stored input */
    int32_t act;                /* DVC_ACT_* */
    float   act_slope;          /* PReLU/LeakyReLU slope when act_slope_ptr == NULL */
    int32_t in_prelu;           /* apply PReLU (slope *in_slope_ptr) to the affine-transformed input */
    int32_t cfg;                /* tile configuration 0..4; -1 = choose automatically; 16 + k = configuration k
                                   with register staging forced (layers without a fused input transform
                                   otherwise stage through LDS-DMA); 32 + k (k = 2, 3, 4) = stream-K decomposition
                                   of a plain stride-1 layer with tile configuration k (needs a workspace) */
    int32_t split_k;            /* 0 = automatic, 1 = off, 2..8 = forced (needs a workspace); with cfg >= 32:
                                   workgroups per CU (0 -> 2) */
    int64_t x_batch_stride;     /* elements; 0 => Cin*H*W */
    int64_t y_batch_stride;     /* elements; 0 => Cout*OH*OW  (lets y be a channel slice) */
    int64_t res_batch_stride;   /* elements; 0 => Cout*OH*OW */
    int32_t flags;              /* DVC_CONV_* bits (0 = none) */
    int64_t w_batch_stride;     /* dvc_conv2d only, elements; 0 => the N images share one filter set (a convolution); != 0 =>
                                   image n uses w_packed + n * w_batch_stride: with ksize 1 that is a BATCHED GEMM
                                   y[n] = w[n]^T x[n] (K = Cin, M = Cout, N = H*W) — the N x N affinity products of the
                                   training-side callers (models/ContextualLoss.py:97-126, train.py:402-427), one launch for
                                   the whole batch.  Multiple of 4 (16-byte aligned slices); general engine only (cfg < 32) */
} DvcConvDesc;
/* dvc_conv2d_winograd only: when the layer is split over input channels (dvc_conv2d_winograd_split > 1), leave the partial
 * sums [split][N][Cout][OH*OW] in the workspace instead of launching the reduce — bias, activation and `y` are then NOT applied /
 * written; the consumer sums them (dvc_instnorm_apply_partials).  No effect when the layer is not split. */
#define DVC_CONV_DEFER_REDUCE 1
/* dvc_conv2d_winograd / _pool / _dual / _split: plan the launch for the WHOLE batch (workgroups of all N images fill the chip
 * together, so layers that would be split over input channels for one image need a smaller split or none) instead of per
 * image.  Default (bit clear): the plan is that of a single image and a batch of N is bit-identical to N single-image calls
 * (train.py:402 calls the path with B = 16).  With the bit set the result is deterministic per (layer, N) but the fp32
 * summation order over input channels — hence the last-place rounding — depends on N.  Used where N images MUST run together
 * anyway: the R references of one clip (/root/reference/test.py:169-181), whose ColorVidNet recurrences advance in lock step. */
#define DVC_CONV_BATCH_PLAN 2
/* dvc_conv2d, the 3 -> 64 image-input layer only (VGG19 conv1_1): the stored input is ONE plane per image — the centred
 * luminance L, x_batch_stride elements apart (0 => H*W) — and every one of the three virtual input channels reads
 * (L + 50) / 100, i.e. gray2rgb_batch(uncenter_l(L)) (utils/util.py:63-64,97-101, FrameColor.py:9) folded into the load:
 * bit-identical to dvc_gray2rgb followed by the plain call, one launch and a 1 MB tensor less per frame.  The input affine
 * (vgg_preprocess) applies to that value as usual. */
#define DVC_CONV_GRAY_INPUT 4

/* Output spatial size implied by a descriptor. */
int dvc_conv2d_out_hw(const DvcConvDesc* d, int32_t* OH, int32_t* OW);

int dvc_conv2d(const DvcConvDesc* d,
               const float* x, const float* w_packed, const float* bias /* may be NULL */,
               const float* in_scale /* [N*Cin] or NULL */, const float* in_shift /* [N*Cin] or NULL */,
               const float* in_slope_ptr /* device scalar, used when in_prelu */,
               const float* act_slope_ptr /* device scalar or NULL */,
               const float* residual /* or NULL */, float* y,
               void* workspace /* or NULL: scratch for split-K partial sums */, size_t workspace_bytes,
               dvcStream stream);

/* The same 3x3 convolution (every 3x3 stride-1 nn.Conv2d of models/ColorVidNet.py:14-81, models/NonlocalNet.py:200-215,
 * 330-339, 359-420 whose input needs no fused per-element transform) in Winograd F(2x2,3x3) form: 2.25x fewer
 * multiplies, fp32 throughout, result within fp32 rounding of the direct sum (what cuDNN selects for these layers under
 * the reference's cudnn.benchmark = True, test.py:140).  Requirements: ksize 3, stride 1, dil 1|2 with pad == dil,
 * Cin % 8 == 0, Cout % 64 == 0, in_prelu == 0.  Descriptor fields as for dvc_conv2d except
 *   cfg      -1 = automatic | tile-block shape (0: 1x32, 1: 2x16, 2: 4x8, 3: 8x4 tiles) + 4 * workgroup shape
 *            (0: 128 channels x 32 tiles, 1: 64 channels x 64 tiles — 8 waves, one workgroup per CU; 2: 64 channels x
 *            32 tiles — 4 waves and 64 KB of LDS, two workgroups per CU)
 *   split_k  0 = automatic | 1..8 = split over input-channel chunks (needs the workspace, as dvc_conv2d)
 * u_packed: the filters in the transform domain, U = G g G^T, laid out [Cout/32][Cin][4][32][4]
 * (dvc_winograd_weight_floats(Cout, Cin) floats, written by dvc_winograd_pack_weight). */
size_t dvc_winograd_weight_floats(int32_t Cout, int32_t Cin);
/* w: [Cout][Cin][3][3] (the nn.Conv2d weight as stored in the reference's checkpoints) -> u_packed, evaluated in double
 * and rounded once.  Cout % 32 == 0. */
int dvc_winograd_pack_weight(const float* w, int32_t Cout, int32_t Cin, float* u_packed, dvcStream stream);
/* the split over input channels dvc_conv2d_winograd uses for this descriptor and workspace size (1 = none) and the number of
 * images one launch of it covers (workspace capacity at that split, 65535-workgroup cap); a pure function.  A caller that
 * wants DVC_CONV_DEFER_REDUCE must check images_per_launch >= d->N first. */
int dvc_conv2d_winograd_split(const DvcConvDesc* d, size_t workspace_bytes, int32_t* split, int32_t* images_per_launch);
int dvc_conv2d_winograd(const DvcConvDesc* d, const float* x, const float* u_packed,
                        const float* bias /* may be NULL */, const float* act_slope_ptr /* device scalar or NULL */,
                        const float* residual /* or NULL */, float* y,
                        void* workspace /* or NULL */, size_t workspace_bytes, dvcStream stream);
/* dvc_conv2d_winograd with nn.MaxPool2d(2, 2) of the activated output fused (VGG19: relu1_2 -> pool, relu2_2 -> pool, relu3_4 ->
 * pool, relu4_4 -> pool, /root/reference/models/NonlocalNet.py:240-254): y_pool [N][Cout][OH/2][OW/2] (floor mode;
 * pool_batch_stride in elements, 0 = dense) is written by the convolution's epilogue — a lane's 2x2 Winograd output tile is a
 * pooling window — or, when the layer is split over its input channels, by the reduce kernel.  y may be NULL when only the pooled
 * tensor is wanted.  Both outputs are bit-identical to dvc_conv2d_winograd followed by dvc_maxpool2x2.  Dilation 1, no residual. */
int dvc_conv2d_winograd_pool(const DvcConvDesc* d, const float* x, const float* u_packed, const float* bias /* may be NULL */,
                             const float* act_slope_ptr /* device scalar or NULL */, float* y /* or NULL */, float* y_pool,
                             int64_t pool_batch_stride, void* workspace /* or NULL */, size_t workspace_bytes, dvcStream stream);
/* TWO 3x3 convolutions whose outputs are added, as ONE launch: y = act(conv(T_A(xA), W_A) + conv(T_B(xB), W_B) + bias).
 * ColorVidNet.py:124-139 adds a skip convolution to the first convolution of every decoder block
 * (conv8_1(up(norm(c7_3))) + conv3_3_short(norm(c3_3)), likewise conv9_1 / conv10_1): the reduction simply runs over the
 * channels of both inputs, each with its own index map (dA / dB: Cin, H, W, in_up, in_sub; batch, Cout, dilation, padding
 * mode, activation and output geometry from dA and equal in dB), so the pair costs one set of per-launch / per-workgroup
 * fixed costs, one pass over the output and no residual read.  u_packed_cat: the two packed filter sets concatenated along
 * the input-channel axis ([Cout/32][CinA + CinB][4][32][4]); bias: the SUM of the two biases (or NULL).  Cin % 8 == 0 for
 * both, Cout % 64 == 0.  Rounding differs from the two-launch form only by the order of the fp32 accumulation. */
int dvc_conv2d_winograd_dual(const DvcConvDesc* dA, const DvcConvDesc* dB, const float* xA, const float* xB,
                             const float* u_packed_cat, const float* bias /* may be NULL */,
                             const float* act_slope_ptr /* device scalar or NULL */, float* y,
                             void* workspace /* or NULL */, size_t workspace_bytes, dvcStream stream);

/* r06 — weights-in-registers direct convolution (csrc/conv_ws.hip) for the large-map, few-channel 3x3 layers of ColorVidNet's
 * encoder (models/ColorVidNet.py:98-103: conv1_1[2] 32 -> 64 and conv1_2 64 -> 64 at full frame size, conv2_1 64 -> 128 at half
 * size): same arithmetic class as dvc_conv2d (exact fp32 products, blocked fp32 accumulation: chains of 8 channels x 9 taps
 * added to a running total), another decomposition — a wave keeps its 32 x 32 x 9 filter block in 144 VGPRs for the whole
 * launch and walks down a 32-pixel column strip whose input rows stream through an LDS ring.
 * Eligible (dvc_conv2d_ws_eligible != 0): ksize 3, stride 1, dil 1, pad 1, zero padding, no up / sub-sampling, no fused input
 * transform, Cin 32, 64 or 128, Cout % 64 == 0, act NONE / RELU / PRELU / LEAKY; no residual.  `u_packed`: Cout * Cin * 9 floats
 * written by dvc_conv2d_ws_pack_weight from the module's [Cout][Cin][3][3] weight (fragment order, 16-byte aligned).
 * x_batch_stride / y_batch_stride of the descriptor are honoured; cfg / split_k / flags are ignored (one plan per geometry,
 * per image: a batch of N is bit-identical to N calls). */
int dvc_conv2d_ws_eligible(const DvcConvDesc* d);
int dvc_conv2d_ws_pack_weight(const float* w, int32_t Cout, int32_t Cin, float* u_packed, dvcStream stream);
int dvc_conv2d_ws(const DvcConvDesc* d, const float* x, const float* u_packed, const float* bias /* may be NULL */,
                  const float* act_slope_ptr /* may be NULL */, float* y, dvcStream stream);

/* r06 — several INDEPENDENT 3x3 layers in ONE launch: the first (and the second) convolutions of WarpNet's four heads
 * (models/NonlocalNet.py:364-410; NonlocalNet.py:451-458 runs the heads on four different VGG taps, nothing connects them).
 * Item i is exactly dvc_conv2d_winograd(&d, x, u_packed, bias, act_slope_ptr, residual, y, workspace, workspace_bytes): the
 * same plan, the same kernel body, the same (deferred or launched) split-K reduce — results are bit-identical to the per-item
 * calls.  The items' workgroups form one grid when every item gets the 64-channel x 32-tile workgroup shape and a single
 * launch (every head layer at the path's sizes does); otherwise the call degrades to one launch per item.  The items must not
 * depend on each other; outputs and workspaces must be disjoint.  1 .. 4 items. */
typedef struct DvcConvGroupItem {
    DvcConvDesc d;
    const float* x;
    const float* u_packed;
    const float* bias;            /* or NULL */
    const float* act_slope_ptr;   /* or NULL */
    const float* residual;        /* or NULL */
    float* y;
    void* workspace;              /* this item's own split-K scratch (may be NULL: no split) */
    size_t workspace_bytes;
} DvcConvGroupItem;
int dvc_conv2d_winograd_group(const DvcConvGroupItem* items, int32_t n_items, dvcStream stream);

/* conv 1x1 with tiny Cout (<= 4) + optional tanh*128: ColorVidNet.conv10_ab, ColorVidNet.py:142-144.
 * w is the unpacked [Cout][Cin] matrix. */
int dvc_conv1x1_small(const float* x, const float* w, const float* bias, int32_t N, int32_t Cin,
                      int32_t HW, int32_t Cout, int32_t act, float* y, dvcStream stream);

/* ------------------------------------------------------------------------------------------------
 * InstanceNorm2d (eps, no affine, biased variance), split as "statistics" + "apply":
 *   models/NonlocalNet.py:335,339,368,...   models/ColorVidNet.py:85-94
 * stats: for every plane p=(n,c) of HW elements computes mean/var (fp64 accumulation, like ATen's CPU
 *   batch-norm statistics) and writes scale[p] = rstd * (chan_scale ? chan_scale[c] : 1),
 *   shift[p] = -mean * scale[p].  chan_scale carries the depthwise `*_ss` weights (ColorVidNet.py:12).
 */
int dvc_instnorm_stats(const float* x, int32_t N, int32_t C, int32_t HW, int64_t x_batch_stride,
                       float eps, const float* chan_scale, float* scale, float* shift,
                       dvcStream stream);

/* apply: y = prelu_or_id( x*scale + shift + residual ), optional nearest x`up` upsample and `rpad`
 * replicated rows on top and bottom (F.pad(...,(0,0,1,1),'replicate'), NonlocalNet.py:461-463).
 * slope_ptr == NULL => no activation.  y is [N][C][(H*up)+2*rpad][W*up] with its own batch stride. */
int dvc_affine_act(const float* x, const float* scale, const float* shift, const float* residual,
                   const float* slope_ptr, int32_t N, int32_t C, int32_t H, int32_t W, int32_t up,
                   int32_t rpad, int64_t x_batch_stride, int64_t res_batch_stride,
                   int64_t y_batch_stride, float* y, dvcStream stream);

/* InstanceNorm and what follows it in ONE launch: the statistics of every (n,c) plane as dvc_instnorm_stats,
 * then y = prelu_or_id( x*scale + shift + residual ) as dvc_affine_act, applied by the workgroup that reduced
 * the plane (its second read hits L2).  sub = 2 keeps every 2nd row/column (the stride-2 depthwise `*_ss`
 * convs, ColorVidNet.py:12,16,21; chan_scale = their weights); up / rpad as in dvc_affine_act; up and sub
 * exclude each other.  y is [N][C][VH + 2*rpad][VW], VH = H*up or ceil(H/2); y may be x itself when
 * up == sub == 1 and rpad == 0.  scale_out / shift_out (both or neither; may be NULL) receive the affine.
 * y2 (optional) is a second consumer's view of the same statistics: y2 = InstanceNorm(x) * chan_scale2 at stride
 * sub2, contiguous [N][C][ceil(H/sub2)][ceil(W/sub2)] (ColorVidNet normalises conv1_2 / conv2_2 / conv3_3 once for
 * the skip convolution and once, scaled and subsampled, for the next block).
 * The convolution that consumes y then needs no fused input transform and stages through LDS-DMA. */
int dvc_instnorm_apply(const float* x, const float* residual /* or NULL */, const float* slope_ptr /* or NULL */,
                       const float* chan_scale /* [C] or NULL */, float eps, int32_t N, int32_t C, int32_t H,
                       int32_t W, int32_t up, int32_t sub, int32_t rpad, int64_t x_batch_stride,
                       int64_t res_batch_stride, int64_t y_batch_stride, float* y,
                       float* scale_out, float* shift_out,
                       const float* chan_scale2 /* [C] or NULL */, int32_t sub2, float* y2 /* or NULL */,
                       dvcStream stream);
/* dvc_instnorm_apply on x = act(sum_s part[s] + bias): `part` = the [S][N][C][H*W] partial sums a convolution left in its
 * workspace (DVC_CONV_DEFER_REDUCE), bias [C] or NULL, act / act_slope / act_slope_ptr as DvcConvDesc.  Bit-identical to
 * reduce -> dvc_instnorm_apply; one launch and three passes over the tensor less.  H*W <= 16384 (the plane is built in LDS). */
int dvc_instnorm_apply_partials(const float* part, int32_t S, const float* bias, int32_t act, float act_slope,
                                const float* act_slope_ptr, const float* residual, const float* slope_ptr,
                                const float* chan_scale, float eps, int32_t N, int32_t C, int32_t H, int32_t W, int32_t up,
                                int32_t sub, int32_t rpad, int64_t res_batch_stride, int64_t y_batch_stride, float* y,
                                float* scale_out, float* shift_out, const float* chan_scale2, int32_t sub2, float* y2,
                                dvcStream stream);

/* r06 — several INDEPENDENT InstanceNorm launches as one: the norms of WarpNet's four heads
 * (models/NonlocalNet.py:364-410, run side by side by NonlocalNet.py:451-458).  Item i is dvc_instnorm_apply (S == 0: `x` is
 * the tensor, x_batch_stride elements per image, 0 => C*H*W) or dvc_instnorm_apply_partials (S >= 1: `x` is the [S][N][C][H*W]
 * partial sums, with bias / act / act_slope / act_slope_ptr; H*W <= 16384) without the second output; 1 .. 4 items.
 * Bit-identical to the per-item calls. */
typedef struct DvcInstNormItem {
    const float* x;
    int32_t S;
    const float* bias;            /* S >= 1 only, [C] or NULL */
    int32_t act;                  /* S >= 1 only, DVC_ACT_* of the convolution that left the partial sums */
    float act_slope;
    const float* act_slope_ptr;   /* or NULL */
    const float* residual;        /* or NULL */
    const float* slope_ptr;       /* PReLU slope of the norm's own activation, or NULL */
    const float* chan_scale;      /* [C] or NULL */
    float eps;
    int32_t N, C, H, W, up, sub, rpad;
    int64_t x_batch_stride, res_batch_stride, y_batch_stride;   /* elements; 0 => dense */
    float* y;
} DvcInstNormItem;
int dvc_instnorm_apply_group(const DvcInstNormItem* items, int32_t n_items, dvcStream stream);

/* nn.MaxPool2d(2,2) floor mode, NonlocalNet.py:237-255. planes = N*C. */
int dvc_maxpool2x2(const float* x, int32_t planes, int32_t H, int32_t W, float* y, dvcStream stream);
/* nn.AvgPool2d(2,2): the pool="avg" variant of VGG19_pytorch, NonlocalNet.py:221-226; also, bit for bit,
 * F.interpolate(x, scale_factor=0.5, mode="bilinear") — full-resolution Lab -> network resolution, test.py:58,71. */
int dvc_avgpool2x2(const float* x, int32_t planes, int32_t H, int32_t W, float* y, dvcStream stream);
/* F.avg_pool2d(x, 4), NonlocalNet.py:491. */
int dvc_avgpool4x4(const float* x, int32_t planes, int32_t H, int32_t W, float* y, dvcStream stream);
/* nn.Upsample(scale_factor=f) nearest, NonlocalNet.py:425,499-500. */
int dvc_upsample_nearest(const float* x, int32_t planes, int32_t H, int32_t W, int32_t f, float* y,
                         dvcStream stream);
/* feature_normalize: x / (||x||_2 over C + eps), utils/util.py:155-158. */
int dvc_channel_l2norm(const float* x, int32_t N, int32_t C, int32_t HW, float eps, float* y,
                       dvcStream stream);
/* The same for up to DVC_L2NORM_MAX_TENSORS feature maps of one batch in ONE launch (FrameColor.py:16-23 normalises relu2_1 ..
 * relu5_1 of a frame back to back; the small maps alone are a handful of workgroups each).  x / y / C / HW: host arrays of
 * `count` entries; every map needs H*W % 4 == 0 and 16-byte aligned pointers.  Same arithmetic per pixel as
 * dvc_channel_l2norm up to the summation order of the channel groups. */
#define DVC_L2NORM_MAX_TENSORS 8
int dvc_channel_l2norm_multi(const float* const* x, float* const* y, const int32_t* C, const int32_t* HW,
                             int32_t count, int32_t N, float eps, dvcStream stream);

/* ------------------------------------------------------------------------------------------------
 * Colour / glue elementwise ops.
 */
/* gray2rgb_batch (utils/util.py:97-101): y[n,0..2] = (L+50)/100. */
int dvc_gray2rgb(const float* l, int32_t N, int32_t HW, int64_t l_batch_stride, float* y,
                 dvcStream stream);
/* tensor_lab2rgb (utils/util.py:379-414): Lab (L in [0,100], i.e. already uncentred when
 * l_offset == 0; pass l_offset = 50 to fold uncenter_l, utils/util.py:63-64) -> sRGB [0,1]. */
int dvc_lab2rgb(const float* lab, int32_t N, int32_t HW, float l_offset, float* rgb,
                dvcStream stream);
/* cat((IA_l, nonlocal_BA_lab[:,1:3], similarity_map, IA_last_lab), 1): models/FrameColor.py:63-64 -> out7 [N][7][HW].
 * IA_l: the luminance plane of the current frame (element stride between images `ia_batch_stride`, 0 = HW; 3 HW when it
 * is channel 0 of a Lab tensor).  IA_last_lab arrives as its two parts, so that the clip loop never materialises
 * test.py:96's cat((IA_l, ab_predict)) between frames: last_l = luminance plane of the previous frame, last_ab = the
 * previous [N][2][HW] prediction (batch strides 0 = HW / 2 HW); for an existing Lab tensor pass (lab, 3 HW, lab + HW, 3 HW).
 * warped_lab: [N][3][HW] (channels 1, 2 are read); sim: [N][1][HW].
 * A NEGATIVE batch stride means "the same plane for every image" (stride 0): one frame against the R references of a clip
 * (/root/reference/test.py:169-181), whose R ColorVidNet inputs share the frame's luminance. */
int dvc_pack_color_input(const float* IA_l, int64_t ia_batch_stride, const float* warped_lab, const float* sim,
                         const float* last_l, int64_t last_l_batch_stride, const float* last_ab,
                         int64_t last_ab_batch_stride, int32_t N, int32_t HW, float* out7, dvcStream stream);

/* ---- clip-driver tail, test.py:98-116 ------------------------------------------------
 * F.interpolate(x, scale_factor=2, mode="bilinear") * mul on `planes` planes of H x W (test.py:100-102; ATen's
 * align_corners=False source index and its fma evaluation order, bit-exact for the sizes the path uses). */
int dvc_upsample_bilinear2x(const float* x, int32_t planes, int32_t H, int32_t W, float mul, float* y,
                            dvcStream stream);
/* (uncenter_l(L) * 255 / 100).astype(uint8): the guide image of the WLS filter, test.py:106-109. */
int dvc_lum_guide_u8(const float* L_centered, int64_t n, uint8_t* guide, dvcStream stream);
/* cv2.ximgproc.createFastGlobalSmootherFilter(guide, lambda, sigma_color).filter(src) on `planes` float planes
 * (test.py:107-111): Min et al. 2014, Alg. 1 — num_iter x {row solve, column solve}, lambda_t = 1.5 * 4^(T-t) /
 * (4^T - 1) * lambda, weights exp(-|dg| / sigma_color).  OpenCV's defaults: num_iter 3, attenuation 0.25.
 * Several frames (guides) are filtered by one call: planes g*planes_per_guide.. use guide g.  dst may be src.
 * Parity unpinned (opencv-contrib is absent from the build image; see oracle/tail_oracle.py). */
size_t dvc_fgs_workspace_bytes(int32_t H, int32_t W, int32_t n_guides, int32_t planes_per_guide, int32_t num_iter);
int dvc_fgs_filter(const uint8_t* guide /* [n_guides][H][W] */, const float* src /* [n_guides*planes_per_guide][H][W] */,
                   int32_t n_guides, int32_t planes_per_guide, int32_t H, int32_t W, float lambda, float sigma_color,
                   int32_t num_iter, float lambda_attenuation, float* dst, void* workspace, size_t workspace_bytes,
                   dvcStream stream);
/* batch_lab2rgb_transpose_mc for one image (utils/util.py:134-151): Lab (L centred) -> skimage lab2rgb (float64)
 * -> clip -> *255 -> uint8, H x W x 3.  ab = [2][H][W].  Parity unpinned (skimage absent). */
int dvc_lab2rgb_u8(const float* L_centered, const float* ab, int32_t H, int32_t W, uint8_t* rgb_hwc,
                   dvcStream stream);
/* Frame ingest, colour part: RGB2Lab() -> ToTensor() -> Normalize(),
 * utils/util_distortion.py:18-23,85-100: skimage rgb2lab (float64) of an 8-bit H x W x 3 image, .float(), L - 50;
 * lab = [3][H][W].  Parity unpinned (skimage absent); round trip with dvc_lab2rgb_u8 tested. */
int dvc_rgb8_to_lab(const uint8_t* rgb_hwc, int32_t H, int32_t W, float* lab, dvcStream stream);
/* Frame ingest, geometric half: CenterPad(image_size)(image), utils/util_distortion.py:217-258 — skimage's
 * anti-aliased resize (Gaussian pre-filter + bilinear sampling, mirror boundaries, float64) to the target width or
 * height, centre crop, astype(uint8).  img = [H0][W0][3], out = [H][W][3].  Parity unpinned (skimage absent; the
 * restatement is checked against the SciPy calls skimage makes).
 * r06: workspace == NULL selects the FUSED kernel (one launch: a workgroup stages the 8-bit source window of its 8 x 32 output
 * tile in LDS, every thread filters only the values its bilinear sample reads — same arithmetic in the same order, the same
 * bytes); it applies when dvc_center_pad_is_fused() says so (down-scaling with a Gaussian radius <= 4 on both axes: factors
 * up to 3.25; also the same-size copy).  With a workspace of dvc_center_pad_workspace_bytes the three full-frame passes of
 * r01 run (any factor up to 21). */
size_t dvc_center_pad_workspace_bytes(int32_t H0, int32_t W0);
int dvc_center_pad_is_fused(int32_t H0, int32_t W0, int32_t H, int32_t W);
int dvc_center_pad(const uint8_t* img, int32_t H0, int32_t W0, int32_t H, int32_t W, uint8_t* out, void* workspace,
                   size_t workspace_bytes, dvcStream stream);

/* ------------------------------------------------------------------------------------------------
 * Dense correlation (the north-star kernel).  Replaces models/NonlocalNet.py:469-500:
 *   centre + normalise theta/phi -> f = theta^T phi (P x P) -> similarity = rowmax f ->
 *   softmax(f / T) -> y = softmax @ avgpool4(B_lab) -> nearest x4 upsample of y and similarity.
 * The P x P affinity is never written to memory: query rows live in registers, exemplar ("key")
 * tiles stream through LDS, the 256-deep dot products run on v_mfma_f32_32x32x2_f32, and the row
 * softmax is an online (running max / running sum) recurrence kept per lane.
 */
/* t_raw[B][C][P] (output of the 1x1 theta/phi conv) -> t[b,c,p] = (t_raw - mean_p) / (||.||_C + eps)
 * mean_scratch: B*C floats. */
int dvc_corr_prepare(const float* t_raw, int32_t B, int32_t C, int32_t P, float eps,
                     float* mean_scratch, float* t_out, dvcStream stream);

size_t dvc_corr_workspace_bytes(int32_t B, int32_t P);

/* theta, phi: [B][C][P] centred+normalised (C must be 256).  blab: [B][3][P] pooled exemplar Lab.
 * temperature > 0.  wta_scale == 1 disables the winner-take-all rescale (NonlocalNet.py:486);
 * otherwise f' = (f == rowmax f) ? f : f * wta_scale (WTA_scale.forward, NonlocalNet.py:295-309).
 * Outputs (any may be NULL): y_small [B][3][h][w], sim_small [B][1][h][w] with h*w == P,
 * y_up [B][3][4h][4w], sim_up [B][1][4h][4w], argmax [B][P] (index of the largest affinity per row,
 * lowest index on exact ties).
 * With EVERY output NULL (B == 1, wta_scale == 1) the merge of the per-workgroup partial softmax states is left to the
 * consumer: they stay in `workspace` (dvc_corr_workspace_bytes(1, P) bytes, which the caller then must not reuse) for
 * dvc_corr_merge_pack. */
int dvc_corr_fwd(const float* theta, const float* phi, const float* blab, float temperature,
                 float wta_scale, int32_t B, int32_t C, int32_t h, int32_t w, float* y_small,
                 float* sim_small, float* y_up, float* sim_up, int32_t* argmax, void* workspace,
                 size_t workspace_bytes, dvcStream stream);

/* The merge of a deferred dvc_corr_fwd (same temperature, h, w; `workspace` as that call left it) fused with the consumer of
 * its results, cat((IA_l, nonlocal_BA_lab[:,1:3], similarity_map, IA_last_lab), 1) of models/FrameColor.py:63-64 (see
 * dvc_pack_color_input) for ONE image: out7 [7][16 P]; IA_l, last_l: [16 P] planes; last_ab: [2][16 P].  The warped colours and
 * the similarity map go straight into channels 1..3 (x4 nearest, NonlocalNet.py:499-500) and are additionally written to
 * y_up [3][16 P] / sim_up [16 P] when those are not NULL (FrameColor.py:41-67 returns nonlocal_BA_lab).  Same arithmetic, same
 * order as the merge inside dvc_corr_fwd: bit-identical values.  Every pointer 16-byte aligned. */
int dvc_corr_merge_pack(const void* workspace, size_t workspace_bytes, float temperature, int32_t h, int32_t w,
                        const float* IA_l, const float* last_l, const float* last_ab, float* out7,
                        float* y_up /* or NULL */, float* sim_up /* or NULL */, dvcStream stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 mixed-precision correlation (BASELINE.json configs[4]).  Same reference lines as dvc_corr_fwd
 * (models/NonlocalNet.py:469-500); bf16 MFMA affinities are used as a candidate filter and every key
 * within 2*2^-8 of the bf16 row maximum is re-scored in exact fp32, which provably contains the fp32
 * argmax (unit-norm columns => |f_bf16 - f| <= 2^-8).  Exact for temperature <= 1e-4 (test.py:94 uses
 * 1e-10); larger temperatures are rejected (use dvc_corr_fwd).
 */
/* t_raw[B][C][P] -> centred + normalised copies in [B][P][C] layout: fp32 and bf16 (round-to-nearest-even). */
int dvc_corr_prepare_bf16(const float* t_raw, int32_t B, int32_t C, int32_t P, float eps,
                          float* mean_scratch, float* t_f32_pc, void* t_bf16_pc, dvcStream stream);
size_t dvc_corr_bf16_workspace_bytes(int32_t B, int32_t P);
int dvc_corr_fwd_bf16(const void* theta_bf16_pc, const void* phi_bf16_pc, const float* theta_f32_pc,
                      const float* phi_f32_pc, const float* blab, float temperature, int32_t B, int32_t C,
                      int32_t h, int32_t w, float* y_small, float* sim_small, float* y_up, float* sim_up,
                      int32_t* argmax, void* workspace, size_t workspace_bytes, dvcStream stream);

/* ------------------------------------------------------------------------------------------------
 * Contextual loss (models/ContextualLoss.py:29-126: `ContextualLoss_forward` :88-126 and `ContextualLoss` :29-77; called by
 * train.py:649-668 on relu3_1 / relu4_1 / relu5_1 features of the prediction and of the exemplar).  Per image, X = predicted,
 * Y = exemplar features [C][N]:   mu = mean_j Y;  Xn, Yn = (. - mu) / (||.||_C + eps);  S = Xn^T Yn;  d = 1 - S;
 * a_i = min_j d_ij + 1e-5;  A = softmax_j((1 - d / a_i) / h);  CX = mean_i max_j A_ij  (_forward)  |  mean_j max_i A_ij;
 * loss = -log CX.  The GEMMs run as 1x1 convolutions with per-image filters — batched GEMMs, DvcConvDesc::w_batch_stride —
 * (dvc_conv2d) from the host (dvc_amd/contextual.py) in blocks of rows of S, all images of the batch per launch; these entries are the pieces in between, forward and backward (gradient w.r.t. X; Y is data, as in train.py). */
/* x [B][C][P] -> out = (x - mean) / (||x - mean||_C + eps).  centre == 0: no mean.  mean_in != NULL: use it ([B*C], e.g. the
 * exemplar's, ContextualLoss.py:49-51); otherwise the own per-channel mean is computed into mean_out.  norm_out [B][P] (or
 * NULL) receives ||x - mean||_C (needed by dvc_cx_normalize_bwd). */
int dvc_cx_prepare(const float* x, const float* mean_in, int32_t centre, int32_t B, int32_t C, int32_t P, float eps,
                   float* mean_out, float* norm_out, float* out, dvcStream stream);
/* The entries below take a BATCH of nb images in one launch (r04; the products in between are batched GEMMs, DvcConvDesc::
 * w_batch_stride): S_bs = elements between the images' S blocks, row_bs = elements between their per-row arrays (a, jstar, l,
 * r, e: [nb][Nx] slices starting at the block's first row), tq_bs = between their t / q arrays; cmax / cargi are [nb][N],
 * gscale / loss [nb], dST [nb][N][ld_t]. */
/* S [nb][rows][N] (a block of rows of Xn^T Yn per image) -> per row: a = (1 - max_j S) + 1e-5, jstar = arg max_j S (lowest index
 * on ties), l = sum_j w, r = max_j A = w_jstar / l, e = sum_j A_ij d_ij, with w = exp((1 - d / a) / h). */
int dvc_cx_rows(const float* S, int32_t nb, int64_t S_bs, int64_t row_bs, int32_t rows, int32_t N, float h, float* a, int32_t* jstar,
                float* l, float* r, float* e, dvcStream stream);
/* ContextualLoss only: running column maxima of A over row blocks (cmax initialised below 0, cargi = the winning global row,
 * row0 = global index of the block's first row). */
int dvc_cx_colmax(const float* S, int32_t nb, int64_t S_bs, int64_t row_bs, const float* a, const float* l, int32_t rows, int32_t N,
                  int32_t row0, float h, float* cmax, int32_t* cargi, dvcStream stream);
/* per image b of v [nb][n]: loss[b] = -log(mean(v[b])),  gscale[b] = d loss / d v_k = -1 / (n mean(v[b])) */
int dvc_cx_finish(const float* v, int32_t nb, int32_t n, float* loss, float* gscale, dvcStream stream);
/* ContextualLoss backward: per row i of the block, t = sum of A_ik over the columns k whose maximum sits in row i
 * (cargi[k] == row0 + i), q = the same sum weighted by d_ik. */
int dvc_cx_rows_tq(const float* S, int32_t nb, int64_t S_bs, int64_t row_bs, int64_t tq_bs, const float* a, const float* l,
                   const int32_t* cargi, int32_t rows, int32_t N, int32_t row0, float h, float* t, float* q, dvcStream stream);
/* d loss / d S for a block of rows, row-major (dS, may be NULL) and transposed (dST [nb][N][ld_t], rows >= `rows` zero-filled:
 * the K-major operand of d Xn = Yn dS^T; may be NULL since r06 when dS is given).  mode 0 = ContextualLoss_forward, 1 = ContextualLoss (needs cargi, t, q).
 * gscale: [nb] device values from dvc_cx_finish (times the upstream gradient of each image's loss); gout: a common factor. */
int dvc_cx_ds(const float* S, int32_t nb, int64_t S_bs, int64_t row_bs, int64_t tq_bs, const float* a, const float* l, const float* r,
              const float* e, const int32_t* jstar, const int32_t* cargi, const float* t, const float* q, const float* gscale,
              float gout, int32_t mode, int32_t rows, int32_t N, int32_t row0, int32_t ld_t, float h, float* dS, float* dST,
              dvcStream stream);
/* backward of xn = xc / (||xc|| + eps) per position: dx = dxn / (n + eps) - xn (xn . dxn) / n. */
int dvc_cx_normalize_bwd(const float* xn, const float* norm, const float* dxn, int32_t B, int32_t C, int32_t P, float eps,
                         float* dx, dvcStream stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the fused correlation (training-side callers: train.py:402-427 runs frame_colorization, hence
 * WarpNet.forward, models/NonlocalNet.py:477-500, under autograd at temperature 0.01).  The host walks the query rows
 * in blocks (dvc_amd/corr_autograd.py); for one block of `rows` queries this entry turns the block's affinities
 * f_blk[rows][P] (= theta_blk^T phi, produced by dvc_conv2d as a 1x1 convolution) into
 *     dS[i][j] = p_ij * (g_i . B_j - g_i . y_i) / T  (+ gsim[i] at j == argmax[i]),   p = softmax_j(f / T)
 * written row-major (dS[rows][P]) and transposed (dST[P][ld_t], rows >= `rows` zero-filled) — the two K-major operands
 * of the 1x1-convolution GEMMs that follow (d phi += theta_blk dS, d theta_blk = phi dS^T).  dST may be NULL (r06): the
 * transposed copy is not written — for a caller whose GEMM takes dS transposed as it is (rocBLAS through torch.bmm).
 * gy, y: [3][.] with `chan_stride` elements between channels, already offset to the block's first query; sim, gsim,
 * argmax likewise (gsim / argmax may both be NULL: no gradient through the similarity map).  The softmax is re-evaluated
 * around the row maximum of f_blk ITSELF (the block is recomputed by another GEMM order than the forward kernel's, so the
 * saved similarity is not guaranteed to bound it: at T <= 1e-7 a 1e-7 excess would overflow the exponential); `sim` is
 * not read and is kept for the signature.  rowstat_scratch: [3][ld_t] floats (row maxima, row sums, raw row maxima).
 * wta_scale != 1 (r05): WTA_scale of NonlocalNet.py:288-327 ahead of the temperature — f' = (f == max_j f) ? f : f * wta_scale, and its
 * backward's factor (1 at the row maximum, the reference's constant 1e-4 elsewhere); gsim (similarity from f BEFORE it) unscaled.
 * batch (r05): images per call, dense: f_blk / dS [batch][ld_t][P], dST [batch][P][ld_t], blab / gy / y [batch][3][chan_stride]
 * (gy, y offset to the block's first row), gsim / argmax [batch][chan_stride], rowstat_scratch [batch][3][ld_t]. */
int dvc_corr_softmax_bwd(const float* f_blk, const float* blab, const float* gy, const float* y, const float* sim,
                         const float* gsim, const int32_t* argmax, float temperature, float wta_scale, int32_t batch, int32_t rows, int32_t P,
                         int64_t chan_stride, int32_t ld_t, float* rowstat_scratch, float* dS, float* dST,
                         dvcStream stream);

#ifdef DVC_DEBUG
/* ------------------------------------------------------------------------------------------------
 * Diagnostics for the timing probes under tools/ — ONLY in a -DDVC_DEBUG build (`make -C csrc DEBUG=1` ->
 * dvc_amd/libdvc_hip_debug.so); the production library has neither these symbols nor the state behind them
 * (process-global switches, not thread-safe; all default to off and every probe switches them off again).
 *   dvc_debug_conv_trace     buf != NULL: LDS-DMA conv kernels record {entry, end of chunk loop, HW_ID, XCC_ID} per workgroup
 *   dvc_debug_conv_variant   1 / 2 / 3: skip all / patch / weight staging after the first chunk (timing only, wrong results)
 *   dvc_debug_corr_timeline  buf != NULL: per-tile s_memtime stamps of wave 0 of every correlation workgroup
 *   dvc_debug_corr_variant   1: skip the softmax arithmetic of dvc_corr_fwd (timing only, wrong results) */
void dvc_debug_conv_trace(long long* buf);
void dvc_debug_conv_variant(int v);
void dvc_debug_corr_timeline(long long* buf, int max_tiles);
void dvc_debug_corr_variant(int v);
#endif /* DVC_DEBUG */

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* DVC_HIP_H */
