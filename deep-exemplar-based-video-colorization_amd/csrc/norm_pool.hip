// Bandwidth-bound helpers around the conv engine: InstanceNorm statistics / apply, pooling,
// nearest upsampling and the channel L2 normalisation.  All are deterministic (no atomics):
// reductions are wave shuffles + one LDS hop, so repeated runs are bit-identical.
#include "common.h"
#include <algorithm>

thread_local char g_dvc_err[512] = {0};
char* dvc_err_buf() { return g_dvc_err; }

extern "C" int dvc_abi_version(void) { return DVC_ABI_VERSION; }
extern "C" const char* dvc_last_error(void) { return g_dvc_err; }

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block-wide sum of doubles (256 threads = 4 waves); result broadcast to all threads
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// ------------------------------------------------------------------------------------------------
// InstanceNorm statistics: one workgroup per (n,c) plane, ONE pass: sum and sum of squares accumulated in
// fp64 (fp32 inputs are exact in fp64 and 53-bit accumulation leaves > 25 bits after the E[x^2]-E[x]^2
// cancellation for any realistic mean/std ratio; ATen's CPU statistics also accumulate in double).
// The summation tree is that of 1024 VIRTUAL threads whatever the block size (256, 512 or 1024): virtual thread v sums the
// float4 pieces v, v + 1024, ... in order, 64 consecutive virtual threads are combined by the wave butterfly, the 16 wave
// sums are added in ascending order — a real thread carries 1024 / blockDim.x virtual ones in separate accumulators.  The
// statistics of a plane are therefore bit-identical from every kernel that calls this (dvc_instnorm_stats, dvc_instnorm_apply
// and dvc_instnorm_apply_partials launch different block sizes).  Every thread returns the plane's
// (scale, shift) = (rstd * chan_scale, -mean * scale).
__device__ __forceinline__ void plane_stats(const float* __restrict__ xp, int HW, float eps, float cs,
                                            double* red /* [36] */, float* sc_out, float* sh_out) {
    constexpr int VT = 1024;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nv = VT / nt;        // virtual threads per real thread: 1, 2 or 4
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    const int HW4 = ((reinterpret_cast<uintptr_t>(xp) & 15) == 0) ? (HW & ~3) : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < nv) {
            const int v = tid + k * nt;
            int i = v * 4;
            // (r06: four pieces in flight per virtual thread, accumulated in the same order as one at a time — as a plain loop
            // the compiler waits for every load before it issues the next.  Four, not eight: the kernels that call this must stay
            // under 64 VGPRs, two 1024-thread workgroups per CU — eight cost the grouped launch 4 us)
            for (; i + 3 * VT * 4 < HW4; i += 4 * VT * 4) {
                float4 x8[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x8[u] = *reinterpret_cast<const float4*>(xp + i + u * (VT * 4));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double a = x8[u].x, b = x8[u].y, cc = x8[u].z, d = x8[u].w;
                    s[k] += (a + b) + (cc + d);
                    q[k] += (a * a + b * b) + (cc * cc + d * d);
                }
            }
            for (; i < HW4; i += VT * 4) {
                float4 x4 = *reinterpret_cast<const float4*>(xp + i);
                double a = x4.x, b = x4.y, cc = x4.z, d = x4.w;
                s[k] += (a + b) + (cc + d);
                q[k] += (a * a + b * b) + (cc * cc + d * d);
            }
            for (int i = HW4 + v; i < HW; i += VT) {
                double a = xp[i];
                s[k] += a;
                q[k] += a * a;
            }
        }
    }
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < nv) {
            const double sk = wave_sum_d(s[k]), qk = wave_sum_d(q[k]);
            if (lane == 0) {
                red[wave + k * nw] = sk;
                red[16 + wave + k * nw] = qk;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        double S = 0.0, Q = 0.0;
        for (int i = 0; i < VT / 64; ++i) {
            S += red[i];
            Q += red[16 + i];
        }
        const double mean = S / (double)HW;
        double var = Q / (double)HW - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double sc = rstd * (double)cs;
        red[32] = sc;
        red[33] = -mean * sc;
        red[34] = mean;
        red[35] = rstd;
    }
    __syncthreads();
    *sc_out = (float)red[32];
    *sh_out = (float)red[33];
}

__global__ __launch_bounds__(1024) void instnorm_stats_kernel(const float* __restrict__ x, int C, int HW,
                                                              long x_bs, float eps,
                                                              const float* __restrict__ chan_scale,
                                                              float* __restrict__ scale,
                                                              float* __restrict__ shift) {
    __shared__ double red[36];
    const int p = blockIdx.x;  // n*C + c
    const int n = p / C, c = p - n * C;
    float sc, sh;
    plane_stats(x + (long)n * x_bs + (long)c * HW, HW, eps, chan_scale ? chan_scale[c] : 1.f, red, &sc, &sh);
    if (threadIdx.x == 0) {
        scale[p] = sc;
        shift[p] = sh;
    }
}

// threads per plane: enough loads in flight for the large (full-resolution) planes
static int instnorm_block(int HW) { return HW >= 32768 ? 1024 : HW >= 8192 ? 512 : 256; }

extern "C" int dvc_instnorm_stats(const float* x, int32_t N, int32_t C, int32_t HW,
                                  int64_t x_batch_stride, float eps, const float* chan_scale,
                                  float* scale, float* shift, dvcStream stream) {
    DVC_REQUIRE(x && scale && shift && N > 0 && C > 0 && HW > 0, "dvc_instnorm_stats: bad argument");
    long bs = x_batch_stride ? x_batch_stride : (long)C * HW;
    hipLaunchKernelGGL(instnorm_stats_kernel, dim3(N * C), dim3(instnorm_block(HW)), 0, (hipStream_t)stream,
                       x, C, HW, bs, eps, chan_scale, scale, shift);
    DVC_CHECK_LAUNCH("dvc_instnorm_stats");
    return 0;
}

// InstanceNorm + what follows it, one launch: the workgroup that reduced the plane also applies
// y = prelu_or_id(x*scale + shift + residual) to it (second read from L2), with the output index maps of
// affine_act_kernel plus stride-`sub` subsampling.
// (body shared by the two kernels below: `xp` is the plane to normalise — global memory, or the LDS image the partial-sum
// variant has just built; generic pointer either way)
__device__ __forceinline__ void instnorm_apply_plane(const float* xp, const int p, const int n, const int c, double* red,
                                                     const float* __restrict__ res, const float* __restrict__ slope_ptr,
                                                     const float* __restrict__ chan_scale, float eps, int C, int H, int W,
                                                     int up, int sub, int rpad, long res_bs, long y_bs, float* __restrict__ y,
                                                     float* __restrict__ scale, float* __restrict__ shift,
                                                     const float* __restrict__ chan_scale2, int sub2, long y2_bs,
                                                     float* __restrict__ y2) {
    float sc, sh;
    plane_stats(xp, H * W, eps, chan_scale ? chan_scale[c] : 1.f, red, &sc, &sh);
    if (threadIdx.x == 0 && scale) {
        scale[p] = sc;
        shift[p] = sh;
    }
    if (y2) {  // second consumer of the same statistics: y2 = x * (rstd * chan_scale2) + shift, stride sub2
        const double sc2d = red[35] * (chan_scale2 ? (double)chan_scale2[c] : 1.0);
        const float sc2 = (float)sc2d, sh2 = (float)(-red[34] * sc2d);
        const int OH2 = sub2 == 2 ? (H + 1) / 2 : H, OW2 = sub2 == 2 ? (W + 1) / 2 : W;
        float* y2p = y2 + (long)n * y2_bs + (long)c * OH2 * OW2;
        for (int i = threadIdx.x; i < OH2 * OW2; i += blockDim.x) {
            const int oy = i / OW2, ox = i - oy * OW2;
            y2p[i] = xp[(oy * sub2) * W + ox * sub2] * sc2 + sh2;
        }
    }
    const bool has_act = slope_ptr != nullptr;
    const float slope = has_act ? *slope_ptr : 1.f;
    const float* rp = res ? res + (long)n * res_bs + (long)c * H * W : nullptr;
    const int VH = sub == 2 ? (H + 1) / 2 : H * up, VW = sub == 2 ? (W + 1) / 2 : W * up;
    const int OH = VH + 2 * rpad, OW = VW;
    float* yp = y + (long)n * y_bs + (long)c * OH * OW;
    const int total = OH * OW, nt = blockDim.x;
    if (up == 1 && sub == 1 && rpad == 0) {  // elementwise (possibly in place): float4 when aligned
        const bool v4 = ((reinterpret_cast<uintptr_t>(xp) | reinterpret_cast<uintptr_t>(yp) |
                          reinterpret_cast<uintptr_t>(rp)) & 15) == 0;
        const int T4 = v4 ? (total & ~3) : 0;
        auto piece = [&](const float4& v, const float4& r, int i) {
            float o[4] = {v.x * sc + sh, v.y * sc + sh, v.z * sc + sh, v.w * sc + sh};
            if (rp) { o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w; }
            if (has_act) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = o[k] >= 0.f ? o[k] : o[k] * slope;
            }
            *reinterpret_cast<float4*>(yp + i) = make_float4(o[0], o[1], o[2], o[3]);
        };
        int i = threadIdx.x * 4;
        // (two pieces in flight per thread — elementwise, so the order is immaterial; in place each thread reads its two
        // pieces before it writes either, and no other thread touches them)
        for (; i + nt * 4 < T4; i += 2 * nt * 4) {
            float4 v[2], r[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                v[u] = *reinterpret_cast<const float4*>(xp + i + u * nt * 4);
                r[u] = rp ? *reinterpret_cast<const float4*>(rp + i + u * nt * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) piece(v[u], r[u], i + u * nt * 4);
        }
        for (; i < T4; i += nt * 4) {
            const float4 v = *reinterpret_cast<const float4*>(xp + i);
            const float4 r = rp ? *reinterpret_cast<const float4*>(rp + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            piece(v, r, i);
        }
        for (int i = T4 + threadIdx.x; i < total; i += nt) {
            float v = xp[i] * sc + sh;
            if (rp) v += rp[i];
            if (has_act) v = v >= 0.f ? v : v * slope;
            yp[i] = v;
        }
        return;
    }
    for (int i = threadIdx.x; i < total; i += nt) {
        int oy = i / OW, ox = i - oy * OW;
        int uy = oy - rpad;
        uy = uy < 0 ? 0 : (uy >= VH ? VH - 1 : uy);
        int sy = sub == 2 ? uy * 2 : (up == 1 ? uy : uy / up);
        int sx = sub == 2 ? ox * 2 : (up == 1 ? ox : ox / up);
        float v = xp[sy * W + sx] * sc + sh;
        if (rp) v += rp[sy * W + sx];
        if (has_act) v = v >= 0.f ? v : v * slope;
        yp[i] = v;
    }
}


__global__ __launch_bounds__(1024) void instnorm_apply_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ res,
                                                              const float* __restrict__ slope_ptr,
                                                              const float* __restrict__ chan_scale, float eps,
                                                              int C, int H, int W, int up, int sub, int rpad,
                                                              long x_bs, long res_bs, long y_bs,
                                                              float* __restrict__ y, float* __restrict__ scale,
                                                              float* __restrict__ shift,
                                                              const float* __restrict__ chan_scale2, int sub2,
                                                              long y2_bs, float* __restrict__ y2) {
    __shared__ double red[36];
    const int p = blockIdx.x;  // n*C + c
    const int n = p / C, c = p - n * C;
    instnorm_apply_plane(x + (long)n * x_bs + (long)c * H * W, p, n, c, red, res, slope_ptr, chan_scale, eps, C, H, W, up, sub,
                         rpad, res_bs, y_bs, y, scale, shift, chan_scale2, sub2, y2_bs, y2);
}

// The same, with the plane given as the S split-K partial sums a convolution left in its workspace (dvc_conv2d_winograd with
// DVC_CONV_DEFER_REDUCE): x = act(sum_s part[s] + bias), summed in the reduce kernel's order (conv_splitk_reduce_kernel: the
// result is bit-identical to reduce -> dvc_instnorm_apply), built once in LDS — the reduce launch, its write of the plane and
// the two reads the statistics and the apply pass would make of it are gone.
__global__ __launch_bounds__(1024) void instnorm_apply_partials_kernel(const float* __restrict__ part, int S, long slab,
                                                                       const float* __restrict__ bias, int act, float act_slope,
                                                                       const float* __restrict__ act_slope_ptr,
                                                                       const float* __restrict__ res,
                                                                       const float* __restrict__ slope_ptr,
                                                                       const float* __restrict__ chan_scale, float eps,
                                                                       int C, int H, int W, int up, int sub, int rpad,
                                                                       long res_bs, long y_bs,
                                                                       float* __restrict__ y, float* __restrict__ scale,
                                                                       float* __restrict__ shift,
                                                                       const float* __restrict__ chan_scale2, int sub2,
                                                                       long y2_bs, float* __restrict__ y2) {
    __shared__ double red[36];
    extern __shared__ __attribute__((aligned(16))) float plane[];   // H * W floats
    const int p = blockIdx.x;  // n*C + c
    const int n = p / C, c = p - n * C;
    const int HW = H * W;
    const float* p0 = part + (long)p * HW;
    const float b = bias ? bias[c] : 0.f;
    const float aslope = act_slope_ptr ? *act_slope_ptr : act_slope;
    auto finish = [&](float v) {
        v += b;
        if (act == DVC_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (act == DVC_ACT_PRELU || act == DVC_ACT_LEAKY) v = v >= 0.f ? v : v * aslope;
        return v;
    };
    // float4 pieces, all S loads of a piece in flight (S <= 8), summed in the reduce kernel's order
    const bool v4 = (HW % 4 == 0) && (slab % 4 == 0) && ((reinterpret_cast<uintptr_t>(part) & 15) == 0);
    const int HW4 = v4 ? HW : 0;
    for (int i = threadIdx.x * 4; i < HW4; i += blockDim.x * 4) {
        float4 t[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) t[s] = s < S ? *reinterpret_cast<const float4*>(p0 + (long)s * slab + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < 8; ++s) { v.x += t[s].x; v.y += t[s].y; v.z += t[s].z; v.w += t[s].w; }
        *reinterpret_cast<float4*>(plane + i) = make_float4(finish(v.x), finish(v.y), finish(v.z), finish(v.w));
    }
    for (int i = HW4 + threadIdx.x; i < HW; i += blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += p0[(long)s * slab + i];
        plane[i] = finish(v);
    }
    __syncthreads();
    instnorm_apply_plane(plane, p, n, c, red, res, slope_ptr, chan_scale, eps, C, H, W, up, sub, rpad, res_bs, y_bs, y, scale,
                         shift, chan_scale2, sub2, y2_bs, y2);
}

extern "C" int dvc_instnorm_apply(const float* x, const float* residual, const float* slope_ptr,
                                  const float* chan_scale, float eps, int32_t N, int32_t C, int32_t H,
                                  int32_t W, int32_t up, int32_t sub, int32_t rpad, int64_t x_batch_stride,
                                  int64_t res_batch_stride, int64_t y_batch_stride, float* y,
                                  float* scale_out, float* shift_out, const float* chan_scale2, int32_t sub2,
                                  float* y2, dvcStream stream) {
    DVC_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, "dvc_instnorm_apply: bad argument");
    DVC_REQUIRE(!y2 || ((sub2 == 1 || sub2 == 2) && y2 != x && y2 != y), "dvc_instnorm_apply: bad second output");
    DVC_REQUIRE(!(y2 && x == y), "dvc_instnorm_apply: a second output excludes in-place operation");
    DVC_REQUIRE(up >= 1 && up <= 4 && (sub == 1 || sub == 2) && rpad >= 0 && !(up != 1 && sub != 1),
                "dvc_instnorm_apply: bad up/sub/rpad");
    DVC_REQUIRE(!(residual && (up != 1 || sub != 1)), "dvc_instnorm_apply: residual requires up == sub == 1");
    DVC_REQUIRE((scale_out == nullptr) == (shift_out == nullptr), "dvc_instnorm_apply: scale/shift come together");
    DVC_REQUIRE(!(x == y && (up != 1 || sub != 1 || rpad != 0)), "dvc_instnorm_apply: in place needs up == sub == 1, rpad == 0");
    const long VH = sub == 2 ? (H + 1) / 2 : (long)H * up, VW = sub == 2 ? (W + 1) / 2 : (long)W * up;
    const long OH = VH + 2 * rpad, OW = VW;
    long xbs = x_batch_stride ? x_batch_stride : (long)C * H * W;
    long rbs = res_batch_stride ? res_batch_stride : (long)C * H * W;
    long ybs = y_batch_stride ? y_batch_stride : (long)C * OH * OW;
    hipLaunchKernelGGL(instnorm_apply_kernel, dim3(N * C), dim3(instnorm_block(H * W)), 0, (hipStream_t)stream,
                       x, residual, slope_ptr, chan_scale, eps, C, H, W, up, sub, rpad, xbs, rbs, ybs, y,
                       scale_out, shift_out, chan_scale2, y2 ? sub2 : 1,
                       (long)C * (sub2 == 2 ? (H + 1) / 2 : H) * (sub2 == 2 ? (W + 1) / 2 : W), y2);
    DVC_CHECK_LAUNCH("dvc_instnorm_apply");
    return 0;
}

extern "C" int dvc_instnorm_apply_partials(const float* part, int32_t S, const float* bias, int32_t act, float act_slope,
                                           const float* act_slope_ptr, const float* residual, const float* slope_ptr,
                                           const float* chan_scale, float eps, int32_t N, int32_t C, int32_t H, int32_t W,
                                           int32_t up, int32_t sub, int32_t rpad, int64_t res_batch_stride,
                                           int64_t y_batch_stride, float* y, float* scale_out, float* shift_out,
                                           const float* chan_scale2, int32_t sub2, float* y2, dvcStream stream) {
    DVC_REQUIRE(part && y && S >= 1 && S <= 8 && N > 0 && C > 0 && H > 0 && W > 0, "dvc_instnorm_apply_partials: bad argument");
    DVC_REQUIRE((long)H * W <= 16384, "dvc_instnorm_apply_partials: plane of %d x %d does not fit the LDS image (<= 16384 elements)", H, W);
    DVC_REQUIRE(act == DVC_ACT_NONE || act == DVC_ACT_RELU || act == DVC_ACT_PRELU || act == DVC_ACT_LEAKY,
                "dvc_instnorm_apply_partials: unsupported activation %d", act);
    DVC_REQUIRE(!y2 || ((sub2 == 1 || sub2 == 2) && y2 != y), "dvc_instnorm_apply_partials: bad second output");
    DVC_REQUIRE(up >= 1 && up <= 4 && (sub == 1 || sub == 2) && rpad >= 0 && !(up != 1 && sub != 1),
                "dvc_instnorm_apply_partials: bad up/sub/rpad");
    DVC_REQUIRE(!(residual && (up != 1 || sub != 1)), "dvc_instnorm_apply_partials: residual requires up == sub == 1");
    DVC_REQUIRE((scale_out == nullptr) == (shift_out == nullptr), "dvc_instnorm_apply_partials: scale/shift come together");
    const long VH = sub == 2 ? (H + 1) / 2 : (long)H * up, VW = sub == 2 ? (W + 1) / 2 : (long)W * up;
    const long OH = VH + 2 * rpad, OW = VW;
    const long rbs = res_batch_stride ? res_batch_stride : (long)C * H * W;
    const long ybs = y_batch_stride ? y_batch_stride : (long)C * OH * OW;
    // (more threads per plane than the plain kernel: the partial sums are S x the plane, all of it first-touch traffic)
    hipLaunchKernelGGL(instnorm_apply_partials_kernel, dim3(N * C), dim3(H * W >= 4096 ? 1024 : 512), (size_t)H * W * sizeof(float),
                       (hipStream_t)stream, part, S, (long)N * C * H * W, bias, act, act_slope, act_slope_ptr, residual, slope_ptr,
                       chan_scale, eps, C, H, W, up, sub, rpad, rbs, ybs, y, scale_out, shift_out, chan_scale2, y2 ? sub2 : 1,
                       (long)C * (sub2 == 2 ? (H + 1) / 2 : H) * (sub2 == 2 ? (W + 1) / 2 : W), y2);
    DVC_CHECK_LAUNCH("dvc_instnorm_apply_partials");
    return 0;
}

// ---- several independent InstanceNorm (+ PReLU / upsample / pad) launches as one (r06): the norms between and behind the two
// convolutions of WarpNet's four heads (NonlocalNet.py:364-410) are mutually independent, 64-256 planes each — four launches of
// ~8 us that cannot fill the chip one by one.  One workgroup per (item, n, c) plane, the bodies of the two kernels above
// (an item is either a tensor or the split-K partial sums of a convolution): bit-identical to the per-item launches, whose
// statistics do not depend on the block size (plane_stats).
#define INSTNORM_GROUP_MAX 4
struct InstNormGroupArgs {
    DvcInstNormItem item[INSTNORM_GROUP_MAX];
    int start[INSTNORM_GROUP_MAX + 1];     // prefix sums of N * C
    int n;
};
__global__ __launch_bounds__(1024) void instnorm_group_kernel(InstNormGroupArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ double red[36];
    extern __shared__ __attribute__((aligned(16))) float plane[];   // the largest H * W among the partial-sum items
    // (compile-time indices only, and the selected item read where it lies in the kernel-argument segment: a dynamic index into
    // the by-value argument block would move the whole block to scratch)
    int k = 0, first = 0;
#pragma unroll
    for (int i = 1; i < INSTNORM_GROUP_MAX; ++i)
        if (i < g.n && (int)blockIdx.x >= g.start[i]) {
            k = i;
            first = g.start[i];
        }
    static_assert(__builtin_offsetof(InstNormGroupArgs, item) == 0, "item table at the start of the argument block");
    typedef __attribute__((address_space(4))) DvcInstNormItem ItemK;
    const ItemK& t = ((const ItemK*)__builtin_amdgcn_kernarg_segment_ptr())[k];
    const int p = blockIdx.x - first;  // n*C + c
    const int C = t.C, H = t.H, W = t.W, HW = H * W;
    const int n = p / C, c = p - n * C;
    const float* xp;
    if (t.S == 0) {
        xp = t.x + (long)n * t.x_batch_stride + (long)c * HW;
    } else {
        const int S = t.S;
        const long slab = (long)t.N * C * HW;
        const float* p0 = t.x + (long)p * HW;
        const float b = t.bias ? t.bias[c] : 0.f;
        const float aslope = t.act_slope_ptr ? *t.act_slope_ptr : t.act_slope;
        const int act = t.act;
        auto finish = [&](float v) {
            v += b;
            if (act == DVC_ACT_RELU) v = v > 0.f ? v : 0.f;
            else if (act == DVC_ACT_PRELU || act == DVC_ACT_LEAKY) v = v >= 0.f ? v : v * aslope;
            return v;
        };
        // (exactly instnorm_apply_partials_kernel's summation)
        const bool v4 = (HW % 4 == 0) && (slab % 4 == 0) && ((reinterpret_cast<uintptr_t>(t.x) & 15) == 0);
        const int HW4 = v4 ? HW : 0;
        for (int i = threadIdx.x * 4; i < HW4; i += blockDim.x * 4) {
            float4 tt[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) tt[q] = q < S ? *reinterpret_cast<const float4*>(p0 + (long)q * slab + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 8; ++q) { v.x += tt[q].x; v.y += tt[q].y; v.z += tt[q].z; v.w += tt[q].w; }
            *reinterpret_cast<float4*>(plane + i) = make_float4(finish(v.x), finish(v.y), finish(v.z), finish(v.w));
        }
        for (int i = HW4 + threadIdx.x; i < HW; i += blockDim.x) {
            float v = 0.f;
            for (int q = 0; q < S; ++q) v += p0[(long)q * slab + i];
            plane[i] = finish(v);
        }
        __syncthreads();
        xp = plane;
    }
    instnorm_apply_plane(xp, p, n, c, red, t.residual, t.slope_ptr, t.chan_scale, t.eps, C, H, W, t.up, t.sub, t.rpad,
                         t.res_batch_stride, t.y_batch_stride, t.y, nullptr, nullptr, nullptr, 1, 0, nullptr);
#endif
}

extern "C" int dvc_instnorm_apply_group(const DvcInstNormItem* items, int32_t n_items, dvcStream stream) {
    DVC_REQUIRE(items && n_items >= 1 && n_items <= INSTNORM_GROUP_MAX, "dvc_instnorm_apply_group: 1..%d items", INSTNORM_GROUP_MAX);
    InstNormGroupArgs g;
    g.n = n_items;
    g.start[0] = 0;
    size_t lds = 0;
    int block = 512;
    for (int i = 0; i < INSTNORM_GROUP_MAX; ++i) {
        const DvcInstNormItem& src = items[i < n_items ? i : n_items - 1];
        DvcInstNormItem& t = g.item[i];
        t = src;
        if (i >= n_items) {
            g.start[i + 1] = g.start[i];
            continue;
        }
        DVC_REQUIRE(t.x && t.y && t.S >= 0 && t.S <= 8 && t.N > 0 && t.C > 0 && t.H > 0 && t.W > 0,
                    "dvc_instnorm_apply_group: bad argument in item %d", i);
        DVC_REQUIRE(t.up >= 1 && t.up <= 4 && (t.sub == 1 || t.sub == 2) && t.rpad >= 0 && !(t.up != 1 && t.sub != 1),
                    "dvc_instnorm_apply_group: bad up/sub/rpad in item %d", i);
        DVC_REQUIRE(!(t.residual && (t.up != 1 || t.sub != 1)), "dvc_instnorm_apply_group: residual requires up == sub == 1 (item %d)", i);
        DVC_REQUIRE(!(t.S == 0 && t.x == t.y && (t.up != 1 || t.sub != 1 || t.rpad != 0)),
                    "dvc_instnorm_apply_group: in place needs up == sub == 1, rpad == 0 (item %d)", i);
        const long HW = (long)t.H * t.W;
        if (t.S >= 1) {
            DVC_REQUIRE(HW <= 16384, "dvc_instnorm_apply_group: plane of %d x %d does not fit the LDS image (item %d)", t.H, t.W, i);
            DVC_REQUIRE(t.act == DVC_ACT_NONE || t.act == DVC_ACT_RELU || t.act == DVC_ACT_PRELU || t.act == DVC_ACT_LEAKY,
                        "dvc_instnorm_apply_group: unsupported activation %d (item %d)", t.act, i);
            lds = std::max(lds, (size_t)HW * sizeof(float));
        }
        if (HW >= 4096) block = 1024;
        const long VH = t.sub == 2 ? (t.H + 1) / 2 : (long)t.H * t.up, VW = t.sub == 2 ? (t.W + 1) / 2 : (long)t.W * t.up;
        if (!t.x_batch_stride) t.x_batch_stride = (long)t.C * HW;
        if (!t.res_batch_stride) t.res_batch_stride = (long)t.C * HW;
        if (!t.y_batch_stride) t.y_batch_stride = (long)t.C * (VH + 2 * t.rpad) * VW;
        DVC_REQUIRE((long)g.start[i] + (long)t.N * t.C < (1L << 31), "dvc_instnorm_apply_group: too many planes");
        g.start[i + 1] = g.start[i] + t.N * t.C;
    }
    hipLaunchKernelGGL(instnorm_group_kernel, dim3(g.start[n_items]), dim3(block), lds, (hipStream_t)stream, g);
    DVC_CHECK_LAUNCH("dvc_instnorm_apply_group");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// y = prelu(x*scale + shift + residual), with optional nearest upsample and replicated row pad.
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ res,
                                                         const float* __restrict__ slope_ptr, int C,
                                                         int H, int W, int up, int rpad, long x_bs,
                                                         long res_bs, long y_bs, float* __restrict__ y) {
    const int OW = W * up, OH = H * up + 2 * rpad;
    const int p = blockIdx.y;  // n*C + c
    const int n = p / C, c = p - n * C;
    const float sc = scale ? scale[p] : 1.f, sh = shift ? shift[p] : 0.f;
    const bool has_act = slope_ptr != nullptr;
    const float slope = has_act ? *slope_ptr : 1.f;
    const float* xp = x + (long)n * x_bs + (long)c * H * W;
    const float* rp = res ? res + (long)n * res_bs + (long)c * H * W : nullptr;
    float* yp = y + (long)n * y_bs + (long)c * OH * OW;
    const int total = OH * OW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int oy = i / OW, ox = i - oy * OW;
        int uy = oy - rpad;
        uy = uy < 0 ? 0 : (uy >= H * up ? H * up - 1 : uy);
        int sy = up == 1 ? uy : uy / up, sx = up == 1 ? ox : ox / up;
        float v = xp[sy * W + sx] * sc + sh;
        if (rp) v += rp[sy * W + sx];
        if (has_act) v = v >= 0.f ? v : v * slope;
        yp[i] = v;
    }
}

extern "C" int dvc_affine_act(const float* x, const float* scale, const float* shift,
                              const float* residual, const float* slope_ptr, int32_t N, int32_t C,
                              int32_t H, int32_t W, int32_t up, int32_t rpad, int64_t x_batch_stride,
                              int64_t res_batch_stride, int64_t y_batch_stride, float* y,
                              dvcStream stream) {
    DVC_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, "dvc_affine_act: bad argument");
    DVC_REQUIRE(up >= 1 && up <= 4 && rpad >= 0, "dvc_affine_act: bad up/rpad");
    DVC_REQUIRE(!(residual && up != 1), "dvc_affine_act: residual requires up == 1");
    long OH = (long)H * up + 2 * rpad, OW = (long)W * up;
    long xbs = x_batch_stride ? x_batch_stride : (long)C * H * W;
    long rbs = res_batch_stride ? res_batch_stride : (long)C * H * W;
    long ybs = y_batch_stride ? y_batch_stride : (long)C * OH * OW;
    int bx = (int)((OH * OW + 1023) / 1024);
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(affine_act_kernel, dim3(bx, N * C), dim3(256), 0, (hipStream_t)stream, x, scale,
                       shift, residual, slope_ptr, C, H, W, up, rpad, xbs, rbs, ybs, y);
    DVC_CHECK_LAUNCH("dvc_affine_act");
    return 0;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ x, int H, int W,
                                                         int OH, int OW, float* __restrict__ y) {
    const float* xp = x + (long)blockIdx.y * H * W;
    float* yp = y + (long)blockIdx.y * OH * OW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < OH * OW; i += gridDim.x * 256) {
        int oy = i / OW, ox = i - oy * OW;
        const float* r0 = xp + (2 * oy) * W + 2 * ox;
        float2 a = *reinterpret_cast<const float2*>(r0);      // W even or ox<OW keeps this in range
        float2 b = *reinterpret_cast<const float2*>(r0 + W);
        yp[i] = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y));
    }
}
__global__ __launch_bounds__(256) void maxpool2x2_scalar_kernel(const float* __restrict__ x, int H, int W,
                                                                int OH, int OW, float* __restrict__ y) {
    const float* xp = x + (long)blockIdx.y * H * W;
    float* yp = y + (long)blockIdx.y * OH * OW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < OH * OW; i += gridDim.x * 256) {
        int oy = i / OW, ox = i - oy * OW;
        const float* r0 = xp + (2 * oy) * W + 2 * ox;
        yp[i] = fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[W], r0[W + 1]));
    }
}

extern "C" int dvc_maxpool2x2(const float* x, int32_t planes, int32_t H, int32_t W, float* y,
                              dvcStream stream) {
    DVC_REQUIRE(x && y && planes > 0 && H >= 2 && W >= 2, "dvc_maxpool2x2: bad argument");
    int OH = H / 2, OW = W / 2;
    dim3 grid(cdiv(OH * OW, 1024) > 0 ? cdiv(OH * OW, 1024) : 1, planes);
    bool vec = (W % 2 == 0) && ((reinterpret_cast<uintptr_t>(x) & 7) == 0) && ((H * W) % 2 == 0);
    if (vec)
        hipLaunchKernelGGL(maxpool2x2_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, H, W, OH, OW, y);
    else
        hipLaunchKernelGGL(maxpool2x2_scalar_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, H, W, OH, OW, y);
    DVC_CHECK_LAUNCH("dvc_maxpool2x2");
    return 0;
}

template <int K>
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, int H, int W, int OH,
                                                      int OW, float* __restrict__ y) {
    const float* xp = x + (long)blockIdx.y * H * W;
    float* yp = y + (long)blockIdx.y * OH * OW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < OH * OW; i += gridDim.x * 256) {
        int oy = i / OW, ox = i - oy * OW;
        const float* r0 = xp + (K * oy) * W + K * ox;
        float s = 0.f;
#pragma unroll
        for (int dy = 0; dy < K; ++dy)
#pragma unroll
            for (int dx = 0; dx < K; ++dx) s += r0[dy * W + dx];
        yp[i] = s * (1.f / (K * K));
    }
}

extern "C" int dvc_avgpool4x4(const float* x, int32_t planes, int32_t H, int32_t W, float* y,
                              dvcStream stream) {
    DVC_REQUIRE(x && y && planes > 0 && H >= 4 && W >= 4, "dvc_avgpool4x4: bad argument");
    int OH = H / 4, OW = W / 4;
    dim3 grid(cdiv(OH * OW, 256), planes);
    hipLaunchKernelGGL(avgpool_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, H, W, OH, OW, y);
    DVC_CHECK_LAUNCH("dvc_avgpool4x4");
    return 0;
}

extern "C" int dvc_avgpool2x2(const float* x, int32_t planes, int32_t H, int32_t W, float* y,
                              dvcStream stream) {
    DVC_REQUIRE(x && y && planes > 0 && H >= 2 && W >= 2, "dvc_avgpool2x2: bad argument");
    int OH = H / 2, OW = W / 2;
    dim3 grid(cdiv(OH * OW, 256), planes);
    hipLaunchKernelGGL(avgpool_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, H, W, OH, OW, y);
    DVC_CHECK_LAUNCH("dvc_avgpool2x2");
    return 0;
}

__global__ __launch_bounds__(256) void upsample_nearest_kernel(const float* __restrict__ x, int H, int W,
                                                               int f, float* __restrict__ y) {
    const int OH = H * f, OW = W * f;
    const float* xp = x + (long)blockIdx.y * H * W;
    float* yp = y + (long)blockIdx.y * OH * OW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < OH * OW; i += gridDim.x * 256) {
        int oy = i / OW, ox = i - oy * OW;
        yp[i] = xp[(oy / f) * W + ox / f];
    }
}

extern "C" int dvc_upsample_nearest(const float* x, int32_t planes, int32_t H, int32_t W, int32_t f,
                                    float* y, dvcStream stream) {
    DVC_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && f >= 1, "dvc_upsample_nearest: bad argument");
    dim3 grid(cdiv(H * f * W * f, 1024) > 0 ? cdiv(H * f * W * f, 1024) : 1, planes);
    hipLaunchKernelGGL(upsample_nearest_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, H, W, f, y);
    DVC_CHECK_LAUNCH("dvc_upsample_nearest");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// feature_normalize: per pixel, divide by the L2 norm over channels (scalar form for maps whose H*W is not a multiple of 4;
// the float4 form is channel_l2norm_multi_kernel below).
__global__ __launch_bounds__(256) void channel_l2norm_kernel(const float* __restrict__ x, int C, long HW,
                                                             float eps, float* __restrict__ y) {
    __shared__ float part[4][64];
    const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long p = (long)blockIdx.x * 64 + px;
    const int n = blockIdx.y;
    const float* xn = x + (long)n * C * HW;
    float* yn = y + (long)n * C * HW;
    const bool ok = p < HW;
    float s = 0.f;
    if (ok)
        for (int c = g; c < C; c += 4) {
            float v = xn[(long)c * HW + p];
            s = fmaf(v, v, s);
        }
    part[g][px] = s;
    __syncthreads();
    float tot = part[0][px] + part[1][px] + part[2][px] + part[3][px];
    float den = sqrtf(tot) + eps;
    if (ok)
        for (int c = g; c < C; c += 4) yn[(long)c * HW + p] = xn[(long)c * HW + p] / den;
}

// The four feature_normalize calls of a frame (FrameColor.py:16-23: relu2_1 .. relu5_1, 10.6 / 5.3 / 2.7 / 0.6 MB) as ONE
// launch: separately they are 324 / 81 / 21 / 5 workgroups on 256 CUs — latency-bound, 62 us for 38 MB of traffic.
// Workgroup = 32 pixels (8 lanes x float4: one 128-byte line per channel row) x 32 channel groups; the tensors' workgroup
// ranges are concatenated (wg_start), so the small maps ride along with the large one.
struct L2MultiArgs {
    const float* x[DVC_L2NORM_MAX_TENSORS];
    float* y[DVC_L2NORM_MAX_TENSORS];
    int C[DVC_L2NORM_MAX_TENSORS], HW[DVC_L2NORM_MAX_TENSORS], wg_start[DVC_L2NORM_MAX_TENSORS + 1];
    int count;
    float eps;
};
__global__ __launch_bounds__(256) void channel_l2norm_multi_kernel(L2MultiArgs a) {
    __shared__ float4 part[32][8];
    int ti = 0;
#pragma unroll
    for (int k = 1; k < DVC_L2NORM_MAX_TENSORS; ++k) ti += (k < a.count && (int)blockIdx.x >= a.wg_start[k]) ? 1 : 0;
    const int C = a.C[ti];
    const long HW = a.HW[ti];
    const int px4 = threadIdx.x & 7, g = threadIdx.x >> 3;
    const long p = ((long)(blockIdx.x - a.wg_start[ti]) * 8 + px4) * 4;
    const int n = blockIdx.y;
    const float* xn = a.x[ti] + (long)n * C * HW;
    float* yn = a.y[ti] + (long)n * C * HW;
    const bool ok = p < HW;  // HW % 4 == 0
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        int c = g;
        // (r06: four channel rows in flight per thread, accumulated in the same order as one at a time)
        for (; c + 96 < C; c += 128) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(xn + (long)(c + 32 * u) * HW + p);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s.x = fmaf(v[u].x, v[u].x, s.x);
                s.y = fmaf(v[u].y, v[u].y, s.y);
                s.z = fmaf(v[u].z, v[u].z, s.z);
                s.w = fmaf(v[u].w, v[u].w, s.w);
            }
        }
        for (; c < C; c += 32) {
            const float4 v = *reinterpret_cast<const float4*>(xn + (long)c * HW + p);
            s.x = fmaf(v.x, v.x, s.x);
            s.y = fmaf(v.y, v.y, s.y);
            s.z = fmaf(v.z, v.z, s.z);
            s.w = fmaf(v.w, v.w, s.w);
        }
    }
    part[g][px4] = s;
    __syncthreads();
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const float4 v = part[k][px4];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    const float4 den = make_float4(sqrtf(t.x) + a.eps, sqrtf(t.y) + a.eps, sqrtf(t.z) + a.eps, sqrtf(t.w) + a.eps);
    if (ok) {
        int c = g;
        for (; c + 96 < C; c += 128) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(xn + (long)(c + 32 * u) * HW + p);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u].x /= den.x; v[u].y /= den.y; v[u].z /= den.z; v[u].w /= den.w;
                *reinterpret_cast<float4*>(yn + (long)(c + 32 * u) * HW + p) = v[u];
            }
        }
        for (; c < C; c += 32) {
            float4 v = *reinterpret_cast<const float4*>(xn + (long)c * HW + p);
            v.x /= den.x; v.y /= den.y; v.z /= den.z; v.w /= den.w;
            *reinterpret_cast<float4*>(yn + (long)c * HW + p) = v;
        }
    }
}

extern "C" int dvc_channel_l2norm_multi(const float* const* x, float* const* y, const int32_t* C, const int32_t* HW,
                                        int32_t count, int32_t N, float eps, dvcStream stream) {
    DVC_REQUIRE(x && y && C && HW && count > 0 && count <= DVC_L2NORM_MAX_TENSORS && N > 0, "dvc_channel_l2norm_multi: bad argument");
    L2MultiArgs a;
    a.count = count;
    a.eps = eps;
    int wgs = 0;
    for (int i = 0; i < DVC_L2NORM_MAX_TENSORS; ++i) {
        const int k = i < count ? i : count - 1;
        a.x[i] = x[k]; a.y[i] = y[k]; a.C[i] = C[k]; a.HW[i] = HW[k];
        a.wg_start[i] = wgs;
        if (i < count) {
            DVC_REQUIRE(x[i] && y[i] && C[i] > 0 && HW[i] > 0, "dvc_channel_l2norm_multi: bad tensor %d", i);
            DVC_REQUIRE(HW[i] % 4 == 0 && (((reinterpret_cast<uintptr_t>(x[i]) | reinterpret_cast<uintptr_t>(y[i])) & 15) == 0),
                        "dvc_channel_l2norm_multi: tensor %d needs H*W %% 4 == 0 and 16-byte aligned pointers (use dvc_channel_l2norm)", i);
            wgs += cdiv(HW[i], 32);
        }
    }
    a.wg_start[DVC_L2NORM_MAX_TENSORS] = wgs;
    hipLaunchKernelGGL(channel_l2norm_multi_kernel, dim3(wgs, N), dim3(256), 0, (hipStream_t)stream, a);
    DVC_CHECK_LAUNCH("dvc_channel_l2norm_multi");
    return 0;
}

extern "C" int dvc_channel_l2norm(const float* x, int32_t N, int32_t C, int32_t HW, float eps, float* y,
                                  dvcStream stream) {
    DVC_REQUIRE(x && y && N > 0 && C > 0 && HW > 0, "dvc_channel_l2norm: bad argument");
    dim3 grid(cdiv(HW, 64), N);
    const bool v4 = (HW % 4 == 0) && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
    // (the float4 case IS the multi-tensor kernel with one tensor: a map normalised alone and the same map normalised in a
    // group of four give the same bits)
    if (v4) return dvc_channel_l2norm_multi(&x, &y, &C, &HW, 1, N, eps, stream);
    hipLaunchKernelGGL(channel_l2norm_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, C, (long)HW, eps, y);
    DVC_CHECK_LAUNCH("dvc_channel_l2norm");
    return 0;
}
