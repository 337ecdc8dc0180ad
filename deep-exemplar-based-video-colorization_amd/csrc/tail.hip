// Clip-driver tail (/root/reference/test.py:98-116): x2 bilinear upsample of the
// predicted ab (* 1.25), 8-bit luminance guide, fast global smoother (WLS, cv2.ximgproc in the reference)
// and Lab -> 8-bit RGB.  All HBM/latency-bound: coalescing and enough independent solves in flight are what
// matter; nothing here is GEMM-shaped.
#include "common.h"

// ---- F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) * mul  (test.py:100-102)
// ATen: src = max(0, 0.5*(dst+0.5) - 0.5), i0 = floor(src), i1 = min(i0+1, S-1), w1 = src - i0, w0 = 1 - w1,
// and its compiled kernel evaluates w0*a + w1*b as fma(w0, a, fl32(w1*b)) — reproduced bit for bit.
__device__ __forceinline__ float lerp_aten(float w0, float a, float w1, float b) {
    return __fmaf_rn(w0, a, __fmul_rn(w1, b));
}
__global__ __launch_bounds__(256) void bilinear2x_kernel(const float* __restrict__ x, int H, int W, float mul,
                                                         float* __restrict__ y) {
    const int OW = 2 * W, OH = 2 * H;
    const float* xp = x + (long)blockIdx.y * H * W;
    float* yp = y + (long)blockIdx.y * OH * OW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < OH * OW; i += gridDim.x * 256) {
        const int oy = i / OW, ox = i - oy * OW;
        const float sy = fmaxf(__fsub_rn(__fmul_rn(0.5f, __fadd_rn((float)oy, 0.5f)), 0.5f), 0.f);
        const float sx = fmaxf(__fsub_rn(__fmul_rn(0.5f, __fadd_rn((float)ox, 0.5f)), 0.5f), 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float wy1 = __fsub_rn(sy, (float)y0), wx1 = __fsub_rn(sx, (float)x0);
        const float wy0 = __fsub_rn(1.f, wy1), wx0 = __fsub_rn(1.f, wx1);
        const float top = lerp_aten(wx0, xp[y0 * W + x0], wx1, xp[y0 * W + x1]);
        const float bot = lerp_aten(wx0, xp[y1 * W + x0], wx1, xp[y1 * W + x1]);
        yp[i] = __fmul_rn(lerp_aten(wy0, top, wy1, bot), mul);
    }
}

extern "C" int dvc_upsample_bilinear2x(const float* x, int32_t planes, int32_t H, int32_t W, float mul, float* y,
                                       dvcStream stream) {
    DVC_REQUIRE(x && y && planes > 0 && H > 0 && W > 0, "dvc_upsample_bilinear2x: bad argument");
    DVC_REQUIRE((long)4 * H * W < (1L << 31), "dvc_upsample_bilinear2x: plane too large");
    hipLaunchKernelGGL(bilinear2x_kernel, dim3(cdiv(4 * H * W, 1024), planes), dim3(256), 0, (hipStream_t)stream,
                       x, H, W, mul, y);
    DVC_CHECK_LAUNCH("dvc_upsample_bilinear2x");
    return 0;
}

// ---- (uncenter_l(L) * 255 / 100).astype(uint8)  (test.py:106-109)
__global__ __launch_bounds__(256) void lum_guide_kernel(const float* __restrict__ L, long n,
                                                        unsigned char* __restrict__ g) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = __fdiv_rn(__fmul_rn(__fadd_rn(L[i], 50.f), 255.f), 100.f);
        g[i] = (unsigned char)(int)v;   // numpy astype(uint8): truncation (wraps modulo 256 outside [0,256))
    }
}
extern "C" int dvc_lum_guide_u8(const float* L_centered, int64_t n, uint8_t* guide, dvcStream stream) {
    DVC_REQUIRE(L_centered && guide && n > 0, "dvc_lum_guide_u8: bad argument");
    hipLaunchKernelGGL(lum_guide_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream,
                       L_centered, (long)n, guide);
    DVC_CHECK_LAUNCH("dvc_lum_guide_u8");
    return 0;
}

// ---- fast global smoother: D. Min et al., "Fast Global Image Smoothing Based on Weighted Least Squares",
// IEEE TIP 2014, Algorithm 1 (the filter cv2.ximgproc.createFastGlobalSmootherFilter implements).
// T iterations of { 1-D WLS solve along every row ; along every column } with
//   lambda_t = 1.5 * 4^(T-t) / (4^T - 1) * lambda,   w(p,q) = exp(-|g_p - g_q| / sigma_color)  (8-bit guide),
// each 1-D solve a tridiagonal system (a_x = -lambda w(x-1,x), c_x = -lambda w(x,x+1), b_x = 1 - a_x - c_x),
// Thomas algorithm in float32 (with the reciprocal of the pivot, see fgs_coeff_kernel).
// Mapping: one thread per LINE; consecutive threads own consecutive lines and march along the other axis, so
// every load / store of the column solve is a coalesced row.  The row solve runs on the transposed image (a
// 32x32 LDS-tile transpose before and after).  The sweeps are latency chains (one divide + two fma per
// element); planes x lines threads (2 x 768 / 2 x 432 at 432x768) are all the parallelism the algorithm has.
__global__ __launch_bounds__(256) void fgs_weights_kernel(const unsigned char* __restrict__ g, int H, int W,
                                                          float inv_sigma, float* __restrict__ wv,
                                                          float* __restrict__ wh_t) {
    // wv[y][x]   = w((y,x),(y+1,x))   [H][W]   (last row unused)
    // wh_t[x][y] = w((y,x),(y,x+1))   [W][H]   (last row unused) — the horizontal weights, transposed
    g += (long)blockIdx.y * H * W;
    wv += (long)blockIdx.y * H * W;
    wh_t += (long)blockIdx.y * H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        const int c = g[i];
        const int dn = y + 1 < H ? abs(c - (int)g[i + W]) : 0;
        const int rt = x + 1 < W ? abs(c - (int)g[i + 1]) : 0;
        wv[i] = expf(-(float)dn * inv_sigma);
        wh_t[(long)x * H + y] = expf(-(float)rt * inv_sigma);
    }
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, int H, int W,
                                                        float* __restrict__ y) {
    __shared__ float t[32][33];
    const float* xp = x + (long)blockIdx.z * H * W;
    float* yp = y + (long)blockIdx.z * H * W;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (by + r < H && bx + tx < W) t[r][tx] = xp[(long)(by + r) * W + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (bx + r < W && by + tx < H) yp[(long)(bx + r) * H + by + tx] = t[tx][r];
}

// The tridiagonal systems depend on the guide and lambda only, so their elimination coefficients are computed
// once per (direction, iteration) and shared by all planes of a guide:
//   m_l = b_l - a_l c'_{l-1},  inv_l = 1 / m_l,  c'_l = c_l inv_l            (one divide per element, here)
//   d'_l = (f_l - a_l d'_{l-1}) inv_l = fma(-(a_l inv_l), d'_{l-1}, f_l inv_l) (one dependent fma per element, below)
// Both kernels: one thread per line, FGS_U elements fetched ahead of the dependent chain.
#define FGS_U 16
__global__ __launch_bounds__(64) void fgs_coeff_kernel(const float* __restrict__ w, int L, int M, float lambda,
                                                       float* __restrict__ cp, float* __restrict__ inv) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= M) return;
    const long off = (long)blockIdx.y * L * M + m;       // blockIdx.y = guide
    const float* wp = w + off;
    float* cpp = cp + off;
    float* ivp = inv + off;
    float a = 0.f;
    float c = L > 1 ? -lambda * wp[0] : 0.f;
    float iv = 1.f / (1.f - a - c);
    float cprev = c * iv;
    cpp[0] = cprev;
    ivp[0] = iv;
    int l0 = 1;
    for (; l0 + FGS_U < L; l0 += FGS_U) {            // full blocks (never contain the last element): no conditions
        float wk[FGS_U];
#pragma unroll
        for (int k = 0; k < FGS_U; ++k) wk[k] = wp[(long)(l0 + k) * M];
#pragma unroll
        for (int k = 0; k < FGS_U; ++k) {
            a = c;                                       // a_l = -lambda w(l-1,l) = c_{l-1}
            c = -lambda * wk[k];
            iv = 1.f / ((1.f - a - c) - a * cprev);
            cprev = c * iv;
            cpp[(long)(l0 + k) * M] = cprev;
            ivp[(long)(l0 + k) * M] = iv;
        }
    }
    for (int l = l0; l < L; ++l) {
        a = c;
        c = l + 1 < L ? -lambda * wp[(long)l * M] : 0.f;
        iv = 1.f / ((1.f - a - c) - a * cprev);
        cprev = c * iv;
        cpp[(long)l * M] = cprev;
        ivp[(long)l * M] = iv;
    }
}

// solve along axis 0 of f [planes][L][M] in place (planes g*ppg .. use guide g's coefficients); dp: scratch
__global__ __launch_bounds__(64) void fgs_solve_kernel(float* __restrict__ f, const float* __restrict__ w,
                                                       const float* __restrict__ cp, const float* __restrict__ inv,
                                                       int L, int M, int planes_per_guide, float lambda,
                                                       float* __restrict__ dp) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= M) return;
    const long goff = (long)(blockIdx.y / planes_per_guide) * L * M + m;
    float* fp = f + (long)blockIdx.y * L * M + m;
    float* dpp = dp + (long)blockIdx.y * L * M + m;
    const float* wp = w + goff;
    const float* cpp = cp + goff;
    const float* ivp = inv + goff;
    float dprev = fp[0] * ivp[0];
    dpp[0] = dprev;
    int l0 = 1;
    for (; l0 + FGS_U <= L; l0 += FGS_U) {           // full blocks: no conditions, 48 independent loads in flight
        float ak[FGS_U], fk[FGS_U];
#pragma unroll
        for (int k = 0; k < FGS_U; ++k) {
            const float iv = ivp[(long)(l0 + k) * M];
            ak[k] = lambda * wp[(long)(l0 + k - 1) * M] * iv;              // -(a_l inv_l), a_l = -lambda w(l-1,l)
            fk[k] = fp[(long)(l0 + k) * M] * iv;
        }
#pragma unroll
        for (int k = 0; k < FGS_U; ++k) {
            dprev = fmaf(ak[k], dprev, fk[k]);
            dpp[(long)(l0 + k) * M] = dprev;
        }
    }
    for (int l = l0; l < L; ++l) {
        const float iv = ivp[(long)l * M];
        dprev = fmaf(lambda * wp[(long)(l - 1) * M] * iv, dprev, fp[(long)l * M] * iv);
        dpp[(long)l * M] = dprev;
    }
    float u = dprev;
    fp[(long)(L - 1) * M] = u;
    int l1 = L - 2;
    for (; l1 - (FGS_U - 1) >= 0; l1 -= FGS_U) {
        float ck[FGS_U], dk[FGS_U];
#pragma unroll
        for (int k = 0; k < FGS_U; ++k) {
            ck[k] = cpp[(long)(l1 - k) * M];
            dk[k] = dpp[(long)(l1 - k) * M];
        }
#pragma unroll
        for (int k = 0; k < FGS_U; ++k) {
            u = fmaf(-ck[k], u, dk[k]);
            fp[(long)(l1 - k) * M] = u;
        }
    }
    for (int l = l1; l >= 0; --l) {
        u = fmaf(-cpp[(long)l * M], u, dpp[(long)l * M]);
        fp[(long)l * M] = u;
    }
}

extern "C" size_t dvc_fgs_workspace_bytes(int32_t H, int32_t W, int32_t n_guides, int32_t planes_per_guide) {
    // per guide: wv, wh_t and (c', 1/m) of the row and of the column system; per plane: transposed image + d'
    return sizeof(float) * ((size_t)6 * n_guides * H * W + (size_t)2 * n_guides * planes_per_guide * H * W);
}

extern "C" int dvc_fgs_filter(const uint8_t* guide, const float* src, int32_t n_guides, int32_t planes_per_guide,
                              int32_t H, int32_t W, float lambda, float sigma_color, int32_t num_iter,
                              float lambda_attenuation, float* dst, void* workspace, size_t workspace_bytes,
                              dvcStream stream) {
    DVC_REQUIRE(guide && src && dst && workspace && n_guides > 0 && planes_per_guide > 0 && H > 0 && W > 0,
                "dvc_fgs_filter: bad argument");
    const int planes = n_guides * planes_per_guide;
    DVC_REQUIRE(num_iter >= 1 && num_iter <= 8 && sigma_color > 0.f && lambda >= 0.f, "dvc_fgs_filter: bad parameters");
    DVC_REQUIRE(workspace_bytes >= dvc_fgs_workspace_bytes(H, W, n_guides, planes_per_guide),
                "dvc_fgs_filter: workspace too small");
    DVC_REQUIRE((long)H * W < (1L << 30), "dvc_fgs_filter: image too large");
    hipStream_t s = (hipStream_t)stream;
    const size_t HW = (size_t)H * W, GHW = (size_t)n_guides * HW;
    float* wv = reinterpret_cast<float*>(workspace);
    float* wh_t = wv + GHW;
    float* cp_r = wh_t + GHW;   // row system (lines of length W, on the transposed image)
    float* iv_r = cp_r + GHW;
    float* cp_c = iv_r + GHW;   // column system
    float* iv_c = cp_c + GHW;
    float* tr = iv_c + GHW;
    float* dp = tr + planes * HW;
    hipLaunchKernelGGL(fgs_weights_kernel, dim3(cdiv((int)HW, 1024), n_guides), dim3(256), 0, s, guide, H, W,
                       1.0f / sigma_color, wv, wh_t);
    DVC_CHECK_LAUNCH("dvc_fgs_filter(weights)");
    if (dst != src) {
        hipError_t e = hipMemcpyAsync(dst, src, sizeof(float) * planes * HW, hipMemcpyDeviceToDevice, s);
        DVC_REQUIRE(e == hipSuccess, "dvc_fgs_filter: copy failed: %s", hipGetErrorString(e));
    }
    double lam = 1.5 * (double)lambda * pow(4.0, num_iter - 1) / (pow(4.0, num_iter) - 1.0);
    float lam_f = (float)lam;
    const dim3 tgrid_fwd(cdiv(W, 32), cdiv(H, 32), planes), tgrid_bwd(cdiv(H, 32), cdiv(W, 32), planes);
    for (int it = 0; it < num_iter; ++it) {
        hipLaunchKernelGGL(fgs_coeff_kernel, dim3(cdiv(H, 64), n_guides), dim3(64), 0, s, wh_t, W, H, lam_f, cp_r, iv_r);
        hipLaunchKernelGGL(fgs_coeff_kernel, dim3(cdiv(W, 64), n_guides), dim3(64), 0, s, wv, H, W, lam_f, cp_c, iv_c);
        // rows: every image row is a line of length W; in the transposed image [W][H] it runs along axis 0
        hipLaunchKernelGGL(transpose_kernel, tgrid_fwd, dim3(256), 0, s, dst, H, W, tr);
        hipLaunchKernelGGL(fgs_solve_kernel, dim3(cdiv(H, 64), planes), dim3(64), 0, s, tr, wh_t, cp_r, iv_r, W, H,
                           planes_per_guide, lam_f, dp);
        hipLaunchKernelGGL(transpose_kernel, tgrid_bwd, dim3(256), 0, s, tr, W, H, dst);
        // columns
        hipLaunchKernelGGL(fgs_solve_kernel, dim3(cdiv(W, 64), planes), dim3(64), 0, s, dst, wv, cp_c, iv_c, H, W,
                           planes_per_guide, lam_f, dp);
        DVC_CHECK_LAUNCH("dvc_fgs_filter(solve)");
        lam_f = lam_f * lambda_attenuation;
    }
    return 0;
}

// ---- batch_lab2rgb_transpose_mc (utils/util.py:134-151) for one image: skimage.color.lab2rgb in float64,
// clip, * 255, astype(uint8); output HWC.
__global__ __launch_bounds__(256) void lab2rgb_u8_kernel(const float* __restrict__ L, const float* __restrict__ ab,
                                                         long HW, unsigned char* __restrict__ rgb) {
    // inverse of skimage's xyz_from_rgb (its rgb_from_xyz = scipy.linalg.inv(xyz_from_rgb), float64)
    const double M[3][3] = {{3.240481343200526, -1.5371515162713185, -0.4985363261688878},
                            {-0.9692549499965682, 1.8759900014898907, 0.04155592655829284},
                            {0.05564663913517716, -0.20404133836651123, 1.0573110696453443}};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        const double l = (double)L[i] + 50.0, a = (double)ab[i], b = (double)ab[HW + i];
        const double fy = (l + 16.0) / 116.0;
        const double fx = a / 500.0 + fy;
        double fz = fy - b / 200.0;
        fz = fz < 0.0 ? 0.0 : fz;
        double xyz[3] = {fx, fy, fz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = xyz[k];
            xyz[k] = v > 0.2068966 ? v * v * v : (v - 16.0 / 116.0) / 7.787;
        }
        xyz[0] *= 0.95047;
        xyz[2] *= 1.08883;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double v = xyz[0] * M[k][0] + xyz[1] * M[k][1] + xyz[2] * M[k][2];
            v = v > 0.0031308 ? 1.055 * pow(v, 1.0 / 2.4) - 0.055 : v * 12.92;
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            rgb[3 * i + k] = (unsigned char)(int)(v * 255.0);
        }
    }
}
extern "C" int dvc_lab2rgb_u8(const float* L_centered, const float* ab, int32_t H, int32_t W, uint8_t* rgb_hwc,
                              dvcStream stream) {
    DVC_REQUIRE(L_centered && ab && rgb_hwc && H > 0 && W > 0, "dvc_lab2rgb_u8: bad argument");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(lab2rgb_u8_kernel, dim3((unsigned)((HW + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream,
                       L_centered, ab, HW, rgb_hwc);
    DVC_CHECK_LAUNCH("dvc_lab2rgb_u8");
    return 0;
}

// ---- frame ingest, colour part: RGB2Lab() -> ToTensor() -> Normalize()
// (utils/util_distortion.py:18-23,85-100, lib/functional.py:85-103): skimage.color.rgb2lab in float64 on the
// 8-bit image, .float(), L - 50.  Output [3][H][W] float32.  (The geometric part — CenterPad / CenterCrop with
// skimage's anti-aliased resize — stays on the host.)
__global__ __launch_bounds__(256) void rgb8_to_lab_kernel(const unsigned char* __restrict__ rgb, long HW,
                                                          float* __restrict__ lab) {
    const double M[3][3] = {{0.412453, 0.357580, 0.180423},
                            {0.212671, 0.715160, 0.072169},
                            {0.019334, 0.119193, 0.950227}};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        double c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = (double)rgb[3 * i + k] / 255.0;
            c[k] = v > 0.04045 ? pow((v + 0.055) / 1.055, 2.4) : v / 12.92;
        }
        double xyz[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) xyz[k] = c[0] * M[k][0] + c[1] * M[k][1] + c[2] * M[k][2];
        xyz[0] /= 0.95047;
        xyz[2] /= 1.08883;
#pragma unroll
        for (int k = 0; k < 3; ++k) xyz[k] = xyz[k] > 0.008856 ? cbrt(xyz[k]) : 7.787 * xyz[k] + 16.0 / 116.0;
        const float L = (float)(116.0 * xyz[1] - 16.0);
        lab[i] = L - 50.0f;                                   // Normalize(): (L - 50) / 1 in float32
        lab[HW + i] = (float)(500.0 * (xyz[0] - xyz[1]));
        lab[2 * HW + i] = (float)(200.0 * (xyz[1] - xyz[2]));
    }
}
extern "C" int dvc_rgb8_to_lab(const uint8_t* rgb_hwc, int32_t H, int32_t W, float* lab, dvcStream stream) {
    DVC_REQUIRE(rgb_hwc && lab && H > 0 && W > 0, "dvc_rgb8_to_lab: bad argument");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(rgb8_to_lab_kernel, dim3((unsigned)((HW + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream,
                       rgb_hwc, HW, lab);
    DVC_CHECK_LAUNCH("dvc_rgb8_to_lab");
    return 0;
}
