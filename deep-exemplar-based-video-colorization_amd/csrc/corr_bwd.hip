// Backward of the fused correlation for the training-side callers (train.py:402-427 runs frame_colorization with
// autograd on, soft temperature 0.01): gradients of
//     f = theta^T phi,  p = softmax_j(f / T),  y = p . B_lab,  sim = max_j f          (models/NonlocalNet.py:477-500)
// with respect to theta and phi.  As in the forward pass the P x P matrices never exist: the host walks the query rows
// in blocks of R rows; per block
//     F  = theta_blk^T phi                    R x P   (1x1 convolution engine)
//     dS = p * (g . B_j - g . y_i) / T  (+ the sim gradient at the row arg-max),  p = exp(F/T - m_i) / l_i     [here]
//     d theta_blk = phi dS^T,   d phi += theta_blk dS                              (1x1 convolution engine)
// This file holds the two bandwidth-bound kernels of the middle step: the row sums l_i, and dS written in both
// layouts (row-major for d phi, transposed for d theta — the engine wants K-major operands).
#include "common.h"

#include <cmath>

// m_i = max_j fl32(F_ij / T),  l_i = sum_j exp(fl32(F_ij / T) - m_i): ATen's softmax arithmetic (true division, exp of the
// difference to the row maximum), one workgroup per row.  The maximum is taken over the block as RECOMPUTED here (the
// forward kernel's similarity comes from another summation order and need not bound it: at T <= 1e-7 an excess of one
// ulp would overflow the exponential).
// r05: WTA_scale (models/NonlocalNet.py:288-327; `WTA_scale_weight != 1`): forward f' = (f == max_j f) ? f : f * scale ahead of
// the temperature, backward grad_f = grad_f' * ((f == max_j f) ? 1 : 1e-4) — the reference's backward uses the constant 1e-4
// whatever `scale` is (NonlocalNet.py:322), and so does this.  The similarity map is taken from f BEFORE the re-weighting
// (NonlocalNet.py:481-483), so its gradient is not scaled.  wta == 1: the plain path, unchanged.
__device__ __forceinline__ float wta_apply(float f, float fmax_raw, float wta) { return (wta == 1.f || f == fmax_raw) ? f : f * wta; }

__global__ __launch_bounds__(256) void corr_bwd_rowstat_kernel(const float* __restrict__ F, float T, int P, int ld, float wta,
                                                               float* __restrict__ m_out, float* __restrict__ l_out,
                                                               float* __restrict__ raw_out) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    // blockIdx.y = image of the batch (r05): F [B][ld][P], the three row-statistics arrays [B][3][ld]
    F += (long)blockIdx.y * ld * P;
    m_out += (long)blockIdx.y * 3 * ld;
    l_out += (long)blockIdx.y * 3 * ld;
    raw_out += (long)blockIdx.y * 3 * ld;
    const float* f = F + (long)row * P;
    float raw = -INFINITY;
    if (wta != 1.f) {       // (block-uniform) raw row maximum first: it decides which elements are re-weighted
        for (int j = threadIdx.x; j < P; j += 256) raw = fmaxf(raw, f[j]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) raw = fmaxf(raw, __shfl_xor(raw, off, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = raw;
        __syncthreads();
        raw = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
    }
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < P; j += 256) mx = fmaxf(mx, wta_apply(f[j], raw, wta) / T);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int j = threadIdx.x; j < P; j += 256) s += expf(wta_apply(f[j], raw, wta) / T - m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        m_out[row] = m;
        l_out[row] = (red[0] + red[1]) + (red[2] + red[3]);
        raw_out[row] = raw;
    }
}

// dS tile 64 rows x 64 columns: computed once, written row-major (coalesced along j) and, through an LDS transpose,
// column-major (coalesced along i).
__global__ __launch_bounds__(256) void corr_bwd_ds_kernel(const float* __restrict__ F, const float* __restrict__ blab,
                                                          const float* __restrict__ gy, const float* __restrict__ y,
                                                          const float* __restrict__ rowmax, const float* __restrict__ gsim,
                                                          const int* __restrict__ amax, const float* __restrict__ lsum,
                                                          const float* __restrict__ rawmax, float wta,
                                                          float T, int rows, int P, long cs, int ldt,
                                                          float* __restrict__ dS, float* __restrict__ dST) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    {   // blockIdx.z = image of the batch (r05): dense batch strides
        const long bz = blockIdx.z;
        F += bz * ldt * P;
        dS += bz * ldt * P;
        if (dST) dST += bz * (long)P * ldt;
        blab += bz * 3 * P;
        gy += bz * 3 * cs;
        y += bz * 3 * cs;
        rowmax += bz * 3 * ldt;
        lsum += bz * 3 * ldt;
        rawmax += bz * 3 * ldt;
        if (gsim) {
            gsim += bz * cs;
            amax += bz * cs;
        }
    }
    const int j = j0 + tx;
    const bool jok = j < P;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;
    if (jok) {
        b0 = blab[j];
        b1 = blab[(long)P + j];
        b2 = blab[2L * P + j];
    }
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int il = ty * 16 + r, i = i0 + il;
        float v = 0.f;
        if (i < rows && jok) {
            const float g0 = gy[i], g1 = gy[cs + i], g2 = gy[2 * cs + i];
            const float delta = g0 * y[i] + g1 * y[cs + i] + g2 * y[2 * cs + i];
            const float fr = F[(long)i * P + j];
            const float p = expf(wta_apply(fr, rawmax[i], wta) / T - rowmax[i]) / lsum[i];
            v = p * ((g0 * b0 + g1 * b1 + g2 * b2) - delta) / T;
            if (wta != 1.f && fr != rawmax[i]) v *= 1e-4f;      // WTA_scale.backward's constant (NonlocalNet.py:322)
            if (gsim && amax[i] == j) v += gsim[i];
            dS[(long)i * P + j] = v;
        }
        tile[il][tx] = v;
    }
    if (!dST) return;           // (uniform: the caller multiplies with dS^T through a GEMM that takes the operand transposed)
    __syncthreads();
    // transposed copy: thread (tx -> row i0 + tx, ty -> 16 columns)
    const int i = i0 + tx;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int jl = ty * 16 + r, jj = j0 + jl;
        if (jj < P && i < ldt) dST[(long)jj * ldt + i] = (i < rows) ? tile[tx][jl] : 0.f;
    }
}

extern "C" int dvc_corr_softmax_bwd(const float* f_blk, const float* blab, const float* gy, const float* y,
                                    const float* sim, const float* gsim, const int32_t* argmax, float temperature,
                                    float wta_scale, int32_t batch, int32_t rows, int32_t P, int64_t chan_stride, int32_t ld_t,
                                    float* rowstat_scratch, float* dS, float* dST, dvcStream stream) {
    DVC_REQUIRE(f_blk && blab && gy && y && rowstat_scratch && dS, "dvc_corr_softmax_bwd: null argument");
    (void)sim;
    float* rowmax = rowstat_scratch;
    float* lsum_scratch = rowstat_scratch + ld_t;
    float* rawmax = rowstat_scratch + 2 * (size_t)ld_t;
    DVC_REQUIRE(std::isfinite(wta_scale), "dvc_corr_softmax_bwd: bad wta_scale");
    DVC_REQUIRE(rows > 0 && P > 0 && ld_t >= rows && batch >= 1 && batch <= 65535, "dvc_corr_softmax_bwd: bad shape");
    DVC_REQUIRE(temperature > 0.f && std::isfinite(temperature), "dvc_corr_softmax_bwd: temperature must be > 0");
    DVC_REQUIRE((gsim == nullptr) == (argmax == nullptr), "dvc_corr_softmax_bwd: gsim and argmax come together");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(corr_bwd_rowstat_kernel, dim3(rows, batch), dim3(256), 0, s, f_blk, temperature, P, ld_t, wta_scale, rowmax, lsum_scratch, rawmax);
    DVC_CHECK_LAUNCH("dvc_corr_softmax_bwd(rowsum)");
    hipLaunchKernelGGL(corr_bwd_ds_kernel, dim3(cdiv(P, 64), cdiv(ld_t, 64), batch), dim3(256), 0, s, f_blk, blab, gy, y, rowmax, gsim,
                       argmax, lsum_scratch, rawmax, wta_scale, temperature, rows, P, (long)chan_stride, ld_t, dS, dST);
    DVC_CHECK_LAUNCH("dvc_corr_softmax_bwd(dS)");
    return 0;
}
