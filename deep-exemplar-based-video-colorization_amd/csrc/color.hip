// Elementwise colour-space / glue kernels on the hot path (all trivially HBM-bound).
#include "common.h"

// gray2rgb_batch, utils/util.py:97-101: y[n, 0..2] = (L + 50) / 100
__global__ __launch_bounds__(256) void gray2rgb_kernel(const float* __restrict__ l, long HW, long l_bs,
                                                       float* __restrict__ y) {
    const int n = blockIdx.y;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        float v = (l[(long)n * l_bs + i] * 1.0f + 50.0f) / 100.0f;
        float* yn = y + (long)n * 3 * HW + i;
        yn[0] = v;
        yn[HW] = v;
        yn[2 * HW] = v;
    }
}

extern "C" int dvc_gray2rgb(const float* l, int32_t N, int32_t HW, int64_t l_batch_stride, float* y,
                            dvcStream stream) {
    DVC_REQUIRE(l && y && N > 0 && HW > 0, "dvc_gray2rgb: bad argument");
    long bs = l_batch_stride ? l_batch_stride : HW;
    hipLaunchKernelGGL(gray2rgb_kernel, dim3(cdiv(HW, 1024), N), dim3(256), 0, (hipStream_t)stream, l,
                       (long)HW, bs, y);
    DVC_CHECK_LAUNCH("dvc_gray2rgb");
    return 0;
}

// tensor_lab2rgb, utils/util.py:379-414 (thresholds and constants as in the reference).
__global__ __launch_bounds__(256) void lab2rgb_kernel(const float* __restrict__ lab, long HW,
                                                      float l_offset, float* __restrict__ rgb) {
    const int n = blockIdx.y;
    const float* in = lab + (long)n * 3 * HW;
    float* out = rgb + (long)n * 3 * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        float L = in[i] + l_offset, a = in[HW + i], b = in[2 * HW + i];
        float fy = (L + 16.0f) / 116.0f;
        float fx = (a / 500.0f) + fy;
        float fz = fy - (b / 200.0f);
        fz = fz < 0.f ? 0.f : fz;
        float xyz[3] = {fx, fy, fz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = xyz[k];
            xyz[k] = v > 0.2068966f ? v * v * v : (v - 16.0f / 116.0f) / 7.787f;
        }
        xyz[0] *= 0.95047f;
        xyz[2] *= 1.08883f;
        const float M[3][3] = {{3.24048134f, -0.96925495f, 0.05564664f},
                               {-1.53715152f, 1.87599f, -0.20404134f},
                               {-0.49853633f, 0.04155593f, 1.05731107f}};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = xyz[0] * M[0][k] + xyz[1] * M[1][k] + xyz[2] * M[2][k];
            v = v > 0.0031308f ? 1.055f * powf(v, 1.0f / 2.4f) - 0.055f : v * 12.92f;
            v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
            out[k * HW + i] = v;
        }
    }
}

extern "C" int dvc_lab2rgb(const float* lab, int32_t N, int32_t HW, float l_offset, float* rgb,
                           dvcStream stream) {
    DVC_REQUIRE(lab && rgb && N > 0 && HW > 0, "dvc_lab2rgb: bad argument");
    hipLaunchKernelGGL(lab2rgb_kernel, dim3(cdiv(HW, 1024), N), dim3(256), 0, (hipStream_t)stream, lab,
                       (long)HW, l_offset, rgb);
    DVC_CHECK_LAUNCH("dvc_lab2rgb");
    return 0;
}

// models/FrameColor.py:63-64: channels [L | warped a,b | similarity | last L,a,b]
// The previous frame arrives as its two parts — the luminance plane (channel 0 of the previous INPUT frame) and the
// previous ab prediction — so that the caller never has to materialise cat(IA_l, ab) (test.py:96) between frames.
__global__ __launch_bounds__(256) void pack_color_input_kernel(const float* __restrict__ IA_l, long ia_bs,
                                                               const float* __restrict__ warped,
                                                               const float* __restrict__ sim,
                                                               const float* __restrict__ last_l, long ll_bs,
                                                               const float* __restrict__ last_ab, long lab_bs, long HW,
                                                               float* __restrict__ out) {
    const int n = blockIdx.y;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        float* o = out + (long)n * 7 * HW + i;
        o[0] = IA_l[(long)n * ia_bs + i];
        o[HW] = warped[(long)n * 3 * HW + HW + i];
        o[2 * HW] = warped[(long)n * 3 * HW + 2 * HW + i];
        o[3 * HW] = sim[(long)n * HW + i];
        o[4 * HW] = last_l[(long)n * ll_bs + i];
        o[5 * HW] = last_ab[(long)n * lab_bs + i];
        o[6 * HW] = last_ab[(long)n * lab_bs + HW + i];
    }
}

extern "C" int dvc_pack_color_input(const float* IA_l, int64_t ia_batch_stride, const float* warped_lab, const float* sim,
                                    const float* last_l, int64_t last_l_batch_stride, const float* last_ab,
                                    int64_t last_ab_batch_stride, int32_t N, int32_t HW, float* out7, dvcStream stream) {
    DVC_REQUIRE(IA_l && warped_lab && sim && last_l && last_ab && out7 && N > 0 && HW > 0,
                "dvc_pack_color_input: bad argument");
    hipLaunchKernelGGL(pack_color_input_kernel, dim3(cdiv(HW, 1024), N), dim3(256), 0,
                       (hipStream_t)stream, IA_l, ia_batch_stride < 0 ? 0L : ia_batch_stride ? (long)ia_batch_stride : (long)HW,
                       warped_lab, sim, last_l,
                       last_l_batch_stride < 0 ? 0L : last_l_batch_stride ? (long)last_l_batch_stride : (long)HW, last_ab,
                       last_ab_batch_stride < 0 ? 0L : last_ab_batch_stride ? (long)last_ab_batch_stride : 2L * HW, (long)HW, out7);
    DVC_CHECK_LAUNCH("dvc_pack_color_input");
    return 0;
}
