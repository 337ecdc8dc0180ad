// Frame ingest, geometric half: CenterPad(image_size) of utils/util_distortion.py:217-258
// on the device — skimage.transform.resize(..., mode="reflect", preserve_range=True, clip=False,
// anti_aliasing=True) = Gaussian pre-filter (sigma = (factor - 1) / 2 per axis, radius int(4 sigma + 0.5), mirror
// boundary) + bilinear sampling at (o + 0.5) * n_in / n_out - 0.5 (mirror boundary), all in float64 like the
// reference's host code, then the centre crop and astype(uint8).  Bandwidth-trivial (one 8-bit frame in, one
// out); written for coalescing: rows of the interleaved H x W x 3 image are contiguous, so the vertical filter
// walks flat columns and the horizontal one strides by 3.
#include "common.h"

#define INGEST_MAX_RADIUS 40   // sigma <= 10, i.e. down-scaling factors up to 21

struct GaussTaps {
    int radius;
    double w[2 * INGEST_MAX_RADIUS + 1];
};

__device__ __forceinline__ int mirror_index(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * n - 2;
    int m = i % p;
    if (m < 0) m += p;
    return m >= n ? p - m : m;
}

// vertical pass (axis 0) of the uint8 image -> float64; radius 0 = plain conversion
__global__ __launch_bounds__(256) void ingest_gauss_v_kernel(const unsigned char* __restrict__ img, int H, int WC,
                                                             GaussTaps k, double* __restrict__ out) {
    const long n = (long)H * WC;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int y = (int)(i / WC), xc = (int)(i - (long)y * WC);
        double acc = 0.0;
        for (int j = -k.radius; j <= k.radius; ++j)
            acc += k.w[j + k.radius] * (double)img[(long)mirror_index(y + j, H) * WC + xc];
        out[i] = acc;
    }
}
// horizontal pass (axis 1), float64 -> float64
__global__ __launch_bounds__(256) void ingest_gauss_h_kernel(const double* __restrict__ in, int H, int W, GaussTaps k,
                                                             double* __restrict__ out) {
    const long n = (long)H * W * 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / (W * 3);
        const int xc = (int)(i - row * W * 3), x = xc / 3, c = xc - 3 * x;
        double acc = 0.0;
        for (int j = -k.radius; j <= k.radius; ++j)
            acc += k.w[j + k.radius] * in[row * W * 3 + (long)mirror_index(x + j, W) * 3 + c];
        out[i] = acc;
    }
}
// bilinear sampling of the filtered image [H0][W0][3] at the resized grid [nh][nw], cropped to the H x W window
// that starts at (y_start, x_start); astype(uint8)
__global__ __launch_bounds__(256) void ingest_zoom_crop_kernel(const double* __restrict__ f, int H0, int W0, int nh,
                                                               int nw, int y_start, int x_start, int H, int W,
                                                               unsigned char* __restrict__ out) {
    const double sy = (double)H0 / (double)nh, sx = (double)W0 / (double)nw;
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int oy = (int)(i / W), ox = (int)(i - (long)oy * W);
        const double cy = ((double)(oy + y_start) + 0.5) * sy - 0.5, cx = ((double)(ox + x_start) + 0.5) * sx - 0.5;
        const double fy = floor(cy), fx = floor(cx);
        const double ty = cy - fy, tx = cx - fx;
        const int y0 = mirror_index((int)fy, H0), y1 = mirror_index((int)fy + 1, H0);
        const int x0 = mirror_index((int)fx, W0), x1 = mirror_index((int)fx + 1, W0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // axis 0 first, then axis 1 (the order scipy.ndimage.zoom interpolates in)
            const double v0 = f[((long)y0 * W0 + x0) * 3 + c] * (1.0 - ty) + f[((long)y1 * W0 + x0) * 3 + c] * ty;
            const double v1 = f[((long)y0 * W0 + x1) * 3 + c] * (1.0 - ty) + f[((long)y1 * W0 + x1) * 3 + c] * ty;
            const double v = v0 * (1.0 - tx) + v1 * tx;
            out[i * 3 + c] = (unsigned char)(int)v;      // numpy astype(uint8) of a value in [0, 256)
        }
    }
}

static bool make_taps(double factor, GaussTaps* k) {
    const double sigma = factor > 1.0 ? (factor - 1.0) / 2.0 : 0.0;
    k->radius = 0;
    k->w[0] = 1.0;
    if (sigma <= 0.0) return true;
    const int r = (int)(4.0 * sigma + 0.5);
    if (r > INGEST_MAX_RADIUS) return false;
    double sum = 0.0;
    for (int j = -r; j <= r; ++j) {
        k->w[j + r] = exp(-0.5 * ((double)j / sigma) * ((double)j / sigma));
        sum += k->w[j + r];
    }
    for (int j = 0; j <= 2 * r; ++j) k->w[j] /= sum;
    k->radius = r;
    return true;
}

extern "C" size_t dvc_center_pad_workspace_bytes(int32_t H0, int32_t W0) {
    return sizeof(double) * 2 * (size_t)H0 * W0 * 3;
}

extern "C" int dvc_center_pad(const uint8_t* img, int32_t H0, int32_t W0, int32_t H, int32_t W, uint8_t* out,
                              void* workspace, size_t workspace_bytes, dvcStream stream) {
    DVC_REQUIRE(img && out && H0 > 0 && W0 > 0 && H > 0 && W > 0, "dvc_center_pad: bad argument");
    DVC_REQUIRE((long)H0 * W0 * 3 < (1L << 31), "dvc_center_pad: image too large");
    hipStream_t s = (hipStream_t)stream;
    // utils/util_distortion.py:235-256, with Python's float arithmetic (height / width etc. are doubles there)
    const double ratio = (double)H / (double)W, ratio_old = (double)H0 / (double)W0;
    int nh, nw, y_start = 0, x_start = 0;
    if (ratio_old == ratio) {
        if (H0 == H) {
            hipError_t e = hipMemcpyAsync(out, img, (size_t)H0 * W0 * 3, hipMemcpyDeviceToDevice, s);
            DVC_REQUIRE(e == hipSuccess, "dvc_center_pad: copy failed: %s", hipGetErrorString(e));
            return 0;
        }
        nh = (int)((double)H0 * H / H0);
        nw = (int)((double)W0 * H / H0);
        DVC_REQUIRE(nh == H && nw == W, "dvc_center_pad: resized size %dx%d != %dx%d (the reference would hand a "
                    "different size to CenterCrop; not supported)", nh, nw, H, W);
    } else if (ratio_old > ratio) {   // resize to the target width, crop the height
        nh = (int)((double)H0 * W / W0);
        nw = (int)((double)W0 * W / W0);
        y_start = (nh - H) / 2;
        DVC_REQUIRE(nw == W && nh >= H, "dvc_center_pad: resized size %dx%d does not cover %dx%d", nh, nw, H, W);
    } else {                          // resize to the target height, crop the width
        nh = (int)((double)H0 * H / H0);
        nw = (int)((double)W0 * H / H0);
        x_start = (nw - W) / 2;
        DVC_REQUIRE(nh == H && nw >= W, "dvc_center_pad: resized size %dx%d does not cover %dx%d", nh, nw, H, W);
    }
    DVC_REQUIRE(workspace && workspace_bytes >= dvc_center_pad_workspace_bytes(H0, W0),
                "dvc_center_pad: workspace too small");
    GaussTaps kv, kh;
    DVC_REQUIRE(make_taps((double)H0 / nh, &kv) && make_taps((double)W0 / nw, &kh),
                "dvc_center_pad: down-scaling factor too large (anti-aliasing radius > %d)", INGEST_MAX_RADIUS);
    double* t0 = reinterpret_cast<double*>(workspace);
    double* t1 = t0 + (size_t)H0 * W0 * 3;
    const long n = (long)H0 * W0 * 3;
    const unsigned grid = (unsigned)((n + 1023) / 1024);
    hipLaunchKernelGGL(ingest_gauss_v_kernel, dim3(grid), dim3(256), 0, s, img, H0, W0 * 3, kv, t0);
    const double* filtered = t0;
    if (kh.radius > 0) {
        hipLaunchKernelGGL(ingest_gauss_h_kernel, dim3(grid), dim3(256), 0, s, t0, H0, W0, kh, t1);
        filtered = t1;
    }
    hipLaunchKernelGGL(ingest_zoom_crop_kernel, dim3((unsigned)(((long)H * W + 255) / 256)), dim3(256), 0, s, filtered,
                       H0, W0, nh, nw, y_start, x_start, H, W, out);
    DVC_CHECK_LAUNCH("dvc_center_pad");
    return 0;
}
