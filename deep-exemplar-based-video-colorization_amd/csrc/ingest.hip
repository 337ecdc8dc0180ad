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

// bilinear sample from the four neighbours (axis 0 first, then axis 1: the order scipy.ndimage.zoom interpolates in); the
// contraction into fused multiply-adds is written out so that the fused kernel below and the three-pass path agree bit for bit
__device__ __forceinline__ double ingest_lerp2(double f00, double f10, double f01, double f11, double ty, double tx) {
    const double v0 = fma(f10, ty, f00 * (1.0 - ty));
    const double v1 = fma(f11, ty, f01 * (1.0 - ty));
    return fma(v1, tx, v0 * (1.0 - tx));
}

// vertical pass (axis 0) of the uint8 image -> float64; radius 0 = plain conversion
__global__ __launch_bounds__(256) void ingest_gauss_v_kernel(const unsigned char* __restrict__ img, int H, int WC,
                                                             GaussTaps k, double* __restrict__ out) {
    const long n = (long)H * WC;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int y = (int)(i / WC), xc = (int)(i - (long)y * WC);
        double acc = 0.0;
        for (int j = -k.radius; j <= k.radius; ++j)
            acc = fma(k.w[j + k.radius], (double)img[(long)mirror_index(y + j, H) * WC + xc], acc);
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
            acc = fma(k.w[j + k.radius], in[row * W * 3 + (long)mirror_index(x + j, W) * 3 + c], acc);
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
            const double v = ingest_lerp2(f[((long)y0 * W0 + x0) * 3 + c], f[((long)y1 * W0 + x0) * 3 + c], f[((long)y0 * W0 + x1) * 3 + c],
                                          f[((long)y1 * W0 + x1) * 3 + c], ty, tx);
            out[i * 3 + c] = (unsigned char)(int)v;      // numpy astype(uint8) of a value in [0, 256)
        }
    }
}

// ---- r06: the three passes as ONE kernel for the usual case (down-scaling by up to 3.25: Gaussian radius R <= 4, the same on both
// axes).  The three-pass path filters the whole source frame in float64 — 1080p: two passes over 6.2 M values, 50 MB written and
// read again twice, 47 + 50 us, to sample 1 M of them.  Here a workgroup owns an 8 x 32 tile of OUTPUT pixels:
//   1. it stages the 8-bit source window the tile needs in LDS (mirror boundaries resolved at staging time);
//   2. the vertical sums V[y][x] = sum_j w_j img[y + j][x] are needed only for the source rows y the tile's samples read — rows
//      floor(cy) and floor(cy) + 1 of its 8 output rows, 16 "row slots" — at every window column: computed once per workgroup,
//      float64, into LDS;
//   3. every thread takes the horizontal sums at its four sample points from those and interpolates.
// Every value is produced by the three-pass path's own sequence of operations (taps -R .. R in order, fused multiply-adds, then
// the sample): the bytes are the same (tests/test_ingest.py).  Needs 0 <= floor(c) and floor(c) + 1 <= n_in - 1 for every
// sample coordinate, which down-scaling (factor >= 1) guarantees.
#define INGEST_TH 8
#define INGEST_TW 32
struct GaussTapsSmall { double v[9], h[9]; };
template <int R>
__global__ __launch_bounds__(256) void ingest_fused_kernel(const unsigned char* __restrict__ img, int H0, int W0, int nh, int nw,
                                                           int y_start, int x_start, int H, int W, GaussTapsSmall k, int win_h,
                                                           int win_w, unsigned char* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ingest_lds[];
    const int wbytes = win_w * 3;
    double* Vs = reinterpret_cast<double*>(ingest_lds);                        // [2 * INGEST_TH row slots][win_w * 3]
    unsigned char* win = ingest_lds + sizeof(double) * 2 * INGEST_TH * wbytes;  // [win_h][win_w][3]
    const double sy = (double)H0 / (double)nh, sx = (double)W0 / (double)nw;
    const int oy0 = blockIdx.y * INGEST_TH, ox0 = blockIdx.x * INGEST_TW;
    // first source row / column the tile touches (the tile's first sample, minus the filter radius)
    const int wr0 = (int)floor(((double)(oy0 + y_start) + 0.5) * sy - 0.5) - R;
    const int wc0 = (int)floor(((double)(ox0 + x_start) + 0.5) * sx - 0.5) - R;
    // (rows / row slots go to the waves, bytes of a row to the lanes: the row's index arithmetic — the mirror, the float64 sample
    // coordinate — is wave-uniform, and a lane's source column offsets are the same for every row: computed once.  As one flat
    // loop over bytes with a division and two modulos per byte the staging alone took ~35 us.  The full mirror rule, not one
    // reflection: the window of an edge tile reaches up to a tile's width beyond a small image.)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NCB = 6;                                  // 64 * 6 >= the widest window's bytes (host: win_w * 3 <= 384)
    int coff[NCB];
#pragma unroll
    for (int t = 0; t < NCB; ++t) {
        const int cb = lane + 64 * t, c = cb / 3, ch = cb - 3 * c;
        coff[t] = cb < wbytes ? mirror_index(wc0 + c, W0) * 3 + ch : -1;
    }
    for (int r = wave; r < win_h; r += 4) {
        const unsigned char* src = img + (long)mirror_index(wr0 + r, H0) * W0 * 3;
#pragma unroll
        for (int t = 0; t < NCB; ++t)
            if (coff[t] >= 0) win[r * wbytes + lane + 64 * t] = src[coff[t]];
    }
    __syncthreads();
    // vertical sums: slot s = 2 * (output row of the tile) + (0: the row floor(cy), 1: the row below it)
    for (int slot = wave; slot < 2 * INGEST_TH; slot += 4) {
        const int fyr = (int)floor(((double)(oy0 + (slot >> 1) + y_start) + 0.5) * sy - 0.5);
        const unsigned char* prow = win + (fyr + (slot & 1) - R - wr0) * wbytes;      // window row of tap -R
        for (int cb = lane; cb < wbytes; cb += 64) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j <= 2 * R; ++j) a = fma(k.v[j], (double)prow[j * wbytes + cb], a);
            Vs[slot * wbytes + cb] = a;
        }
    }
    __syncthreads();
    const int ty_ = threadIdx.x / INGEST_TW, tx_ = threadIdx.x % INGEST_TW;
    const int oy = oy0 + ty_, ox = ox0 + tx_;
    if (oy >= H || ox >= W) return;
    const double cy = ((double)(oy + y_start) + 0.5) * sy - 0.5, cx = ((double)(ox + x_start) + 0.5) * sx - 0.5;
    const double fy = floor(cy), fx = floor(cx);
    const double ty = cy - fy, tx = cx - fx;
    const int lc = (int)fx - R - wc0;                                   // window column of the horizontal tap -R around x0
    const double* v0 = Vs + (2 * ty_) * wbytes + lc * 3;                // row slot of y0; the next slot is y1's
    const double* v1 = v0 + wbytes;
    unsigned char res[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        double f00 = 0.0, f01 = 0.0, f10 = 0.0, f11 = 0.0;      // f[y][x]: y0 / y1, x0 / x1
        double a0[2 * R + 2], a1[2 * R + 2];
#pragma unroll
        for (int c = 0; c < 2 * R + 2; ++c) {
            a0[c] = v0[c * 3 + ch];
            a1[c] = v1[c * 3 + ch];
        }
#pragma unroll
        for (int j = 0; j <= 2 * R; ++j) {
            f00 = fma(k.h[j], a0[j], f00);
            f01 = fma(k.h[j], a0[j + 1], f01);
            f10 = fma(k.h[j], a1[j], f10);
            f11 = fma(k.h[j], a1[j + 1], f11);
        }
        res[ch] = (unsigned char)(int)ingest_lerp2(f00, f10, f01, f11, ty, tx);
    }
    unsigned char* o = out + ((long)oy * W + ox) * 3;
    o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
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

// resized size and crop offsets of CenterPad (utils/util_distortion.py:235-256, with Python's float arithmetic: height / width
// etc. are doubles there); 0 = same size (plain copy), 1 = resize, -1 = a size the reference would hand on wrongly
static int center_pad_geometry(int H0, int W0, int H, int W, int* nh, int* nw, int* y_start, int* x_start) {
    const double ratio = (double)H / (double)W, ratio_old = (double)H0 / (double)W0;
    *y_start = *x_start = 0;
    if (ratio_old == ratio) {
        if (H0 == H) { *nh = H; *nw = W; return 0; }
        *nh = (int)((double)H0 * H / H0);
        *nw = (int)((double)W0 * H / H0);
        return (*nh == H && *nw == W) ? 1 : -1;
    } else if (ratio_old > ratio) {   // resize to the target width, crop the height
        *nh = (int)((double)H0 * W / W0);
        *nw = (int)((double)W0 * W / W0);
        *y_start = (*nh - H) / 2;
        return (*nw == W && *nh >= H) ? 1 : -1;
    }
    *nh = (int)((double)H0 * H / H0);   // resize to the target height, crop the width
    *nw = (int)((double)W0 * H / H0);
    *x_start = (*nw - W) / 2;
    return (*nh == H && *nw >= W) ? 1 : -1;
}

// the fused kernel's window: source rows / columns an INGEST_TH x INGEST_TW tile of samples can touch (+ the filter radius on
// both sides, + the second bilinear neighbour, + 1 for the tile origin's rounding)
static int ingest_win(int tile, double scale, int radius) { return (int)ceil(tile * scale) + 2 * radius + 3; }

// dynamic LDS of the fused kernel: the vertical sums of 16 row slots (float64) + the 8-bit window
static size_t ingest_fused_lds(int win_h, int win_w) { return sizeof(double) * 2 * INGEST_TH * win_w * 3 + (size_t)win_h * win_w * 3; }

static bool center_pad_fused_ok(int H0, int W0, int nh, int nw, const GaussTaps& kv, const GaussTaps& kh) {
    if (nh > H0 || nw > W0 || kv.radius != kh.radius || kv.radius > 4) return false;
    const int win_w = ingest_win(INGEST_TW, (double)W0 / nw, kv.radius);
    return win_w * 3 <= 384 && ingest_fused_lds(ingest_win(INGEST_TH, (double)H0 / nh, kv.radius), win_w) <= 64 * 1024;
}

extern "C" int dvc_center_pad_is_fused(int32_t H0, int32_t W0, int32_t H, int32_t W) {
    if (H0 <= 0 || W0 <= 0 || H <= 0 || W <= 0) return 0;
    int nh, nw, ys, xs;
    const int g = center_pad_geometry(H0, W0, H, W, &nh, &nw, &ys, &xs);
    if (g == 0) return 1;       // plain copy: no workspace either
    if (g < 0) return 0;
    GaussTaps kv, kh;
    if (!make_taps((double)H0 / nh, &kv) || !make_taps((double)W0 / nw, &kh)) return 0;
    return center_pad_fused_ok(H0, W0, nh, nw, kv, kh) ? 1 : 0;
}

extern "C" int dvc_center_pad(const uint8_t* img, int32_t H0, int32_t W0, int32_t H, int32_t W, uint8_t* out,
                              void* workspace, size_t workspace_bytes, dvcStream stream) {
    DVC_REQUIRE(img && out && H0 > 0 && W0 > 0 && H > 0 && W > 0, "dvc_center_pad: bad argument");
    DVC_REQUIRE((long)H0 * W0 * 3 < (1L << 31), "dvc_center_pad: image too large");
    hipStream_t s = (hipStream_t)stream;
    int nh, nw, y_start, x_start;
    const int geo = center_pad_geometry(H0, W0, H, W, &nh, &nw, &y_start, &x_start);
    if (geo == 0) {
        hipError_t e = hipMemcpyAsync(out, img, (size_t)H0 * W0 * 3, hipMemcpyDeviceToDevice, s);
        DVC_REQUIRE(e == hipSuccess, "dvc_center_pad: copy failed: %s", hipGetErrorString(e));
        return 0;
    }
    DVC_REQUIRE(geo > 0, "dvc_center_pad: resized size %dx%d does not give %dx%d (the reference would hand a different size to "
                "CenterCrop; not supported)", nh, nw, H, W);
    GaussTaps kv, kh;
    DVC_REQUIRE(make_taps((double)H0 / nh, &kv) && make_taps((double)W0 / nw, &kh),
                "dvc_center_pad: down-scaling factor too large (anti-aliasing radius > %d)", INGEST_MAX_RADIUS);
    // r06: no workspace = the fused kernel (dvc_center_pad_is_fused says whether it applies); with one, the three passes
    if (!workspace && center_pad_fused_ok(H0, W0, nh, nw, kv, kh)) {
        GaussTapsSmall k;
        for (int j = 0; j < 9; ++j) {
            k.v[j] = j <= 2 * kv.radius ? kv.w[j] : 0.0;
            k.h[j] = j <= 2 * kh.radius ? kh.w[j] : 0.0;
        }
        const int win_h = ingest_win(INGEST_TH, (double)H0 / nh, kv.radius), win_w = ingest_win(INGEST_TW, (double)W0 / nw, kv.radius);
        const dim3 grid(cdiv(W, INGEST_TW), cdiv(H, INGEST_TH));
        const size_t lds = ingest_fused_lds(win_h, win_w);
#define INGEST_FUSED(R_) case R_: hipLaunchKernelGGL((ingest_fused_kernel<R_>), grid, dim3(256), lds, s, img, H0, W0, nh, nw, y_start, x_start, H, W, k, win_h, win_w, out); break;
        switch (kv.radius) {
            INGEST_FUSED(0) INGEST_FUSED(1) INGEST_FUSED(2) INGEST_FUSED(3) INGEST_FUSED(4)
            default: break;
        }
#undef INGEST_FUSED
        DVC_CHECK_LAUNCH("dvc_center_pad(fused)");
        return 0;
    }
    DVC_REQUIRE(workspace && workspace_bytes >= dvc_center_pad_workspace_bytes(H0, W0),
                "dvc_center_pad: workspace too small");
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
