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
                                                          float* __restrict__ wh_t, int line_major) {
    // thread-per-line solver (line_major == 0): every line's elements are M apart
    //   wv[y][x]   = w((y,x),(y+1,x))   [H][W]   (last row unused)
    //   wh_t[x][y] = w((y,x),(y,x+1))   [W][H]   (last row unused) — the horizontal weights, transposed
    // scan solver (line_major == 1, r05): every line contiguous — the first array [H][W] holds the HORIZONTAL weights (row lines),
    // the second [W][H] the VERTICAL ones transposed (column lines)
    // r06: a workgroup = one 32 x 32 tile; the transposed array goes through LDS so that both are written in rows (as a
    // grid-stride loop with wh_t[x * H + y] written element by element the launch took 9.7 us at 432x768, most of it that store).
    __shared__ float t[32][33];
    g += (long)blockIdx.z * H * W;
    wv += (long)blockIdx.z * H * W;
    wh_t += (long)blockIdx.z * H * W;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x = bx + tx;
    for (int r = ty; r < 32; r += 8) {
        const int y = by + r;
        if (y < H && x < W) {
            const int i = y * W + x;
            const int c = g[i];
            const int dn = y + 1 < H ? abs(c - (int)g[i + W]) : 0;
            const int rt = x + 1 < W ? abs(c - (int)g[i + 1]) : 0;
            const float wd = expf(-(float)dn * inv_sigma), wr = expf(-(float)rt * inv_sigma);
            wv[i] = line_major ? wr : wd;
            t[r][tx] = line_major ? wd : wr;
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (bx + r < W && by + tx < H) wh_t[(long)(bx + r) * H + by + tx] = t[tx][r];
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
// r05: the loads of block k+1 are issued BEFORE the dependent chain of block k runs (two register sets, the block loop unrolled
// twice).  With one thread per line there are only planes x lines threads on the whole chip (24-48 waves at 432x768), so nothing
// else hides a load's latency: the r04 kernels fetched a block, waited ~1.5 us for it, ran 16 dependent steps, fetched the next
// one — 48 such round trips per 768-long sweep, i.e. the sweeps ran at memory LATENCY (profiles/r04_tail_probe.txt: 635 us of WLS
// per frame against ~30 us of dependent arithmetic).  Same operations in the same order on every element: bit-identical results.
#define FGS_U 16
// blockIdx.z = iteration t (r05: the coefficients of ALL iterations come from one launch per direction pair — they depend on the
// guide and lambda_t only, not on the image being filtered, so nothing orders them behind the solves; r04 launched 2 x T of these
// latency chains one after the other, ~25 % of the filter's time).  lambda_t arrives in `lam[t]` (the host's float recurrence).
struct FgsLambdas { float v[8]; };
// One direction's problem: weights w [guide][L][M] (M lines of length L, thread-per-line layout), outputs at cp / inv / ap.
struct FgsCoeffDir {
    const float* w;
    float *cp, *inv, *ap;
    int L, M;
};
// blockIdx.z = direction * num_iter + iteration: BOTH directions' chains of every iteration in one launch (the row system's
// 768-long chains and the column system's 432-long ones run side by side: ~55 us instead of 59 + 39, rocprofv3,
// profiles/r05_tail_kernel_stats*.csv).  A direction with cp == NULL is skipped.
__global__ __launch_bounds__(64) void fgs_coeff_kernel(FgsCoeffDir d0, FgsCoeffDir d1, int num_iter, FgsLambdas lam, long iter_stride) {
    const int dir = blockIdx.z / num_iter, it = blockIdx.z - dir * num_iter;
    const FgsCoeffDir& d = dir ? d1 : d0;
    const int L = d.L, M = d.M;
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= M || !d.cp) return;
    const float lambda = lam.v[it];
    const long off = (long)blockIdx.y * L * M + m;       // blockIdx.y = guide
    const float* wp = d.w + off;
    float* cpp = d.cp + off + (long)it * iter_stride;
    float* ivp = d.inv + off + (long)it * iter_stride;
    // ap_l = -(a_l inv_l) = (lambda w(l-1,l)) inv_l, the forward sweep's multiplier (ap_0 = 0), for the scan solver; optional
    float* app = d.ap ? d.ap + off + (long)it * iter_stride : nullptr;
    if (app) app[0] = 0.f;
    float a = 0.f;
    float c = L > 1 ? -lambda * wp[0] : 0.f;
    float iv = 1.f / (1.f - a - c);
    float cprev = c * iv;
    cpp[0] = cprev;
    ivp[0] = iv;
    // full blocks [1 + b U, 1 + (b + 1) U), b < nblk, never contain the last element: no conditions inside
    const int nblk = L > 1 ? (L - 2) / FGS_U : 0;
    float wa[FGS_U], wb[FGS_U];
    auto load = [&](int l0, float (&wk)[FGS_U]) {
#pragma unroll
        for (int k = 0; k < FGS_U; ++k) wk[k] = wp[(long)(l0 + k) * M];
    };
    auto chain = [&](int l0, const float (&wk)[FGS_U]) {
#pragma unroll
        for (int k = 0; k < FGS_U; ++k) {
            a = c;                                       // a_l = -lambda w(l-1,l) = c_{l-1}
            c = -lambda * wk[k];
            iv = 1.f / ((1.f - a - c) - a * cprev);
            cprev = c * iv;
            cpp[(long)(l0 + k) * M] = cprev;
            ivp[(long)(l0 + k) * M] = iv;
            if (app) app[(long)(l0 + k) * M] = -a * iv;
        }
    };
    if (nblk > 0) load(1, wa);
    int b = 0;
    for (; b + 1 < nblk; b += 2) {
        load(1 + (b + 1) * FGS_U, wb);
        chain(1 + b * FGS_U, wa);
        if (b + 2 < nblk) load(1 + (b + 2) * FGS_U, wa);
        chain(1 + (b + 1) * FGS_U, wb);
    }
    if (b < nblk) chain(1 + b * FGS_U, wa);
    for (int l = 1 + nblk * FGS_U; l < L; ++l) {
        a = c;
        c = l + 1 < L ? -lambda * wp[(long)l * M] : 0.f;
        iv = 1.f / ((1.f - a - c) - a * cprev);
        cprev = c * iv;
        cpp[(long)l * M] = cprev;
        ivp[(long)l * M] = iv;
        if (app) app[(long)l * M] = -a * iv;
    }
}

// ---- r05: the sweeps as SCANS, one wave per line.  d'_l = ap_l d'_{l-1} + f_l inv_l and u_l = -c'_l u_{l+1} + d'_l are first-order
// linear recurrences, i.e. compositions of affine maps x -> P x + S, which is associative: lane j composes the maps of its E
// consecutive elements (E = ceil(L / 64) dependent steps), a 6-step shuffle scan composes the lanes' maps, and a second pass of E
// steps produces the values — 2 E + 6 dependent steps per sweep instead of L, and planes x lines WAVES instead of threads
// (432x768: 864 / 1536 waves instead of 14 / 24).  The thread-per-line kernel below ran at one wave per SIMD with nothing to
// overlap its 13 instructions per element: ~20 us per 768-long sweep, 12 sweeps per frame.  |ap_l|, |c'_l| < 1 (diagonally
// dominant systems), so the composed maps are contractions and the result differs from the sequential sweep by a few ulp
// (tests/test_tail.py: same tolerance against the oracle as before, and against the fp64 banded solve).
// Layout: LINE-major ([plane][line][L]: a line is contiguous, lane j of the wave loads element k * 64 + j, coalesced) with the
// blocked re-distribution (lane j owns elements j E .. j E + E - 1) through LDS at an odd pitch (conflict-free).
template <int E>
__global__ __launch_bounds__(256) void fgs_solve_scan_kernel(const float* fin, float* f, const float* __restrict__ ap,
                                                            const float* __restrict__ inv, const float* __restrict__ cp, int L,
                                                            int nlines, int planes_per_guide) {
    // (fin: where the lines are read — f itself, or for the first sweep of a filter the caller's source image: r06, the copy
    // dst <- src that used to precede the sweeps is gone)
    constexpr int EP = E | 1;
    __shared__ float sA[4][64 * EP], sG[4][64 * EP], sC[4][64 * EP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int line_raw = blockIdx.x * 4 + wv;
    const bool active = line_raw < nlines;
    const int line = active ? line_raw : nlines - 1;           // (every wave reaches the barriers; inactive ones store nothing)
    const int plane = blockIdx.y;
    const long goff = ((long)(plane / planes_per_guide) * nlines + line) * L;
    float* fp = f + ((long)plane * nlines + line) * L;
    const float* fip = fin + ((long)plane * nlines + line) * L;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int p = k * 64 + lane;
        float a = 0.f, g = 0.f, c = 0.f;
        if (p < L) {
            a = ap[goff + p];
            g = fip[p] * inv[goff + p];
            c = cp[goff + p];
        }
        const int q = (p / E) * EP + (p % E);
        sA[wv][q] = a;
        sG[wv][q] = g;
        sC[wv][q] = c;
    }
    __syncthreads();
    float a[E], g[E], c[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        a[k] = sA[wv][lane * EP + k];
        g[k] = sG[wv][lane * EP + k];
        c[k] = -sC[wv][lane * EP + k];
    }
    // forward: the lane's composed map, inclusive scan over the lanes below, then the values
    float P = 1.f, S = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        S = fmaf(a[k], S, g[k]);
        P *= a[k];
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float Pp = __shfl_up(P, off), Sp = __shfl_up(S, off);
        if (lane >= off) {
            S = fmaf(P, Sp, S);
            P *= Pp;
        }
    }
    float d = __shfl_up(S, 1);
    if (lane == 0) d = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        d = fmaf(a[k], d, g[k]);
        g[k] = d;                                   // g now holds d'
    }
    // backward: u_l = c_l u_{l+1} + d'_l (c = -c'), scan over the lanes above
    P = 1.f, S = 0.f;
#pragma unroll
    for (int k = E - 1; k >= 0; --k) {
        S = fmaf(c[k], S, g[k]);
        P *= c[k];
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float Pp = __shfl_down(P, off), Sp = __shfl_down(S, off);
        if (lane + off < 64) {
            S = fmaf(P, Sp, S);
            P *= Pp;
        }
    }
    float u = __shfl_down(S, 1);
    if (lane == 63) u = 0.f;
#pragma unroll
    for (int k = E - 1; k >= 0; --k) {
        u = fmaf(c[k], u, g[k]);
        sG[wv][lane * EP + k] = u;
    }
    __syncthreads();
    if (!active) return;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int p = k * 64 + lane;
        if (p < L) fp[p] = sG[wv][(p / E) * EP + (p % E)];
    }
}

// ---- r06: the COLUMN solves without the two transpositions around them (r05: transpose -> row-form scan -> transpose back, six
// transpositions per frame = 21 % of the tail's kernel time).  Same scan, same arithmetic in the same order (bit-identical to the
// transposed form); what changes is how a line reaches its wave.  A workgroup = 16 waves = 16 ADJACENT columns: the 16 x L tile of
// the row-major image is read cooperatively (a row of the tile = 64 contiguous bytes) into LDS, column-major in the blocked
// order the scan wants (lane j's E consecutive elements contiguous, odd pitch), each wave scans its column — the
// coefficients of the column system are line-major already (fgs_coeff_window_kernel writes [guide][W][H]), so a lane reads its E
// consecutive values of ap / inv / c' straight from memory — and the tile goes back the way it came.
#define FGS_COLS 16
template <int E>
__global__ __launch_bounds__(64 * FGS_COLS) void fgs_solve_scan_cols_kernel(float* __restrict__ f, const float* __restrict__ ap,
                                                                            const float* __restrict__ inv, const float* __restrict__ cp,
                                                                            int L, int ncols, int planes_per_guide) {
    constexpr int EP = E | 1;
    constexpr int PITCH = 64 * EP + 1;                      // (+1: the 16 columns of one tile row land in 16 different banks)
    __shared__ float sG[FGS_COLS * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0 = blockIdx.x * FGS_COLS;
    const int plane = blockIdx.y;
    float* fpl = f + (long)plane * L * ncols;               // [L][ncols]: element l of column x at l * ncols + x
    // tile in: thread = (row r of a 64-row pass, column cx)
    const int cx = tid & (FGS_COLS - 1), r0 = tid / FGS_COLS;
    const bool colok = x0 + cx < ncols;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int l = k * 64 + r0;
        if (l < L && colok) sG[cx * PITCH + (l / E) * EP + (l % E)] = fpl[(long)l * ncols + x0 + cx];
    }
    __syncthreads();
    const int col = x0 + wv;
    const bool active = col < ncols;
    if (active) {
        const long goff = ((long)(plane / planes_per_guide) * ncols + col) * L + lane * E;
        float a[E], g[E], c[E];
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const bool ok = lane * E + k < L;
            a[k] = ok ? ap[goff + k] : 0.f;
            g[k] = ok ? sG[wv * PITCH + lane * EP + k] * inv[goff + k] : 0.f;
            c[k] = ok ? -cp[goff + k] : 0.f;
        }
        // (from here on: fgs_solve_scan_kernel's arithmetic, statement for statement)
        float P = 1.f, S = 0.f;
#pragma unroll
        for (int k = 0; k < E; ++k) {
            S = fmaf(a[k], S, g[k]);
            P *= a[k];
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float Pp = __shfl_up(P, off), Sp = __shfl_up(S, off);
            if (lane >= off) {
                S = fmaf(P, Sp, S);
                P *= Pp;
            }
        }
        float d = __shfl_up(S, 1);
        if (lane == 0) d = 0.f;
#pragma unroll
        for (int k = 0; k < E; ++k) {
            d = fmaf(a[k], d, g[k]);
            g[k] = d;
        }
        P = 1.f, S = 0.f;
#pragma unroll
        for (int k = E - 1; k >= 0; --k) {
            S = fmaf(c[k], S, g[k]);
            P *= c[k];
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float Pp = __shfl_down(P, off), Sp = __shfl_down(S, off);
            if (lane + off < 64) {
                S = fmaf(P, Sp, S);
                P *= Pp;
            }
        }
        float u = __shfl_down(S, 1);
        if (lane == 63) u = 0.f;
#pragma unroll
        for (int k = E - 1; k >= 0; --k) {
            u = fmaf(c[k], u, g[k]);
            sG[wv * PITCH + lane * EP + k] = u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int l = k * 64 + r0;
        if (l < L && colok) fpl[(long)l * ncols + x0 + cx] = sG[cx * PITCH + (l / E) * EP + (l % E)];
    }
}

static void fgs_solve_scan_cols(hipStream_t s, float* f, const float* ap, const float* inv, const float* cp, int L, int ncols, int planes,
                                int planes_per_guide) {
    const dim3 grid(cdiv(ncols, FGS_COLS), planes);
#define FGS_SCANC(E_) case E_: hipLaunchKernelGGL((fgs_solve_scan_cols_kernel<E_>), grid, dim3(64 * FGS_COLS), 0, s, f, ap, inv, cp, L, ncols, planes_per_guide); break;
    switch (cdiv(L, 64)) {
        FGS_SCANC(1) FGS_SCANC(2) FGS_SCANC(3) FGS_SCANC(4) FGS_SCANC(5) FGS_SCANC(6) FGS_SCANC(7) FGS_SCANC(8)
        FGS_SCANC(9) FGS_SCANC(10) FGS_SCANC(11) FGS_SCANC(12) FGS_SCANC(13) FGS_SCANC(14) FGS_SCANC(15) FGS_SCANC(16)
        default: break;
    }
#undef FGS_SCANC
}

// ---- r05: the elimination coefficients without the L-long chain.  c'_l = c_l / (b_l - a_l c'_{l-1}) FORGETS its start: the
// map is a contraction whose rate is largest for a flat guide (all weights 1), rho(lambda) = x*^2 with x* the fixed point
// -((1 + 2 lambda) - sqrt(1 + 4 lambda)) / (2 lambda) — 0.865 for the first iteration's lambda_1 = 190 (lambda = 500), 0.75 and
// 0.56 for the next two.  A wave per line: lane j owns E consecutive elements and first runs the recurrence from 0 over the
// `warm` elements in front of them (warm = ln(1e-8) / ln(rho): 127, 64, 32 steps: what is left of the arbitrary start is below
// 1e-8 of c', under the rounding of the fp32 recurrence itself), then over its own with the reference's expressions, writing
// c', 1/den and ap = -a / den LINE-major: chains of warm + E steps instead of L (139 instead of 768), nlines x 64 lanes instead
// of nlines threads, and no transposition of the coefficient sets.  (The Moebius-map scan — compose the maps as 2x2 matrices,
// scan across the lanes — was built first and is too ill-conditioned: the composed maps are nearly constant, the matrix products
// cancel, c' came out 1.5e-5 off in fp32 (3e-9 in fp64) and the filter 2e-2: profiles/EXPERIMENTS.md.)  lambda so large that
// warm exceeds FGS_WARM_MAX keeps the thread-per-line chain kernel.
#define FGS_WARM_MAX 256
struct FgsWarm { int v[8]; };
template <int E>
__global__ __launch_bounds__(256) void fgs_coeff_window_kernel(const float* __restrict__ w, int L, int nlines, FgsLambdas lam,
                                                              FgsWarm warm, long iter_stride, float* __restrict__ cp,
                                                              float* __restrict__ inv, float* __restrict__ ap) {
    constexpr int EP = E | 1;
    __shared__ float sw[4][64 * E + 1];
    __shared__ float s0[4][64 * EP], s1[4][64 * EP], s2[4][64 * EP];
    // (68.6 KB at E = 16: more than the 64 KB a workgroup gets on gfx90a / gfx942 — this library is written for gfx950's 160 KB)
    static_assert(sizeof(float) * (4 * (64 * E + 1) + 3 * 4 * 64 * EP) <= 160 * 1024, "static LDS of fgs_coeff_window_kernel");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int line_raw = blockIdx.x * 4 + wv;
    const bool active = line_raw < nlines;
    const int line = active ? line_raw : nlines - 1;
    const float lambda = lam.v[blockIdx.z];
    const int W = warm.v[blockIdx.z];
    const long loff = ((long)blockIdx.y * nlines + line) * L;       // blockIdx.y = guide
    const float* wp = w + loff;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int p = k * 64 + lane;
        sw[wv][p] = p < L ? wp[p] : 0.f;
    }
    __syncthreads();
    const float* wl = sw[wv];
    const int l0 = lane * E;
    // a_l = -lambda w(l-1,l) (0 for l = 0), c_l = -lambda w(l,l+1) (0 for the last element); before the line: x stays 0
    float x = 0.f;
    int l = l0 - W;
    float wprev = (l > 0 && l - 1 < L) ? wl[l - 1] : 0.f;
    // warm-up: W is wave-uniform and a multiple of 8.  Blocks of 8 steps: the eight LDS reads first (clamped indices, no
    // branches: as a rolled loop with a conditional read every step waited ~150 cycles for its own ds_read), then the eight
    // dependent steps.  (hardware reciprocal, 1 ulp: what the warm-up computes is forgotten at the same rate as its start
    // value — the line's own elements below use the reference's true division)
    for (int t = 0; t < W; t += 8) {
        float wk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wk[u] = wl[min(max(l + u, 0), 64 * E - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int li = l + u;
            const float wc = li >= 0 ? wk[u] : 0.f;                   // (entries beyond the line are zero in LDS)
            const float a = (li > 0 && li < L) ? -lambda * wprev : 0.f;
            const float c = (li >= 0 && li + 1 < L) ? -lambda * wc : 0.f;
            x = c * __builtin_amdgcn_rcpf((1.f - a - c) - a * x);
            wprev = wc;
        }
        l += 8;
    }
#pragma unroll
    for (int k = 0; k < E; ++k, ++l) {
        const float wc = l < L ? wl[l] : 0.f;
        const float a = (l > 0 && l < L) ? -lambda * wprev : 0.f;
        const float c = (l + 1 < L) ? -lambda * wc : 0.f;
        const float iv = 1.f / ((1.f - a - c) - a * x);
        x = c * iv;
        s0[wv][lane * EP + k] = x;
        s1[wv][lane * EP + k] = iv;
        s2[wv][lane * EP + k] = -a * iv;
        wprev = wc;
    }
    __syncthreads();
    if (!active) return;
    const long ooff = loff + (long)blockIdx.z * iter_stride;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int pi = k * 64 + lane;
        if (pi < L) {
            const int qi = (pi / E) * EP + (pi % E);
            cp[ooff + pi] = s0[wv][qi];
            inv[ooff + pi] = s1[wv][qi];
            ap[ooff + pi] = s2[wv][qi];
        }
    }
}

static void fgs_coeff_window(hipStream_t s, const float* w, int L, int nlines, int n_guides, int num_iter, const FgsLambdas& lam,
                             const FgsWarm& warm, long iter_stride, float* cp, float* inv, float* ap) {
    const dim3 grid(cdiv(nlines, 4), n_guides, num_iter);
#define FGS_CWIN(E_) case E_: hipLaunchKernelGGL((fgs_coeff_window_kernel<E_>), grid, dim3(256), 0, s, w, L, nlines, lam, warm, iter_stride, cp, inv, ap); break;
    switch (cdiv(L, 64)) {
        FGS_CWIN(1) FGS_CWIN(2) FGS_CWIN(3) FGS_CWIN(4) FGS_CWIN(5) FGS_CWIN(6) FGS_CWIN(7) FGS_CWIN(8)
        FGS_CWIN(9) FGS_CWIN(10) FGS_CWIN(11) FGS_CWIN(12) FGS_CWIN(13) FGS_CWIN(14) FGS_CWIN(15) FGS_CWIN(16)
        default: break;
    }
#undef FGS_CWIN
}

static void fgs_solve_scan(hipStream_t s, const float* fin, float* f, const float* ap, const float* inv, const float* cp, int L, int nlines,
                           int planes, int planes_per_guide) {
    const dim3 grid(cdiv(nlines, 4), planes);
#define FGS_SCAN(E_) case E_: hipLaunchKernelGGL((fgs_solve_scan_kernel<E_>), grid, dim3(256), 0, s, fin, f, ap, inv, cp, L, nlines, planes_per_guide); break;
    switch (cdiv(L, 64)) {
        FGS_SCAN(1) FGS_SCAN(2) FGS_SCAN(3) FGS_SCAN(4) FGS_SCAN(5) FGS_SCAN(6) FGS_SCAN(7) FGS_SCAN(8)
        FGS_SCAN(9) FGS_SCAN(10) FGS_SCAN(11) FGS_SCAN(12) FGS_SCAN(13) FGS_SCAN(14) FGS_SCAN(15) FGS_SCAN(16)
        default: break;
    }
#undef FGS_SCAN
}
#define FGS_SCAN_MAX_L 1024

// (lines longer than FGS_SCAN_MAX_L: one thread per line, as up to r04)
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
    // ---- forward sweep: full blocks [1 + b U, 1 + (b + 1) U)
    const int nblk = (L - 1) / FGS_U;
    {
        float iva[FGS_U], wa[FGS_U], fa[FGS_U], ivb[FGS_U], wb[FGS_U], fb[FGS_U];
        auto load = [&](int l0, float (&ivk)[FGS_U], float (&wk)[FGS_U], float (&fk)[FGS_U]) {
#pragma unroll
            for (int k = 0; k < FGS_U; ++k) {
                ivk[k] = ivp[(long)(l0 + k) * M];
                wk[k] = wp[(long)(l0 + k - 1) * M];
                fk[k] = fp[(long)(l0 + k) * M];
            }
        };
        auto chain = [&](int l0, const float (&ivk)[FGS_U], const float (&wk)[FGS_U], const float (&fk)[FGS_U]) {
            float ak[FGS_U], gk[FGS_U];
#pragma unroll
            for (int k = 0; k < FGS_U; ++k) {
                ak[k] = lambda * wk[k] * ivk[k];                            // -(a_l inv_l), a_l = -lambda w(l-1,l)
                gk[k] = fk[k] * ivk[k];
            }
#pragma unroll
            for (int k = 0; k < FGS_U; ++k) {
                dprev = fmaf(ak[k], dprev, gk[k]);
                dpp[(long)(l0 + k) * M] = dprev;
            }
        };
        if (nblk > 0) load(1, iva, wa, fa);
        int b = 0;
        for (; b + 1 < nblk; b += 2) {
            load(1 + (b + 1) * FGS_U, ivb, wb, fb);
            chain(1 + b * FGS_U, iva, wa, fa);
            if (b + 2 < nblk) load(1 + (b + 2) * FGS_U, iva, wa, fa);
            chain(1 + (b + 1) * FGS_U, ivb, wb, fb);
        }
        if (b < nblk) chain(1 + b * FGS_U, iva, wa, fa);
    }
    for (int l = 1 + nblk * FGS_U; l < L; ++l) {
        const float iv = ivp[(long)l * M];
        dprev = fmaf(lambda * wp[(long)(l - 1) * M] * iv, dprev, fp[(long)l * M] * iv);
        dpp[(long)l * M] = dprev;
    }
    // ---- backward sweep: full blocks (L - 2 - b U) downwards
    float u = dprev;
    fp[(long)(L - 1) * M] = u;
    const int nbb = (L - 1) / FGS_U;
    {
        float ca[FGS_U], da[FGS_U], cb[FGS_U], db[FGS_U];
        auto load = [&](int l1, float (&ck)[FGS_U], float (&dk)[FGS_U]) {
#pragma unroll
            for (int k = 0; k < FGS_U; ++k) {
                ck[k] = cpp[(long)(l1 - k) * M];
                dk[k] = dpp[(long)(l1 - k) * M];
            }
        };
        auto chain = [&](int l1, const float (&ck)[FGS_U], const float (&dk)[FGS_U]) {
#pragma unroll
            for (int k = 0; k < FGS_U; ++k) {
                u = fmaf(-ck[k], u, dk[k]);
                fp[(long)(l1 - k) * M] = u;
            }
        };
        if (nbb > 0) load(L - 2, ca, da);
        int b = 0;
        for (; b + 1 < nbb; b += 2) {
            load(L - 2 - (b + 1) * FGS_U, cb, db);
            chain(L - 2 - b * FGS_U, ca, da);
            if (b + 2 < nbb) load(L - 2 - (b + 2) * FGS_U, ca, da);
            chain(L - 2 - (b + 1) * FGS_U, cb, db);
        }
        if (b < nbb) chain(L - 2 - b * FGS_U, ca, da);
    }
    for (int l = L - 2 - nbb * FGS_U; l >= 0; --l) {
        u = fmaf(-cpp[(long)l * M], u, dpp[(long)l * M]);
        fp[(long)l * M] = u;
    }
}

#define FGS_MAX_ITER 8
extern "C" size_t dvc_fgs_workspace_bytes(int32_t H, int32_t W, int32_t n_guides, int32_t planes_per_guide, int32_t num_iter) {
    // per guide: wv, wh_t; (c', 1/m, ap) of every iteration and both directions (line-major for the scan solver, or in the
    // thread-per-line kernels' [L][M] layout: ONE copy — the rare case that needs both, a lambda too large for the windowed
    // coefficients on an image the scan solver takes, uses a second copy when the caller's workspace has room for it and the
    // thread-per-line solver otherwise); per plane: the transposed image (+ d' of the thread-per-line fall-back)
    if (num_iter < 1) num_iter = 1;
    return sizeof(float) * ((size_t)(2 + 6 * num_iter) * n_guides * H * W + (size_t)2 * n_guides * planes_per_guide * H * W);
}

extern "C" int dvc_fgs_filter(const uint8_t* guide, const float* src, int32_t n_guides, int32_t planes_per_guide,
                              int32_t H, int32_t W, float lambda, float sigma_color, int32_t num_iter,
                              float lambda_attenuation, float* dst, void* workspace, size_t workspace_bytes,
                              dvcStream stream) {
    DVC_REQUIRE(guide && src && dst && workspace && n_guides > 0 && planes_per_guide > 0 && H > 0 && W > 0,
                "dvc_fgs_filter: bad argument");
    const int planes = n_guides * planes_per_guide;
    DVC_REQUIRE(num_iter >= 1 && num_iter <= FGS_MAX_ITER && sigma_color > 0.f && lambda >= 0.f, "dvc_fgs_filter: bad parameters");
    DVC_REQUIRE(workspace_bytes >= dvc_fgs_workspace_bytes(H, W, n_guides, planes_per_guide, num_iter),
                "dvc_fgs_filter: workspace too small");
    DVC_REQUIRE((long)H * W < (1L << 30), "dvc_fgs_filter: image too large");
    hipStream_t s = (hipStream_t)stream;
    const size_t HW = (size_t)H * W, GHW = (size_t)n_guides * HW;
    const size_t IT = (size_t)num_iter * GHW;
    float* wv = reinterpret_cast<float*>(workspace);
    float* wh_t = wv + GHW;
    float* row_lm = wh_t + GHW;     // (c', 1/m, ap) x iterations, line-major: row system [guide][H][W]
    float* col_lm = row_lm + 3 * IT;  //                                         column system [guide][W][H]
    float* tr = col_lm + 3 * IT;    // transposed image [plane][W][H]
    float* dp = tr + planes * HW;   // (fall-back only)
    // the thread-per-line coefficient kernel's [L][M] layout (row system, column system): behind everything else when the
    // workspace has room for a second copy, else IN PLACE of the line-major sets (then the thread-per-line solver runs too)
    const bool room2 = workspace_bytes >= dvc_fgs_workspace_bytes(H, W, n_guides, planes_per_guide, num_iter) + sizeof(float) * 6 * IT;
    float* tmp = room2 ? dp + planes * HW : row_lm;
    float* tmp2 = tmp + 3 * IT;
    bool scan = H <= FGS_SCAN_MAX_L && W <= FGS_SCAN_MAX_L;
    double lam = 1.5 * (double)lambda * pow(4.0, num_iter - 1) / (pow(4.0, num_iter) - 1.0);
    FgsLambdas lams;
    lams.v[0] = (float)lam;
    for (int it = 1; it < FGS_MAX_ITER; ++it) lams.v[it] = lams.v[it - 1] * lambda_attenuation;      // (the float recurrence of r01-r04)
    // warm-up lengths of the windowed coefficient kernel: rho^warm <= 1e-8 for the slowest-forgetting (flat) guide
    FgsWarm warm;
    int warm_max = 0;
    for (int it = 0; it < FGS_MAX_ITER; ++it) {
        const double l = (double)lams.v[it];
        int wlen = 0;
        if (l > 0.0) {
            const double xs = ((1.0 + 2.0 * l) - sqrt(1.0 + 4.0 * l)) / (2.0 * l), rho = xs * xs;
            // (in double, clamped BEFORE the cast: rho -> 1 for a huge lambda, the quotient overflows an int or is not finite)
            const double steps = rho < 1e-8 ? 1.0 : ceil(log(1e-8) / log(rho)) + 1.0;
            wlen = (steps >= 1.0 && steps <= (double)FGS_WARM_MAX) ? (int)steps : FGS_WARM_MAX + 8;
            wlen = (wlen + 7) / 8 * 8;          // (the kernel's warm-up runs in blocks of 8 steps)
        }
        warm.v[it] = wlen;
        if (it < num_iter && wlen > warm_max) warm_max = wlen;
    }
    const bool window = scan && warm_max <= FGS_WARM_MAX;
    if (!window && !room2) scan = false;        // (one coefficient copy only: the thread-per-line solver reads it where it is)
    // (windowed coefficients: wv = the horizontal weights [H][W], wh_t = the vertical ones transposed [W][H]: every line contiguous)
    hipLaunchKernelGGL(fgs_weights_kernel, dim3(cdiv(W, 32), cdiv(H, 32), n_guides), dim3(256), 0, s, guide, H, W,
                       1.0f / sigma_color, wv, wh_t, window ? 1 : 0);
    DVC_CHECK_LAUNCH("dvc_fgs_filter(weights)");
    if (dst != src && !scan) {     // (the scan solver's first row sweep reads src itself)
        hipError_t e = hipMemcpyAsync(dst, src, sizeof(float) * planes * HW, hipMemcpyDeviceToDevice, s);
        DVC_REQUIRE(e == hipSuccess, "dvc_fgs_filter: copy failed: %s", hipGetErrorString(e));
    }
    const dim3 tgrid_fwd(cdiv(W, 32), cdiv(H, 32), planes), tgrid_bwd(cdiv(H, 32), cdiv(W, 32), planes);
    // elimination coefficients of every iteration, both directions
    if (window) {
        // wave-per-line windowed recurrences, written line-major: row system (lines of length W) and column system (length H)
        fgs_coeff_window(s, wv, W, H, n_guides, num_iter, lams, warm, (long)GHW, row_lm, row_lm + IT, row_lm + 2 * IT);
        fgs_coeff_window(s, wh_t, H, W, n_guides, num_iter, lams, warm, (long)GHW, col_lm, col_lm + IT, col_lm + 2 * IT);
    } else {
        // thread-per-line chains, one launch: rows = lines of length W, M = H of them (the row system's [L = W][M = H] layout is
        // the transposed image's); columns: L = H, M = W; for the scan solver the sets are then transposed into the line-major layout
        const FgsCoeffDir rows_d{wh_t, tmp, tmp + IT, scan ? tmp + 2 * IT : nullptr, W, H};
        const FgsCoeffDir cols_d{wv, tmp2, tmp2 + IT, scan ? tmp2 + 2 * IT : nullptr, H, W};
        hipLaunchKernelGGL(fgs_coeff_kernel, dim3(cdiv(H > W ? H : W, 64), n_guides, 2 * num_iter), dim3(64), 0, s, rows_d, cols_d, num_iter,
                           lams, (long)GHW);
        if (scan) {
            const int ncoef = 3 * num_iter * n_guides;      // planes of one direction's coefficient set
            hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(H, 32), cdiv(W, 32), ncoef), dim3(256), 0, s, tmp, W, H, row_lm);
            hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(W, 32), cdiv(H, 32), ncoef), dim3(256), 0, s, tmp2, H, W, col_lm);
        }
    }
    DVC_CHECK_LAUNCH("dvc_fgs_filter(coefficients)");
    for (int it = 0; it < num_iter; ++it) {
        if (scan) {
            // rows in place on the image; columns on the transposed image
            fgs_solve_scan(s, it == 0 ? src : dst, dst, row_lm + 2 * IT + it * GHW, row_lm + IT + it * GHW, row_lm + it * GHW, W, H, planes, planes_per_guide);
            // columns in place too (r06): 16-column tiles through LDS instead of a transposed copy of the image
            fgs_solve_scan_cols(s, dst, col_lm + 2 * IT + it * GHW, col_lm + IT + it * GHW, col_lm + it * GHW, H, W, planes, planes_per_guide);
        } else {
            // rows: every image row is a line of length W; in the transposed image [W][H] it runs along axis 0
            hipLaunchKernelGGL(transpose_kernel, tgrid_fwd, dim3(256), 0, s, dst, H, W, tr);
            hipLaunchKernelGGL(fgs_solve_kernel, dim3(cdiv(H, 64), planes), dim3(64), 0, s, tr, wh_t, tmp + it * GHW, tmp + IT + it * GHW, W, H,
                               planes_per_guide, lams.v[it], dp);
            hipLaunchKernelGGL(transpose_kernel, tgrid_bwd, dim3(256), 0, s, tr, W, H, dst);
            // columns
            hipLaunchKernelGGL(fgs_solve_kernel, dim3(cdiv(W, 64), planes), dim3(64), 0, s, dst, wv, tmp2 + it * GHW, tmp2 + IT + it * GHW, H, W,
                               planes_per_guide, lams.v[it], dp);
        }
        DVC_CHECK_LAUNCH("dvc_fgs_filter(solve)");
    }
    return 0;
}

// ---- batch_lab2rgb_transpose_mc (utils/util.py:134-151) for one image: skimage.color.lab2rgb in float64,
// clip, * 255, astype(uint8); output HWC.
// r06: two tiers.  As one float64 chain per pixel the launch took 15.5 us at 432x768 — three float64 pow() per pixel, the
// chip's float64 rate.  What is needed of the float64 value v * 255 is only its integer part; a float32 evaluation t of the same
// chain is within  delta = 0.004 S + 0.002  of it, S = the sum of the magnitudes of the three terms of the matrix row (rounding of
// the terms and of their sum ~1.2e-6 S, amplified by at most 12.92 — the slope of the sRGB curve at its steepest, the linear
// piece — times 255; measured over 2e7 random Lab triples: 0.0041 where S <= 5.2; the two piecewise functions are continuous
// at their thresholds to 1e-7, so a different branch in the two evaluations is inside the same bound).  A channel whose t is
// further than delta from every integer (and from the clipping points) is final; a channel closer than that — a few per cent of
// them, and every NaN — is queued in LDS and the workgroup evaluates the queue in float64 with all lanes busy.
// The bytes written are those of the float64 chain in every case (tests/test_tail.py: bit-exact against the oracle).
// one channel of one pixel, float64 (the reference chain)
__device__ __forceinline__ unsigned char lab2rgb_ch_f64(double l_c, double a, double b, int k) {
    // inverse of skimage's xyz_from_rgb (its rgb_from_xyz = scipy.linalg.inv(xyz_from_rgb), float64)
    const double m0 = k == 0 ? 3.240481343200526 : k == 1 ? -0.9692549499965682 : 0.05564663913517716;
    const double m1 = k == 0 ? -1.5371515162713185 : k == 1 ? 1.8759900014898907 : -0.20404133836651123;
    const double m2 = k == 0 ? -0.4985363261688878 : k == 1 ? 0.04155592655829284 : 1.0573110696453443;
    const double l = l_c + 50.0;
    const double fy = (l + 16.0) / 116.0;
    const double fx = a / 500.0 + fy;
    double fz = fy - b / 200.0;
    fz = fz < 0.0 ? 0.0 : fz;
    double xyz[3] = {fx, fy, fz};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double v = xyz[c];
        xyz[c] = v > 0.2068966 ? v * v * v : (v - 16.0 / 116.0) / 7.787;
    }
    xyz[0] *= 0.95047;
    xyz[2] *= 1.08883;
    double v = xyz[0] * m0 + xyz[1] * m1 + xyz[2] * m2;
    v = v > 0.0031308 ? 1.055 * pow(v, 1.0 / 2.4) - 0.055 : v * 12.92;
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    return (unsigned char)(int)(v * 255.0);
}

// float32 tier: the decided bytes in out[0..2] (0 where undecided), returns the mask of the UNDECIDED channels
__device__ __forceinline__ unsigned lab2rgb_px_f32(float l_c, float a, float b, unsigned char* out) {
    const float M[3][3] = {{3.240481343200526f, -1.5371515162713185f, -0.4985363261688878f},
                           {-0.9692549499965682f, 1.8759900014898907f, 0.04155592655829284f},
                           {0.05564663913517716f, -0.20404133836651123f, 1.0573110696453443f}};
    const float l = l_c + 50.f;
    const float fy = (l + 16.f) / 116.f;
    const float fx = a / 500.f + fy;
    float fz = fy - b / 200.f;
    fz = fz < 0.f ? 0.f : fz;
    float xyz[3] = {fx, fy, fz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = xyz[k];
        xyz[k] = v > 0.2068966f ? v * v * v : (v - 16.f / 116.f) / 7.787f;
    }
    xyz[0] *= 0.95047f;
    xyz[2] *= 1.08883f;
    unsigned undecided = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float t0 = xyz[0] * M[k][0], t1 = xyz[1] * M[k][1], t2 = xyz[2] * M[k][2];
        float v = t0 + t1 + t2;
        const float delta = 0.004f * (fabsf(t0) + fabsf(t1) + fabsf(t2)) + 0.002f;
        v = v > 0.0031308f ? 1.055f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(v) * (1.f / 2.4f)) - 0.055f : v * 12.92f;
        const float t = v * 255.f;
        out[k] = 0;
        if (t <= -delta) out[k] = 0;
        else if (t >= 255.f + delta) out[k] = 255;
        else {
            const float fl = floorf(t), q = t - fl;
            // (written so that a NaN anywhere fails the test; t in [-delta, delta] or [255 - delta, 255 + delta] fails it as well)
            if (t >= delta && t <= 255.f - delta && q >= delta && q <= 1.f - delta) out[k] = (unsigned char)(int)fl;
            else undecided |= 1u << k;
        }
    }
    return undecided;
}

// A workgroup = 1024 consecutive pixels, a thread = 4 consecutive ones (12 output bytes = three aligned words).  Undecided
// CHANNELS (not pixels: one float64 pow per entry, the latency of the second phase is one pow) are queued in LDS; their bytes
// are written after the barrier, over the placeholder the word stores put there.
__global__ __launch_bounds__(256) void lab2rgb_u8_kernel(const float* __restrict__ L, const float* __restrict__ ab,
                                                         long HW, unsigned char* rgb) {
    __shared__ int queue[3 * 1024];
    __shared__ int qn;
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * 1024;
    const long i0 = base + 4 * threadIdx.x;
    union { unsigned char b[12]; unsigned w[3]; } out;
    unsigned und = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long i = i0 + j;
        out.b[3 * j] = out.b[3 * j + 1] = out.b[3 * j + 2] = 0;
        if (i < HW) und |= lab2rgb_px_f32(L[i], ab[i], ab[HW + i], out.b + 3 * j) << (3 * j);
    }
    if (i0 + 3 < HW && (reinterpret_cast<uintptr_t>(rgb) & 3) == 0) {
        unsigned* wp = reinterpret_cast<unsigned*>(rgb + 3 * i0);
        wp[0] = out.w[0]; wp[1] = out.w[1]; wp[2] = out.w[2];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < HW) {
                rgb[3 * (i0 + j) + 0] = out.b[3 * j + 0];
                rgb[3 * (i0 + j) + 1] = out.b[3 * j + 1];
                rgb[3 * (i0 + j) + 2] = out.b[3 * j + 2];
            }
    }
    if (und) {
        const int cnt = __popc(und);
        int at = atomicAdd(&qn, cnt);
#pragma unroll
        for (int e = 0; e < 12; ++e)
            if (und & (1u << e)) queue[at++] = (4 * threadIdx.x + e / 3) * 4 + e % 3;
    }
    __syncthreads();
    const int n = qn;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int ent = queue[e], k = ent & 3;
        const long i = base + (ent >> 2);
        rgb[3 * i + k] = lab2rgb_ch_f64((double)L[i], (double)ab[i], (double)ab[HW + i], k);
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
    // r06: the sRGB -> linear step has 256 possible inputs: every workgroup tabulates it (one float64 pow per thread, the same
    // expression on the same argument — the same values) instead of evaluating three per pixel; 16.5 -> ~9 us per 432x768 frame
    __shared__ double lin[256];
    {
        const double v = (double)threadIdx.x / 255.0;
        lin[threadIdx.x] = v > 0.04045 ? pow((v + 0.055) / 1.055, 2.4) : v / 12.92;
    }
    __syncthreads();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        double c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = lin[rgb[3 * i + k]];
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
