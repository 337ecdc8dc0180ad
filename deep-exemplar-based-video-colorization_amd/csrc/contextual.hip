// Contextual loss (models/ContextualLoss.py:29-126, used by train.py:649-668 on relu3_1 / relu4_1 / relu5_1 features):
// the other N x N cosine-affinity consumer of the reference.  Per image, with X = predicted and
// Y = exemplar features [C][N]:
//     mu = mean_j Y[:, j];  Xn, Yn = (. - mu) / (||.||_C + eps)                                   ContextualLoss.py:48-61
//     S = Xn^T Yn;  d = 1 - S;  a_i = min_j d_ij + 1e-5;  w = exp((1 - d / a_i) / h);  A = w / sum_j w       :64-72
//     ContextualLoss_forward:  CX = mean_i max_j A_ij        ContextualLoss:  CX = mean_j max_i A_ij        :75 / :125
//     loss = -log CX
// The GEMMs (S and the two gradient products) run on the 1x1-convolution engine from the host side
// (dvc_amd/contextual.py), in blocks of R rows of S; this file holds the bandwidth-bound pieces in between, forward
// and backward.  Feature maps on this path have N <= 1296 positions (27x48), so a block of S (R x N) is small.
#include "common.h"

#include <cmath>

// ---- centre (with a GIVEN or own per-channel mean) + L2-normalise over channels; also returns the norms (for backward)
__global__ __launch_bounds__(256) void cx_rowmean_kernel(const float* __restrict__ t, int P, float* __restrict__ mean) {
    __shared__ double red[4];
    const float* row = t + (long)blockIdx.x * P;
    double s = 0.0;
    for (int i = threadIdx.x; i < P; i += 256) s += (double)row[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) mean[blockIdx.x] = (float)((red[0] + red[1] + red[2] + red[3]) / (double)P);
}

// (r06: 16 channel groups x 64 positions per workgroup and four channel rows in flight per thread.  With 4 groups and one load
// at a time a thread walked C / 4 = 128 dependent round trips twice and the grid was P / 64 x B workgroups — 72 us for 16 maps of
// 512 x 13 x 24, 42 % of the contextual loss's GPU time at relu5_1)
#define CX_NG 16
__global__ __launch_bounds__(64 * CX_NG) void cx_normalize_kernel(const float* __restrict__ t, const float* __restrict__ mean, int C,
                                                                  int P, float eps, float* __restrict__ out,
                                                                  float* __restrict__ norm) {
    __shared__ float part[CX_NG][64];
    const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + px;
    const int b = blockIdx.y;
    const float* tb = t + (long)b * C * P;
    const float* mb = mean ? mean + (long)b * C : nullptr;
    float* ob = out + (long)b * C * P;
    const bool ok = p < P;
    float s = 0.f;
    if (ok) {
        int c = g;
        for (; c + 3 * CX_NG < C; c += 4 * CX_NG) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = tb[(long)(c + u * CX_NG) * P + p] - (mb ? mb[c + u * CX_NG] : 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) s = fmaf(v[u], v[u], s);
        }
        for (; c < C; c += CX_NG) {
            const float v = tb[(long)c * P + p] - (mb ? mb[c] : 0.f);
            s = fmaf(v, v, s);
        }
    }
    part[g][px] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < CX_NG; ++k) tot += part[k][px];        // fixed order
    const float n = sqrtf(tot);
    if (ok) {
        if (g == 0 && norm) norm[(long)b * P + p] = n;
        int c = g;
        for (; c + 3 * CX_NG < C; c += 4 * CX_NG) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = tb[(long)(c + u * CX_NG) * P + p] - (mb ? mb[c + u * CX_NG] : 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) ob[(long)(c + u * CX_NG) * P + p] = v[u] / (n + eps);
        }
        for (; c < C; c += CX_NG) ob[(long)c * P + p] = (tb[(long)c * P + p] - (mb ? mb[c] : 0.f)) / (n + eps);
    }
}

extern "C" int dvc_cx_prepare(const float* x, const float* mean_in, int32_t centre, int32_t B, int32_t C, int32_t P, float eps,
                              float* mean_out, float* norm_out, float* out, dvcStream stream) {
    DVC_REQUIRE(x && out && B > 0 && C > 0 && P > 0, "dvc_cx_prepare: bad argument");
    DVC_REQUIRE(!centre || mean_in || mean_out, "dvc_cx_prepare: centring needs a mean to use or a place to put its own");
    hipStream_t s = (hipStream_t)stream;
    const float* mean = nullptr;
    if (centre) {
        mean = mean_in;
        if (!mean) {
            hipLaunchKernelGGL(cx_rowmean_kernel, dim3(B * C), dim3(256), 0, s, x, P, mean_out);
            DVC_CHECK_LAUNCH("dvc_cx_prepare(mean)");
            mean = mean_out;
        }
    }
    hipLaunchKernelGGL(cx_normalize_kernel, dim3(cdiv(P, 64), B), dim3(64 * CX_NG), 0, s, x, mean, C, P, eps, out, norm_out);
    DVC_CHECK_LAUNCH("dvc_cx_prepare(normalise)");
    return 0;
}

// ---- per row of a block of S (rows x N): a_i = (1 - max_j S_ij) + 1e-5, arg-max j*_i (lowest index on ties),
// l_i = sum_j w_ij, r_i = max_j A_ij = w_{i j*} / l_i, E_i = sum_j A_ij d_ij.  One workgroup per row.
__device__ __forceinline__ float cx_w(float s, float a, float h) { return expf((1.f - (1.f - s) / a) / h); }

// (every kernel below takes a batch: blockIdx.y / .z = image; S_bs = elements between the images' S blocks, row_bs = between
// their per-row arrays, which are [B][Nx] slices at the block's first row)
__global__ __launch_bounds__(256) void cx_rows_kernel(const float* __restrict__ S, long S_bs, long row_bs, int N, float h,
                                                      float* __restrict__ a_out, int* __restrict__ jstar,
                                                      float* __restrict__ l_out, float* __restrict__ r_out,
                                                      float* __restrict__ e_out) {
    __shared__ float redf[4];
    __shared__ int redi[4];
    __shared__ float red2[2][4];
    const int row = blockIdx.x;
    const long ro = (long)blockIdx.y * row_bs;
    a_out += ro; jstar += ro; l_out += ro; r_out += ro; e_out += ro;
    const float* s = S + (long)blockIdx.y * S_bs + (long)row * N;
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int j = threadIdx.x; j < N; j += 256) {
        const float v = s[j];
        if (v > m) { m = v; mi = j; }       // ascending j per thread: the lowest index wins ties
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(m, off, 64);
        const int oi = __shfl_xor(mi, off, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { redf[threadIdx.x >> 6] = m; redi[threadIdx.x >> 6] = mi; }
    __syncthreads();
    m = redf[0]; mi = redi[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (redf[k] > m || (redf[k] == m && redi[k] < mi)) { m = redf[k]; mi = redi[k]; }
    const float dmin = 1.f - m;
    const float a = dmin + 1e-5f;
    float l = 0.f, e = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) {
        const float w = cx_w(s[j], a, h);
        l += w;
        e = fmaf(w, 1.f - s[j], e);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { l += __shfl_xor(l, off, 64); e += __shfl_xor(e, off, 64); }
    if ((threadIdx.x & 63) == 0) { red2[0][threadIdx.x >> 6] = l; red2[1][threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        l = (red2[0][0] + red2[0][1]) + (red2[0][2] + red2[0][3]);
        e = (red2[1][0] + red2[1][1]) + (red2[1][2] + red2[1][3]);
        a_out[row] = a;
        jstar[row] = mi;
        l_out[row] = l;
        r_out[row] = cx_w(m, a, h) / l;
        e_out[row] = e / l;
    }
}

extern "C" int dvc_cx_rows(const float* S, int32_t nb, int64_t S_bs, int64_t row_bs, int32_t rows, int32_t N, float h, float* a,
                           int32_t* jstar, float* l, float* r, float* e, dvcStream stream) {
    DVC_REQUIRE(S && a && jstar && l && r && e && nb > 0 && nb < 65536 && rows > 0 && N > 0 && h > 0.f, "dvc_cx_rows: bad argument");
    hipLaunchKernelGGL(cx_rows_kernel, dim3(rows, nb), dim3(256), 0, (hipStream_t)stream, S, (long)S_bs, (long)row_bs, N, h, a, jstar,
                       l, r, e);
    DVC_CHECK_LAUNCH("dvc_cx_rows");
    return 0;
}

// ---- ContextualLoss (max over ROWS for every column): running column maxima of A over the row blocks.
// Workgroup = 64 columns x 16 row groups (r04: one thread per column walking all rows serially left a 16-image batch with 96
// workgroups of 1300-step latency chains); group g walks rows g, g + 16, ... in ascending order (strict '>' keeps its lowest
// row on ties), the groups are combined through LDS: larger value, then LOWER row index — the result of the serial scan.
// (r06: 16 row groups and four rows in flight per thread — with 4 groups a thread walked rows / 4 load -> exp -> divide steps one
// at a time: 31 us for 16 blocks of 312 x 312)
__global__ __launch_bounds__(64 * CX_NG) void cx_colmax_kernel(const float* __restrict__ S, long S_bs, long row_bs,
                                                               const float* __restrict__ a, const float* __restrict__ l, int rows, int N,
                                                               int i0, float h, float* __restrict__ cmax, int* __restrict__ cargi) {
    __shared__ float sv[CX_NG][64];
    __shared__ int si[CX_NG][64];
    const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tx;
    const bool ok = j < N;
    S += (long)blockIdx.y * S_bs; a += (long)blockIdx.y * row_bs; l += (long)blockIdx.y * row_bs;
    cmax += (long)blockIdx.y * N; cargi += (long)blockIdx.y * N;
    // (group 0 carries the running maximum of the earlier row blocks: their rows are lower, so they win ties below)
    float best = (ok && g == 0) ? cmax[j] : -INFINITY;
    int bi = (ok && g == 0) ? cargi[j] : 0x7fffffff;
    if (ok) {
        int i = g;
        for (; i + 3 * CX_NG < rows; i += 4 * CX_NG) {
            float sv4[4], av[4], lv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sv4[u] = S[(long)(i + u * CX_NG) * N + j];
                av[u] = a[i + u * CX_NG];
                lv[u] = l[i + u * CX_NG];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {                   // ascending rows: strict '>' keeps the lowest row on ties
                const float v = cx_w(sv4[u], av[u], h) / lv[u];
                if (v > best) { best = v; bi = i0 + i + u * CX_NG; }
            }
        }
        for (; i < rows; i += CX_NG) {
            const float v = cx_w(S[(long)i * N + j], a[i], h) / l[i];
            if (v > best) { best = v; bi = i0 + i; }
        }
    }
    sv[g][tx] = best;
    si[g][tx] = bi;
    __syncthreads();
    if (g == 0 && ok) {
#pragma unroll
        for (int k = 1; k < CX_NG; ++k) {
            const float v = sv[k][tx];
            const int vi = si[k][tx];
            // (group 0's carried-in value may tie a later row's: the earlier block's row index is lower and is kept)
            if (v > best || (v == best && vi < bi)) { best = v; bi = vi; }
        }
        cmax[j] = best;
        cargi[j] = bi;
    }
}

extern "C" int dvc_cx_colmax(const float* S, int32_t nb, int64_t S_bs, int64_t row_bs, const float* a, const float* l, int32_t rows,
                             int32_t N, int32_t row0, float h, float* cmax, int32_t* cargi, dvcStream stream) {
    DVC_REQUIRE(S && a && l && cmax && cargi && nb > 0 && nb < 65536 && rows > 0 && N > 0 && h > 0.f, "dvc_cx_colmax: bad argument");
    hipLaunchKernelGGL(cx_colmax_kernel, dim3(cdiv(N, 64), nb), dim3(64 * CX_NG), 0, (hipStream_t)stream, S, (long)S_bs, (long)row_bs, a, l,
                       rows, N, row0, h, cmax, cargi);
    DVC_CHECK_LAUNCH("dvc_cx_colmax");
    return 0;
}

// ---- loss = -log(mean(v)) and the factor every gradient carries: gscale = -1 / (n * mean(v)) = d loss / d v_k
__global__ __launch_bounds__(256) void cx_finish_kernel(const float* __restrict__ v, int n, float* __restrict__ loss,
                                                        float* __restrict__ gscale) {
    __shared__ double red[4];
    double s = 0.0;
    v += (long)blockIdx.x * n; loss += blockIdx.x; gscale += blockIdx.x;      // image blockIdx.x of a [B][n] array
    for (int i = threadIdx.x; i < n; i += 256) s += (double)v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double cx = (red[0] + red[1] + red[2] + red[3]) / (double)n;
        *loss = (float)(-log(cx));
        *gscale = (float)(-1.0 / ((double)n * cx));
    }
}

extern "C" int dvc_cx_finish(const float* v, int32_t nb, int32_t n, float* loss, float* gscale, dvcStream stream) {
    DVC_REQUIRE(v && loss && gscale && nb > 0 && n > 0, "dvc_cx_finish: bad argument");
    hipLaunchKernelGGL(cx_finish_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, v, n, loss, gscale);
    DVC_CHECK_LAUNCH("dvc_cx_finish");
    return 0;
}

// ---- backward, ContextualLoss only: per row i, T_i = sum_{k: argmax_i'(A_i'k) = i} A_ik and Q_i = the same sum weighted
// by d_ik (the columns whose maximum sits in row i).  One workgroup per row, columns in a fixed order (deterministic).
__global__ __launch_bounds__(256) void cx_rows_tq_kernel(const float* __restrict__ S, long S_bs, long row_bs, long tq_bs,
                                                         const float* __restrict__ a, const float* __restrict__ l,
                                                         const int* __restrict__ cargi, int N, int i0, float h,
                                                         float* __restrict__ t_out, float* __restrict__ q_out) {
    __shared__ float red[2][4];
    const int row = blockIdx.x;
    a += (long)blockIdx.y * row_bs; l += (long)blockIdx.y * row_bs; cargi += (long)blockIdx.y * N;
    t_out += (long)blockIdx.y * tq_bs; q_out += (long)blockIdx.y * tq_bs;
    const float* s = S + (long)blockIdx.y * S_bs + (long)row * N;
    const float ai = a[row], li = l[row];
    float t = 0.f, q = 0.f;
    for (int j = threadIdx.x; j < N; j += 256)
        if (cargi[j] == i0 + row) {
            const float A = cx_w(s[j], ai, h) / li;
            t += A;
            q = fmaf(A, 1.f - s[j], q);
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { t += __shfl_xor(t, off, 64); q += __shfl_xor(q, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = t; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        t_out[row] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        q_out[row] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// ---- dS for a block of rows, both layouts (row-major for products with Xn, transposed for products with Yn; the GEMM
// engine wants K-major operands).  z_ik = (1 - d_ik / a_i) / h, A = softmax_k z, a_i = d_{i j*} + 1e-5:
//   d z_ik / d S_ik = 1 / (h a_i),   d z_ik / d S_{i j*} (through a_i) = -d_ik / (h a_i^2)
// mode 0 (ContextualLoss_forward, loss = -log mean_i A_{i j*}):  dz_ik = g r_i ([k = j*_i] - A_ik)
// mode 1 (ContextualLoss,          loss = -log mean_k max_i A_ik): dz_ik = g A_ik ([cargi_k = i] - T_i)
//   dS_ik = dz_ik / (h a_i) - [k = j*_i] sum_m dz_im d_im / (h a_i^2)
//   with  sum_m dz_im d_im = g r_i (d_{i j*} - E_i)  (mode 0)  |  g (Q_i - T_i E_i)  (mode 1);  g = *gscale * gout
__global__ __launch_bounds__(256) void cx_ds_kernel(const float* __restrict__ S, const float* __restrict__ a,
                                                    const float* __restrict__ l, const float* __restrict__ r,
                                                    const float* __restrict__ e, const int* __restrict__ jstar,
                                                    const int* __restrict__ cargi, const float* __restrict__ tt,
                                                    const float* __restrict__ qq, const float* __restrict__ gscale, float gout,
                                                    int mode, int rows, int N, int i0, int ldt, float h,
                                                    float* __restrict__ dS, float* __restrict__ dST, long S_bs, long row_bs,
                                                    long tq_bs) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 64, ib = blockIdx.y * 64;
    const int j = j0 + tx;
    const bool jok = j < N;
    {   // image blockIdx.z
        const long b = blockIdx.z;
        S += b * S_bs; a += b * row_bs; l += b * row_bs; r += b * row_bs; e += b * row_bs; jstar += b * row_bs;
        if (cargi) cargi += b * N;
        if (tt) { tt += b * tq_bs; qq += b * tq_bs; }
        gscale += b;
        if (dS) dS += b * S_bs;
        if (dST) dST += b * (long)N * ldt;
    }
    const float g = *gscale * gout;
    const int cj = (mode == 1 && jok) ? cargi[j] : -1;
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
        const int il = ty * 16 + rr, i = ib + il;
        float v = 0.f;
        if (i < rows && jok) {
            const float ai = a[i], s = S[(long)i * N + j];
            const float A = cx_w(s, ai, h) / l[i];
            const bool star = jstar[i] == j;
            float dz, sum_dz_d;
            if (mode == 0) {
                dz = g * r[i] * ((star ? 1.f : 0.f) - A);
                sum_dz_d = g * r[i] * ((ai - 1e-5f) - e[i]);
            } else {
                dz = g * A * ((cj == i0 + i ? 1.f : 0.f) - tt[i]);
                sum_dz_d = g * (qq[i] - tt[i] * e[i]);
            }
            v = dz / (h * ai);
            if (star) v -= sum_dz_d / (h * ai * ai);
            if (dS) dS[(long)i * N + j] = v;
        }
        tile[il][tx] = v;
    }
    if (!dST) return;       // (uniform: row-major only — the caller's GEMM takes dS transposed as it is)
    __syncthreads();
    const int i = ib + tx;
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
        const int jl = ty * 16 + rr, jj = j0 + jl;
        if (jj < N && i < ldt) dST[(long)jj * ldt + i] = (i < rows) ? tile[tx][jl] : 0.f;
    }
}

extern "C" int dvc_cx_ds(const float* S, int32_t nb, int64_t S_bs, int64_t row_bs, int64_t tq_bs, const float* a, const float* l,
                         const float* r, const float* e, const int32_t* jstar, const int32_t* cargi, const float* t, const float* q,
                         const float* gscale, float gout, int32_t mode, int32_t rows, int32_t N, int32_t row0, int32_t ld_t, float h,
                         float* dS, float* dST, dvcStream stream) {
    DVC_REQUIRE(S && a && l && r && e && jstar && gscale && (dS || dST) && nb > 0 && nb < 65536 && rows > 0 && N > 0 && ld_t >= rows && h > 0.f,
                "dvc_cx_ds: bad argument");
    DVC_REQUIRE(mode == 0 || (mode == 1 && cargi && t && q), "dvc_cx_ds: mode 1 needs the column arg-max and the T / Q row sums");
    hipLaunchKernelGGL(cx_ds_kernel, dim3(cdiv(N, 64), cdiv(ld_t, 64), nb), dim3(256), 0, (hipStream_t)stream, S, a, l, r, e, jstar,
                       cargi, t, q, gscale, gout, mode, rows, N, row0, ld_t, h, dS, dST, (long)S_bs, (long)row_bs, (long)tq_bs);
    DVC_CHECK_LAUNCH("dvc_cx_ds");
    return 0;
}

extern "C" int dvc_cx_rows_tq(const float* S, int32_t nb, int64_t S_bs, int64_t row_bs, int64_t tq_bs, const float* a, const float* l,
                              const int32_t* cargi, int32_t rows, int32_t N, int32_t row0, float h, float* t, float* q,
                              dvcStream stream) {
    DVC_REQUIRE(S && a && l && cargi && t && q && nb > 0 && nb < 65536 && rows > 0 && N > 0 && h > 0.f, "dvc_cx_rows_tq: bad argument");
    hipLaunchKernelGGL(cx_rows_tq_kernel, dim3(rows, nb), dim3(256), 0, (hipStream_t)stream, S, (long)S_bs, (long)row_bs, (long)tq_bs, a,
                       l, cargi, N, row0, h, t, q);
    DVC_CHECK_LAUNCH("dvc_cx_rows_tq");
    return 0;
}

// ---- backward of  xn = xc / (||xc|| + eps):  d xc_k = d xn_k / (n + eps) - xn_k (xn . d xn) / n
__global__ __launch_bounds__(64 * CX_NG) void cx_normalize_bwd_kernel(const float* __restrict__ xn, const float* __restrict__ norm,
                                                                      const float* __restrict__ dxn, int C, int P, float eps,
                                                                      float* __restrict__ dx) {
    __shared__ float part[CX_NG][64];
    const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + px;
    const int b = blockIdx.y;
    const long base = (long)b * C * P + p;
    const bool ok = p < P;
    float s = 0.f;
    if (ok) {
        int c = g;
        for (; c + 3 * CX_NG < C; c += 4 * CX_NG) {
            float a[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = xn[base + (long)(c + u * CX_NG) * P];
                d[u] = dxn[base + (long)(c + u * CX_NG) * P];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) s = fmaf(a[u], d[u], s);
        }
        for (; c < C; c += CX_NG) s = fmaf(xn[base + (long)c * P], dxn[base + (long)c * P], s);
    }
    part[g][px] = s;
    __syncthreads();
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < CX_NG; ++k) dot += part[k][px];        // fixed order
    if (ok) {
        const float n = norm[(long)b * P + p];
        const float inv = 1.f / (n + eps), k = n > 0.f ? dot / n : 0.f;
        int c = g;
        for (; c + 3 * CX_NG < C; c += 4 * CX_NG) {
            float a[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = xn[base + (long)(c + u * CX_NG) * P];
                d[u] = dxn[base + (long)(c + u * CX_NG) * P];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) dx[base + (long)(c + u * CX_NG) * P] = d[u] * inv - a[u] * k;
        }
        for (; c < C; c += CX_NG) {
            const long o = base + (long)c * P;
            dx[o] = dxn[o] * inv - xn[o] * k;
        }
    }
}

extern "C" int dvc_cx_normalize_bwd(const float* xn, const float* norm, const float* dxn, int32_t B, int32_t C, int32_t P, float eps,
                                    float* dx, dvcStream stream) {
    DVC_REQUIRE(xn && norm && dxn && dx && B > 0 && C > 0 && P > 0, "dvc_cx_normalize_bwd: bad argument");
    hipLaunchKernelGGL(cx_normalize_bwd_kernel, dim3(cdiv(P, 64), B), dim3(64 * CX_NG), 0, (hipStream_t)stream, xn, norm, dxn, C, P, eps, dx);
    DVC_CHECK_LAUNCH("dvc_cx_normalize_bwd");
    return 0;
}
