// Fused dense exemplar<->frame correlation (models/NonlocalNet.py:469-500) for gfx950.
//
//   f[i][j] = <theta[:,i], phi[:,j]>            P x P cosine affinities, C = 256 deep
//   sim[i]  = max_j f[i][j]
//   p[i][:] = softmax_j( f[i][j] / T )          (T = 1e-10 in test.py:94 -> one-hot at the argmax)
//   y[i][:] = sum_j p[i][j] * B_lab[j][:]       (3 Lab channels of the 4x4-average-pooled exemplar)
//
// The P x P matrix (107.5 MB at 216x384, 1.72 GB at 432x768) is never materialised.  Flash-style:
//   * a wave owns 32 query positions; their 256-deep theta columns live in 128 VGPRs as MFMA B
//     operands (B[k = lane>>5][j = lane&31]) for the whole kernel;
//   * key tiles (32 exemplar positions x 256 channels = 32 KB) are staged through LDS once per
//     workgroup and shared by its 4 waves (128 queries), A[i = lane&31][k = lane>>5] is one
//     conflict-free ds_read_b32 per MFMA;
//   * S^T = phi_tile^T . theta_tile on v_mfma_f32_32x32x2_f32: D[row = key][col = query], so each lane
//     ends up with 16 keys of ONE query -> the row softmax is lane-local: per-lane online state
//     (running max m, running sum l, 3-vector numerator, running max-affinity and its index);
//   * keys are additionally split across workgroups (grid.y) to fill 256 CUs; the 2*nsplit partial
//     states per query are combined by a tiny merge kernel that also writes the x4 nearest upsample.
// fp32 MFMA (exact fma chain) bounds this kernel: 2*P*P*C flop at 157.3 TFLOP/s -> 88.5 us at P=5184.
//
// Softmax arithmetic follows ATen's: s = fl32(f / T) (true IEEE division, so distinct affinities that
// collapse onto one fp32 value at T=1e-10 tie exactly as in the reference), p = exp(s - max s).
#include "common.h"

#include <cmath>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CORR_C 256
#define CORR_KT 32        // keys per LDS tile
#define CORR_QB 128       // queries per workgroup (4 waves x 32)
#define CORR_NF 7         // fields per partial state: m, l, y0, y1, y2, fmax, argmax

// ------------------------------------------------------------------------------------------------
// corr_prepare: per-channel mean over positions, then per-position L2 normalisation over channels
__global__ __launch_bounds__(256) void corr_rowmean_kernel(const float* __restrict__ t, int P,
                                                           float* __restrict__ mean) {
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

__global__ __launch_bounds__(256) void corr_normalize_kernel(const float* __restrict__ t,
                                                             const float* __restrict__ mean, int C, int P,
                                                             float eps, float* __restrict__ out) {
    __shared__ float part[4][64];
    const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + px;
    const int b = blockIdx.y;
    const float* tb = t + (long)b * C * P;
    const float* mb = mean + (long)b * C;
    float* ob = out + (long)b * C * P;
    const bool ok = p < P;
    float s = 0.f;
    if (ok)
        for (int c = g; c < C; c += 4) {
            float v = tb[(long)c * P + p] - mb[c];
            s = fmaf(v, v, s);
        }
    part[g][px] = s;
    __syncthreads();
    float den = sqrtf(part[0][px] + part[1][px] + part[2][px] + part[3][px]) + eps;
    if (ok)
        for (int c = g; c < C; c += 4) ob[(long)c * P + p] = (tb[(long)c * P + p] - mb[c]) / den;
}

extern "C" int dvc_corr_prepare(const float* t_raw, int32_t B, int32_t C, int32_t P, float eps,
                                float* mean_scratch, float* t_out, dvcStream stream) {
    DVC_REQUIRE(t_raw && mean_scratch && t_out && B > 0 && C > 0 && P > 0, "dvc_corr_prepare: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(corr_rowmean_kernel, dim3(B * C), dim3(256), 0, s, t_raw, P, mean_scratch);
    DVC_CHECK_LAUNCH("dvc_corr_prepare(mean)");
    hipLaunchKernelGGL(corr_normalize_kernel, dim3(cdiv(P, 64), B), dim3(256), 0, s, t_raw, mean_scratch, C,
                       P, eps, t_out);
    DVC_CHECK_LAUNCH("dvc_corr_prepare(normalise)");
    return 0;
}

// ------------------------------------------------------------------------------------------------
struct CorrArgs {
    const float* theta;
    const float* phi;
    const float* blab;
    const float* fmax_in;  // WTA pass only: row maxima from the first pass
    float* part;           // [B][nslot][CORR_NF][P]
    float T, invT, wta_scale;
    int P, ntiles, tiles_per_split, nslot;
};

template <bool WTA, bool VEC4>
__global__ __launch_bounds__(256, 2) void corr_fwd_kernel(CorrArgs a) {
    __shared__ __attribute__((aligned(16))) float ks[CORR_C * CORR_KT + 3 * CORR_KT];
    float* bl = ks + CORR_C * CORR_KT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, split = blockIdx.y;
    const int P = a.P;
    const int query = blockIdx.x * CORR_QB + wave * 32 + l31;
    const bool qvalid = query < P;
    const float* th = a.theta + (long)b * CORR_C * P;
    const float* ph = a.phi + (long)b * CORR_C * P;
    const float* blb = a.blab + (long)b * 3 * P;

    // query fragment: B[k = 2s+hi][j = l31] for s = 0..127
    float qreg[CORR_C / 2];
#pragma unroll
    for (int s = 0; s < CORR_C / 2; ++s) qreg[s] = qvalid ? th[(long)(2 * s + hi) * P + query] : 0.f;

    float m = -INFINITY, l = 0.f, y0 = 0.f, y1 = 0.f, y2 = 0.f, fmax = -INFINITY;
    int amax = 0;
    float fq = 0.f;
    if (WTA) fq = qvalid ? a.fmax_in[(long)b * P + query] : 0.f;

    const int t0 = split * a.tiles_per_split;
    const int t1 = min(a.ntiles, t0 + a.tiles_per_split);

    // register prefetch of the next key tile (issued before the MFMA loop, committed to LDS after it)
    constexpr int KR = VEC4 ? (CORR_C * CORR_KT / 4) / 256 : (CORR_C * CORR_KT) / 256;
    typedef typename std::conditional<VEC4, float4, float>::type kreg_t;
    kreg_t kr[KR];
    float blr = 0.f;
    auto issue = [&](int t) {
        const int k0 = t * CORR_KT;
        if (VEC4) {
#pragma unroll
            for (int i = 0; i < KR; ++i) {
                int q = tid + i * 256;
                int row = q >> 3, c4 = (q & 7) * 4;
                bool ok = k0 + c4 < P;
                *reinterpret_cast<float4*>(&kr[i]) =
                    *reinterpret_cast<const float4*>(ph + (ok ? (unsigned)(row * P + k0 + c4) : 0u));
            }
        } else {
#pragma unroll
            for (int i = 0; i < KR; ++i) {
                int e = tid + i * 256;
                int row = e >> 5, c = e & 31;
                bool ok = k0 + c < P;
                *reinterpret_cast<float*>(&kr[i]) = ph[ok ? (unsigned)(row * P + k0 + c) : 0u];
            }
        }
        if (tid < 3 * CORR_KT) {
            int c = tid >> 5, j = tid & 31;
            blr = (k0 + j < P) ? blb[(long)c * P + k0 + j] : 0.f;
        }
    };
    auto commit = [&](int t) {
        const int k0 = t * CORR_KT;
        if (VEC4) {
#pragma unroll
            for (int i = 0; i < KR; ++i) {
                int q = tid + i * 256;
                int row = q >> 3, c4 = (q & 7) * 4;
                float4 v = *reinterpret_cast<float4*>(&kr[i]);
                if (!(k0 + c4 < P)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(ks + row * CORR_KT + c4) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < KR; ++i) {
                int e = tid + i * 256;
                int c = e & 31;
                float v = *reinterpret_cast<float*>(&kr[i]);
                ks[e] = (k0 + c < P) ? v : 0.f;
            }
        }
        if (tid < 3 * CORR_KT) bl[tid] = blr;
    };

    if (t0 < t1) {
        issue(t0);
        commit(t0);
    }
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const int k0 = t * CORR_KT;
        const bool has_next = t + 1 < t1;
        if (has_next) issue(t + 1);

        // ---- S^T tile: 128 dependent MFMAs (K = 256), A from LDS, B from registers
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* kp = ks + hi * CORR_KT + l31;
#pragma unroll
        for (int s = 0; s < CORR_C / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s * CORR_KT], qreg[s], acc, 0, 0, 0);

        // ---- lane-local online softmax over this lane's 16 keys of its query
        float tilemax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            float f = acc[r];
            const bool kvalid = key < P;
            if (kvalid && f > fmax) {  // strict '>' keeps the lowest index on ties
                fmax = f;
                amax = key;
            }
            if (WTA) f = (f == fq) ? f : f * a.wta_scale;
            f = kvalid ? f : -INFINITY;
            acc[r] = f;
            tilemax = fmaxf(tilemax, f);
        }
        if (tilemax > -INFINITY) {
            const float tmax = tilemax / a.T;
            if (tmax > m) {
                const float sc = expf(m - tmax);  // m == -inf on first use -> 0
                l *= sc;
                y0 *= sc;
                y1 *= sc;
                y2 *= sc;
                m = tmax;
            }
            // candidates: exp(f/T - m) can only be non-zero in fp32 within ~104 of the max; the slack
            // also covers the rounding of f*invT vs the exact division (ulp(m) can be ~1e3 at T=1e-10)
            const float slack = 120.f + fabsf(m) * 4.8e-7f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = acc[r];
                if (f * a.invT - m > -slack) {
                    const float tt = f / a.T;
                    const float p = expf(tt - m);
                    const int kl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    l += p;
                    y0 = fmaf(p, bl[kl], y0);
                    y1 = fmaf(p, bl[CORR_KT + kl], y1);
                    y2 = fmaf(p, bl[2 * CORR_KT + kl], y2);
                }
            }
        }
        if (has_next) {
            __syncthreads();  // all waves done with this tile's LDS reads
            commit(t + 1);
            __syncthreads();
        }
    }

    // ---- write this lane's partial state: slot = split*2 + hi
    if (qvalid) {
        float* pp = a.part + (((long)b * a.nslot + split * 2 + hi) * CORR_NF) * P + query;
        pp[0] = m;
        pp[(long)P] = l;
        pp[2L * P] = y0;
        pp[3L * P] = y1;
        pp[4L * P] = y2;
        pp[5L * P] = fmax;
        pp[6L * P] = __int_as_float(amax);
    }
}

// merge the 2*nsplit partial states of each query; write small + x4-upsampled outputs
__global__ __launch_bounds__(256) void corr_merge_kernel(const float* __restrict__ part, int nslot, int P,
                                                         int h, int w, float* __restrict__ y_small,
                                                         float* __restrict__ sim_small,
                                                         float* __restrict__ y_up,
                                                         float* __restrict__ sim_up,
                                                         int* __restrict__ argmax) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (q >= P) return;
    const float* pb = part + (long)b * nslot * CORR_NF * P + q;
    float M = -INFINITY, F = -INFINITY;
    int A = 0x7fffffff;
    for (int s = 0; s < nslot; ++s) {
        const float* ps = pb + (long)s * CORR_NF * P;
        M = fmaxf(M, ps[0]);
        float f = ps[5L * P];
        int ai = __float_as_int(ps[6L * P]);
        if (f > F || (f == F && ai < A)) {
            F = f;
            A = ai;
        }
    }
    float L = 0.f, Y0 = 0.f, Y1 = 0.f, Y2 = 0.f;
    for (int s = 0; s < nslot; ++s) {
        const float* ps = pb + (long)s * CORR_NF * P;
        float ms = ps[0];
        if (ms == -INFINITY) continue;
        float sc = expf(ms - M);
        L = fmaf(ps[(long)P], sc, L);
        Y0 = fmaf(ps[2L * P], sc, Y0);
        Y1 = fmaf(ps[3L * P], sc, Y1);
        Y2 = fmaf(ps[4L * P], sc, Y2);
    }
    const float yv[3] = {Y0 / L, Y1 / L, Y2 / L};
    if (y_small)
        for (int c = 0; c < 3; ++c) y_small[((long)b * 3 + c) * P + q] = yv[c];
    if (sim_small) sim_small[(long)b * P + q] = F;
    if (argmax) argmax[(long)b * P + q] = A;
    const int qy = q / w, qx = q - qy * w;
    const long W4 = 4L * w, HW16 = 16L * P;
    if (y_up) {
        for (int c = 0; c < 3; ++c) {
            float4 v = make_float4(yv[c], yv[c], yv[c], yv[c]);
            float* o = y_up + ((long)b * 3 + c) * HW16 + (4L * qy) * W4 + 4L * qx;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) *reinterpret_cast<float4*>(o + dy * W4) = v;
        }
    }
    if (sim_up) {
        float4 v = make_float4(F, F, F, F);
        float* o = sim_up + (long)b * HW16 + (4L * qy) * W4 + 4L * qx;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) *reinterpret_cast<float4*>(o + dy * W4) = v;
    }
}

static void corr_split(int B, int P, int* ntiles, int* tps, int* nsplit) {
    int qblocks = cdiv(P, CORR_QB);
    *ntiles = cdiv(P, CORR_KT);
    int want = 512 / (qblocks * B);  // 2 resident workgroups per CU x 256 CUs
    if (want < 1) want = 1;
    if (want > *ntiles) want = *ntiles;
    *tps = cdiv(*ntiles, want);
    *nsplit = cdiv(*ntiles, *tps);
}

extern "C" size_t dvc_corr_workspace_bytes(int32_t B, int32_t P) {
    if (B <= 0 || P <= 0) return 0;
    int ntiles, tps, nsplit;
    corr_split(B, P, &ntiles, &tps, &nsplit);
    size_t part = (size_t)B * nsplit * 2 * CORR_NF * P * sizeof(float);
    size_t fmax = (size_t)B * P * sizeof(float);
    return part + fmax + 256;
}

extern "C" int dvc_corr_fwd(const float* theta, const float* phi, const float* blab, float temperature,
                            float wta_scale, int32_t B, int32_t C, int32_t h, int32_t w, float* y_small,
                            float* sim_small, float* y_up, float* sim_up, int32_t* argmax,
                            void* workspace, size_t workspace_bytes, dvcStream stream) {
    DVC_REQUIRE(theta && phi && blab && workspace, "dvc_corr_fwd: null argument");
    DVC_REQUIRE(C == CORR_C, "dvc_corr_fwd: C must be %d (got %d)", CORR_C, C);
    DVC_REQUIRE(B > 0 && h > 0 && w > 0, "dvc_corr_fwd: bad shape");
    DVC_REQUIRE(temperature > 0.f && std::isfinite(temperature), "dvc_corr_fwd: temperature must be > 0");
    const int P = h * w;
    DVC_REQUIRE(workspace_bytes >= dvc_corr_workspace_bytes(B, P), "dvc_corr_fwd: workspace too small");
    DVC_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "dvc_corr_fwd: workspace must be 16-byte aligned");
    if (y_up || sim_up)
        DVC_REQUIRE(((reinterpret_cast<uintptr_t>(y_up) | reinterpret_cast<uintptr_t>(sim_up)) & 15) == 0,
                    "dvc_corr_fwd: upsampled outputs must be 16-byte aligned");
    CorrArgs a;
    a.theta = theta; a.phi = phi; a.blab = blab;
    a.T = temperature; a.invT = 1.0f / temperature; a.wta_scale = wta_scale; a.P = P;
    int nsplit;
    corr_split(B, P, &a.ntiles, &a.tiles_per_split, &nsplit);
    a.nslot = nsplit * 2;
    a.part = reinterpret_cast<float*>(workspace);
    float* fmax_buf = a.part + (size_t)B * a.nslot * CORR_NF * P;
    a.fmax_in = fmax_buf;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(cdiv(P, CORR_QB), nsplit, B);
    const bool vec4 = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(phi) & 15) == 0);
    dim3 mgrid(cdiv(P, 256), B);
    const bool wta = wta_scale != 1.0f;
    if (wta) {
        // pass 1: row maxima only (identical MFMA order => `f == rowmax` is exact in pass 2)
        if (vec4) hipLaunchKernelGGL((corr_fwd_kernel<false, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((corr_fwd_kernel<false, false>), grid, dim3(256), 0, s, a);
        DVC_CHECK_LAUNCH("dvc_corr_fwd(pass1)");
        hipLaunchKernelGGL(corr_merge_kernel, mgrid, dim3(256), 0, s, a.part, a.nslot, P, h, w,
                           (float*)nullptr, fmax_buf, (float*)nullptr, (float*)nullptr, (int*)nullptr);
        DVC_CHECK_LAUNCH("dvc_corr_fwd(merge1)");
        if (vec4) hipLaunchKernelGGL((corr_fwd_kernel<true, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((corr_fwd_kernel<true, false>), grid, dim3(256), 0, s, a);
    } else {
        if (vec4) hipLaunchKernelGGL((corr_fwd_kernel<false, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((corr_fwd_kernel<false, false>), grid, dim3(256), 0, s, a);
    }
    DVC_CHECK_LAUNCH("dvc_corr_fwd");
    hipLaunchKernelGGL(corr_merge_kernel, mgrid, dim3(256), 0, s, a.part, a.nslot, P, h, w, y_small,
                       sim_small, y_up, sim_up, argmax);
    DVC_CHECK_LAUNCH("dvc_corr_fwd(merge)");
    return 0;
}
