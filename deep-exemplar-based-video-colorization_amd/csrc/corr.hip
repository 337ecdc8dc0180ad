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
//   * the (query block, key tile) work units are dealt out in equal contiguous ranges to 512 workgroups
//     (stream-K style: 2 resident per CU, same tile count +-1 for every workgroup whatever P is); the
//     partial states per query are combined by a tiny merge kernel that also writes the x4 upsample.
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

// float4 form (P % 4 == 0, aligned): workgroup = 32 positions (8 lanes x float4: one 128-byte line per channel row) x 32
// channel groups — 162 workgroups at P = 5184 with 8 channels per thread and pass, where the scalar form below runs 81
// workgroups of 64-deep serial loops (27 us for 10.6 MB: pure latency).
__global__ __launch_bounds__(256) void corr_normalize_v4_kernel(const float* __restrict__ t, const float* __restrict__ mean,
                                                                int C, int P, float eps, float* __restrict__ out) {
    __shared__ float4 part[32][8];
    const int px4 = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int p = (blockIdx.x * 8 + px4) * 4;
    const int b = blockIdx.y;
    const float* tb = t + (long)b * C * P;
    const float* mb = mean + (long)b * C;
    float* ob = out + (long)b * C * P;
    const bool ok = p < P;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok)
        for (int c = g; c < C; c += 32) {
            const float4 v = *reinterpret_cast<const float4*>(tb + (long)c * P + p);
            const float mu = mb[c];
            const float a0 = v.x - mu, a1 = v.y - mu, a2 = v.z - mu, a3 = v.w - mu;
            s.x = fmaf(a0, a0, s.x); s.y = fmaf(a1, a1, s.y); s.z = fmaf(a2, a2, s.z); s.w = fmaf(a3, a3, s.w);
        }
    part[g][px4] = s;
    __syncthreads();
    float4 tt = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const float4 v = part[k][px4];
        tt.x += v.x; tt.y += v.y; tt.z += v.z; tt.w += v.w;
    }
    const float d0 = sqrtf(tt.x) + eps, d1 = sqrtf(tt.y) + eps, d2 = sqrtf(tt.z) + eps, d3 = sqrtf(tt.w) + eps;
    if (ok)
        for (int c = g; c < C; c += 32) {
            const float4 v = *reinterpret_cast<const float4*>(tb + (long)c * P + p);
            const float mu = mb[c];
            *reinterpret_cast<float4*>(ob + (long)c * P + p) = make_float4((v.x - mu) / d0, (v.y - mu) / d1, (v.z - mu) / d2, (v.w - mu) / d3);
        }
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
    if (P % 4 == 0 && (((reinterpret_cast<uintptr_t>(t_raw) | reinterpret_cast<uintptr_t>(t_out)) & 15) == 0))
        hipLaunchKernelGGL(corr_normalize_v4_kernel, dim3(cdiv(P, 32), B), dim3(256), 0, s, t_raw, mean_scratch, C, P, eps, t_out);
    else
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
    int P, ntiles, nqb, nslot;   // nqb: query blocks per image; nslot: partial-state slots per query (max)
    long U;                      // work units = B * nqb * ntiles, dealt out evenly to gridDim.x workgroups
    long long* dbg;        // debug timeline (NULL in production): [workgroup][tile][4] s_memtime stamps of wave 0
    int dbg_tiles;
    int dbg_variant;       // debug only: 1 = skip the softmax arithmetic (timing experiment, wrong results)
};

// stream-K style decomposition: unit u = (image * nqb + query block) * ntiles + key tile; workgroup w owns the
// contiguous range [w*U/G, (w+1)*U/G) — every workgroup gets the same number of key tiles (+-1), whatever
// P is.  A range spans at most two query blocks (ranges are shorter than ntiles); each (workgroup, query
// block) pair produces one partial state, stored in slot (w - first workgroup of that query block).
__host__ __device__ __forceinline__ long corr_unit_start(long w, long U, long G) { return w * U / G; }
__host__ __device__ __forceinline__ long corr_unit_owner(long u, long U, long G) { return ((u + 1) * G - 1) / U; }

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

// ---- the 128-MFMA chain of one key tile, hand-scheduled.
// The compiler emits `ds_read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs` with a single fragment register pair
// (no scheduler option or sched_group_barrier pattern made it prefetch), which exposes the LDS latency
// 64 times per tile.  Here the A fragments run two MFMA pairs ahead in two register pairs (a0,a1 / b0,b1):
//   wait(pair p landed) ; 2 MFMAs on pair p ; issue the reads of pair p+2 into the same registers.
// LDS returns in order, so `s_waitcnt lgkmcnt(2)` (the two reads of the younger pair may be outstanding)
// is exact.  One asm statement covers 16 MFMAs (operand limit); fragment s lives at byte offset s*256.
#define CORR_STR2(x) #x
#define CORR_STR(x) CORR_STR2(x)
#define CORR_RD(r0, r1, s)                                                      \
    "ds_read_b32 %[" #r0 "], %[addr] offset:(" CORR_STR(s) ")*256\n\t"          \
    "ds_read_b32 %[" #r1 "], %[addr] offset:((" CORR_STR(s) ")+1)*256\n\t"
#define CORR_MM1(r0, r1, q0, q1)                                                \
    "v_mfma_f32_32x32x2_f32 %[acc], %[" #r0 "], %[" #q0 "], %[acc]\n\t"         \
    "v_mfma_f32_32x32x2_f32 %[acc], %[" #r1 "], %[" #q1 "], %[acc]\n\t"
#define CORR_W2 "s_waitcnt lgkmcnt(2)\n\t"
#define CORR_W0 "s_waitcnt lgkmcnt(0)\n\t"
#define CORR_CHAIN_OPS(B)                                                                                   \
    : [acc] "+v"(acc), [a0] "+v"(fa0), [a1] "+v"(fa1), [b0] "+v"(fb0), [b1] "+v"(fb1)                      \
    : [addr] "v"(kaddr), [q0] "v"(qreg[(B) * 16 + 0]), [q1] "v"(qreg[(B) * 16 + 1]),                       \
      [q2] "v"(qreg[(B) * 16 + 2]), [q3] "v"(qreg[(B) * 16 + 3]), [q4] "v"(qreg[(B) * 16 + 4]),            \
      [q5] "v"(qreg[(B) * 16 + 5]), [q6] "v"(qreg[(B) * 16 + 6]), [q7] "v"(qreg[(B) * 16 + 7]),            \
      [q8] "v"(qreg[(B) * 16 + 8]), [q9] "v"(qreg[(B) * 16 + 9]), [q10] "v"(qreg[(B) * 16 + 10]),          \
      [q11] "v"(qreg[(B) * 16 + 11]), [q12] "v"(qreg[(B) * 16 + 12]), [q13] "v"(qreg[(B) * 16 + 13]),      \
      [q14] "v"(qreg[(B) * 16 + 14]), [q15] "v"(qreg[(B) * 16 + 15])                                       \
    : "memory"
// blocks 0..6: every pair refills its registers with the pair four fragments later
#define CORR_CHAIN_BLOCK(MM, B)                                                                             \
    asm volatile(CORR_W2 MM(a0, a1, q0, q1) CORR_RD(a0, a1, (B) * 16 + 4)                                   \
                 CORR_W2 MM(b0, b1, q2, q3) CORR_RD(b0, b1, (B) * 16 + 6)                                   \
                 CORR_W2 MM(a0, a1, q4, q5) CORR_RD(a0, a1, (B) * 16 + 8)                                   \
                 CORR_W2 MM(b0, b1, q6, q7) CORR_RD(b0, b1, (B) * 16 + 10)                                  \
                 CORR_W2 MM(a0, a1, q8, q9) CORR_RD(a0, a1, (B) * 16 + 12)                                  \
                 CORR_W2 MM(b0, b1, q10, q11) CORR_RD(b0, b1, (B) * 16 + 14)                                \
                 CORR_W2 MM(a0, a1, q12, q13) CORR_RD(a0, a1, (B) * 16 + 16)                                \
                 CORR_W2 MM(b0, b1, q14, q15) CORR_RD(b0, b1, (B) * 16 + 18) CORR_CHAIN_OPS(B))
// block 7: no reads past fragment 127; ends with the wait states an MFMA result needs before a VALU read
#define CORR_CHAIN_LAST(MM)                                                                                 \
    asm volatile(CORR_W2 MM(a0, a1, q0, q1) CORR_RD(a0, a1, 7 * 16 + 4)                                     \
                 CORR_W2 MM(b0, b1, q2, q3) CORR_RD(b0, b1, 7 * 16 + 6)                                     \
                 CORR_W2 MM(a0, a1, q4, q5) CORR_RD(a0, a1, 7 * 16 + 8)                                     \
                 CORR_W2 MM(b0, b1, q6, q7) CORR_RD(b0, b1, 7 * 16 + 10)                                    \
                 CORR_W2 MM(a0, a1, q8, q9) CORR_RD(a0, a1, 7 * 16 + 12)                                    \
                 CORR_W2 MM(b0, b1, q10, q11) CORR_RD(b0, b1, 7 * 16 + 14)                                  \
                 CORR_W2 MM(a0, a1, q12, q13)                                                               \
                 CORR_W0 MM(b0, b1, q14, q15) "s_nop 15\n\ts_nop 7\n\t" CORR_CHAIN_OPS(7))
#define CORR_CHAIN_ALL(MM)                                                                                  \
    do {                                                                                                    \
        CORR_CHAIN_BLOCK(MM, 0);                                                                            \
        CORR_CHAIN_BLOCK(MM, 1);                                                                            \
        CORR_CHAIN_BLOCK(MM, 2);                                                                            \
        CORR_CHAIN_BLOCK(MM, 3);                                                                            \
        CORR_CHAIN_BLOCK(MM, 4);                                                                            \
        CORR_CHAIN_BLOCK(MM, 5);                                                                            \
        CORR_CHAIN_BLOCK(MM, 6);                                                                            \
        CORR_CHAIN_LAST(MM);                                                                                \
    } while (0)

// SOFT: the log2-domain organisations of the softmax step (see (4s) below) as their own instantiations — as a run-time
// branch next to the sharp / exact paths they cost the production kernel 35 registers and 20 spills.
//   1: soft temperatures, T >= 1e-3: p = exp2(f * c1 - m * log2 e), one fma + one v_exp_f32 per affinity;
//   2: the middle regime 8.3e-9 <= T < 1e-3, where |f / T| reaches 1e8 and the fma form would cancel catastrophically:
//      p = exp2((f - mf) * c1) with the running maximum mf kept in the AFFINITY domain (the difference of two nearby fp32
//      affinities is exact or rounded at 6e-8 relative), one sub + one mul + one v_exp_f32 per affinity; the partial
//      states then carry mf instead of m = fl32(mf / T) and the merge scales differences by c1 (CorrArgs::mscale).
// DBG: the timeline / timing-experiment hooks (dvc_debug_corr_timeline, dvc_debug_corr_variant) as their own instantiation
// too: their pointers and per-tile tests live in SGPRs the production kernel is short of.
template <bool WTA, bool VEC4, int SOFT = 0, bool DBG = false>
__global__ __launch_bounds__(256, 2) void corr_fwd_kernel(CorrArgs a) {
    // two key tiles (double buffer, 2 x 32 KB) + three pooled-Lab tiles [3][256] (first 96 floats used).
    // Three, because with ONE barrier per iteration the pending tile's Lab (read during the chain by slow
    // waves) and the next tile's Lab (written after the chain by fast waves) must never share a buffer.
    __shared__ __attribute__((aligned(16))) float smem[2 * CORR_C * CORR_KT + 3 * 256];
    float* bl = smem + 2 * CORR_C * CORR_KT;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int P = a.P;
    const long G = gridDim.x;
    // XCD-aware order (speed only): the dispatcher places workgroup b on XCD b % 8, each XCD has its own L2.
    // Give the workgroups of one XCD CONSECUTIVE unit ranges: they then sweep the same few query blocks,
    // i.e. the same key tiles at about the same time, and phi is fetched ~once per XCD instead of once
    // per workgroup.  (bijective for any G: the first G % 8 XCDs get one extra workgroup)
    const long xq = G / 8, xr = G % 8, xcd = blockIdx.x % 8, xi = blockIdx.x / 8;
    const long wlog = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    long u = corr_unit_start(wlog, a.U, G);
    const long u_end = corr_unit_start(wlog + 1, a.U, G);
    long long* dbgh = nullptr;  // debug header slot: [entry, loop start, loop end, exit] of the first segment
    if (DBG && a.dbg && tid == 0) {
        dbgh = a.dbg + ((long)blockIdx.x * a.dbg_tiles + (a.dbg_tiles - 1)) * 4;
        dbgh[0] = __builtin_amdgcn_s_memtime();
    }
  while (u < u_end) {  // one iteration per (query block) segment of this workgroup's unit range: 1 or 2
    const int qbg = (int)(u / a.ntiles);
    const int t0 = (int)(u - (long)qbg * a.ntiles);
    const int t1 = (int)min((long)a.ntiles, t0 + (u_end - u));
    const int b = qbg / a.nqb, qb = qbg - b * a.nqb;
    const int slot0 = (int)(wlog - corr_unit_owner((long)qbg * a.ntiles, a.U, G));
    const int query = qb * CORR_QB + wave * 32 + l31;
    const bool qvalid = query < P;
    const float* th = a.theta + (long)b * CORR_C * P;
    const float* ph = a.phi + (long)b * CORR_C * P;
    const float* blb = a.blab + (long)b * 3 * P;

    // query fragment: B[k = 2s+hi][j = l31] for s = 0..127 (loaded below, after the first key tile's DMA)
    float qreg[CORR_C / 2];

    float m = -INFINITY, l = 0.f, y0 = 0.f, y1 = 0.f, y2 = 0.f, fmax = -INFINITY;
    int amax = 0;
    float fq = 0.f;
    if (WTA) fq = qvalid ? a.fmax_in[(long)b * P + query] : 0.f;

    // ---- key-tile staging into smem[buf][c][0..31] = phi[c][k0..k0+31]
    // VEC4 (P % 4 == 0): LDS-DMA, no VGPRs — a wave instruction moves 8 rows x 128 B (lane -> row
    // lane/8, 16-byte column lane%8); keys beyond P read a valid in-row address (their affinities are
    // masked to -inf later).  Otherwise: scalar loads staged through registers.
    // Everything in the steady-state loop is branch-free (one basic block) so that the scheduler can
    // interleave the matrix and vector streams.
    float blr = 0.f;
    unsigned dma_ofs[8];  // element offset of this lane's 16-byte piece of chunk i (row 8c + lane/8, col 4*(lane%8))
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_ofs[i] = (unsigned)(((i * 4 + wave) * 8 + (lane >> 3)) * P + (lane & 7) * 4);
    float kr[VEC4 ? 1 : (CORR_C * CORR_KT) / 256];
    const int blc = tid < 3 * CORR_KT ? (tid >> 5) : 0, blj = tid & 31;
    auto issue = [&](int t, int buf) {
        const int k0 = t * CORR_KT;
        if (VEC4) {
            float* kb = smem + buf * CORR_C * CORR_KT;
            if (k0 + CORR_KT <= P) {  // full tile: scalar base + loop-invariant lane offset, no address VALU
                const float* src = ph + k0;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_global_load_lds((const AS1 void*)(src + dma_ofs[i]),
                                                     (AS3 void*)(kb + (i * 4 + wave) * 256), 16, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = i * 4 + wave;  // 1 KB chunk = rows 8c .. 8c+7
                    const int row = c * 8 + (lane >> 3), col = (lane & 7) * 4;
                    const float* src = ph + (unsigned)(row * P + (k0 + col < P ? k0 + col : 0));
                    __builtin_amdgcn_global_load_lds((const AS1 void*)src, (AS3 void*)(kb + c * 256), 16, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < (CORR_C * CORR_KT) / 256; ++i) {
                int e = tid + i * 256;
                int row = e >> 5, c = e & 31;
                kr[VEC4 ? 0 : i] = ph[(k0 + c < P) ? (unsigned)(row * P + k0 + c) : 0u];
            }
        }
        blr = blb[(k0 + blj < P) ? (unsigned)(blc * P + k0 + blj) : 0u];  // lanes >= 96 fill unused slots
    };
    auto commit = [&](int buf, int lbuf) {
        if (!VEC4) {
            float* kb = smem + buf * CORR_C * CORR_KT;
#pragma unroll
            for (int i = 0; i < (CORR_C * CORR_KT) / 256; ++i) kb[tid + i * 256] = kr[VEC4 ? 0 : i];
        }
        bl[lbuf * 256 + tid] = blr;
    };

    // ---- online softmax, run on a tile's accumulators right after its MFMA chain.
    // The fp32 MFMA and ordinary fp32 VALU work share the SIMD's fp32 lanes (the vector and matrix fp32
    // peaks are the same number), so softmax instructions can not be hidden under the chain; what can be
    // hidden is latency, and that is the job of the OTHER wave on the SIMD (2 workgroups per CU).  The
    // chain is therefore one straight block of 128 MFMAs, and the softmax a separate block whose cost is
    // kept low by guarding every step with a cheap WAVE-UNIFORM test:
    //   * mf = running maximum in the affinity domain (exact), m = fl32(mf / T) its image (ATen computes
    //     s = fl32(f / T) with a true division and p = exp(s - max s));
    //   * an affinity can have p != 0 only if s - m > -104, which implies f >= thr := mf - (120 T +
    //     4.8e-7 |mf|) (the second term covers the two roundings of s and m); at T = 1e-10 that is ~8 ulp
    //     below the maximum, so all but the row maxima themselves skip the arithmetic;
    //   * the row arg-max (lowest index on ties) is tracked in the same guarded step: the row maximum
    //     always passes the guard.
    // fl32(f / T) for the fixed divisor T: q0 = f*y, two residual corrections with fma (y = fl32(1/T));
    // correctly rounded (Markstein), checked against true division in tests/test_gpu_ops.py.
    float mf = -INFINITY;    // running max affinity (after WTA / masking), exact
    float thr = -INFINITY;
    const float Tn = -a.T, ry = a.invT;
    const bool sharp = !SOFT;   // (the launcher sends 120 T >= 1e-6 to the SOFT instantiations; (4b) is the sharp path's rare fallback)
    auto div_T = [&](float f) {
        float q = f * ry;
        q = fmaf(fmaf(Tn, q, f), ry, q);
        q = fmaf(fmaf(Tn, q, f), ry, q);
        return q;
    };
    auto process_tile = [&](f32x16& sacc, int k0, const float* blp) {
        // (1) WTA / partial last tile only: raw max bookkeeping, then rewrite the affinities in place
        if (WTA || k0 + CORR_KT > P) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float f = sacc[r];
                const bool kvalid = key < P;
                const bool better = kvalid & (f > fmax);  // strict '>' keeps the lowest index on ties
                fmax = better ? f : fmax;
                amax = better ? key : amax;
                if (WTA) f = (f == fq) ? f : f * a.wta_scale;
                sacc[r] = kvalid ? f : -INFINITY;
            }
        }
        // (2) tile maximum (v_max3)
        float t0m = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
        float t1m = fmaxf(fmaxf(sacc[3], sacc[4]), sacc[5]);
        float t2m = fmaxf(fmaxf(sacc[6], sacc[7]), sacc[8]);
        float t3m = fmaxf(fmaxf(sacc[9], sacc[10]), sacc[11]);
        float t4m = fmaxf(fmaxf(sacc[12], sacc[13]), sacc[14]);
        const float tmax = fmaxf(fmaxf(fmaxf(t0m, t1m), fmaxf(t2m, t3m)), fmaxf(t4m, sacc[15]));
        // (3) the tile raises some lane's running max: new image m, rescale the running sums, new guard
        // (lanes of queries beyond P hold all-zero fragments: every affinity ties at 0 — keep them out of the
        // wave-uniform tests, their state is never stored)
        if (__any(qvalid & (tmax > mf))) {
            const float mf_new = fmaxf(mf, tmax);
            // (SOFT 2: the state lives in the affinity domain, m == mf)
            const float m_new = SOFT == 2 ? mf_new : (mf_new == -INFINITY) ? -INFINITY : div_T(mf_new);   // (-inf: all keys masked so far)
            const float sc = (mf == -INFINITY) ? 0.f
                             : SOFT == 2 ? __builtin_amdgcn_exp2f((mf - mf_new) * (ry * 1.44269504088896f)) : __expf(m - m_new);
            l *= sc;
            y0 *= sc;
            y1 *= sc;
            y2 *= sc;
            mf = mf_new;
            m = m_new;
            thr = mf_new - fmaf(4.8e-7f, fabsf(mf_new), 120.f * a.T);
        }
        // (4) affinities that can contribute (or be the arg-max).  __expf (hardware exp2, rel. error ~2e-6
        // for |x| < 100) gives exp(0) == 1 and exp(-big) == 0 exactly — the two cases that decide the
        // T -> 0 regime; at soft temperatures its error is far below the fp32 noise of the affinities / T.
        // (4a) sharp temperatures (guard window below ~1e-6): a lane almost never has more than ONE
        // affinity above the guard in a tile, and that one is its tile maximum.  Count and locate them
        // (3 VALU per affinity, no branches); unless some lane has two, a single update per TILE replaces
        // the per-affinity steps of (4b) — the terms it skips are exact zeros, so the sums are bit-identical.
        if (sharp) {
            // (count first, locate afterwards by equality with the tile maximum — the one candidate of a lane IS its tile
            // maximum: keeping the 16 compare masks alive for a fused count-and-locate loop costs 32 SGPRs and sends the
            // kernel's scalar state through v_writelane / v_readlane spills)
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) cnt += (sacc[r] >= thr) ? 1 : 0;
            cnt = qvalid ? cnt : 0;
            if (!__any(cnt > 1)) {
                if (__any(cnt == 1)) {
                    const bool has = cnt == 1;   // (tmax is then finite: a masked -inf never passes a finite guard)
                    int idx = 15;
#pragma unroll
                    for (int r = 14; r >= 0; --r) idx = (sacc[r] == tmax) ? r : idx;
                    const int kl = (idx & 3) + 8 * (idx >> 2) + 4 * hi;
                    // (a candidate that IS the lane's running maximum — the usual case: its max has just risen, or ties an
                    // earlier tile's — has s == m and p = exp(0) = 1 exactly; the division / exponential only run when some
                    // lane's candidate sits below its maximum inside the guard window)
                    float pe = has ? 1.f : 0.f;
                    if (__any(has & (tmax != mf))) pe = has ? __expf(div_T(tmax) - m) : 0.f;
                    l += pe;
                    y0 = fmaf(pe, blp[kl], y0);
                    y1 = fmaf(pe, blp[CORR_KT + kl], y1);
                    y2 = fmaf(pe, blp[2 * CORR_KT + kl], y2);
                    if (!WTA) {
                        const bool better = has & (tmax > fmax);
                        fmax = better ? tmax : fmax;
                        amax = better ? k0 + kl : amax;
                    }
                }
                return;
            }
        }
        // (4s) soft temperatures, T >= 1e-3 (the training-side values 0.01 / 0.005): every affinity contributes, and
        // |f / T| <= 1000, where one fp32 ulp of s = fl32(f / T) is already 6e-5 — emulating the exact division (step 4b)
        // buys nothing there.  p = exp(f / T - m) as ONE fma + ONE v_exp_f32 per affinity (log2 domain), no guards (a
        // masked -inf gives exactly 0), and the arg-max bookkeeping once per tile instead of once per affinity:
        // 6 instead of ~18 VALU per affinity, the same cost class as the sharp path.
        // (4m) the middle regime (SOFT == 2): the same step with p = exp2((f - mf) * c1).  The reference evaluates exp(s - max s)
        // on s = fl32(f / T); here s is never formed — the two differ by the rounding of s, 6e-8 |f| in units of f, below the
        // 3e-7 rounding of the affinities themselves (tests/test_gpu_ops.py: first-order bound, T = 1e-8 ... 9e-4).
        if (SOFT) {
            const float c1 = ry * 1.44269504088896f;
            const float nm2 = (mf == -INFINITY) ? 0.f : -m * 1.44269504088896f;
            const float base = (mf == -INFINITY) ? 0.f : mf;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float pe = __builtin_amdgcn_exp2f(SOFT == 2 ? (sacc[r] - base) * c1 : fmaf(sacc[r], c1, nm2));
                l += pe;
                y0 = fmaf(pe, blp[kl], y0);
                y1 = fmaf(pe, blp[CORR_KT + kl], y1);
                y2 = fmaf(pe, blp[2 * CORR_KT + kl], y2);
            }
            if (!WTA) {
                const bool better = tmax > fmax;             // strict: an earlier tile keeps the arg-max on ties
                if (__any(qvalid & better)) {
                    int idx = 15;
#pragma unroll
                    for (int r = 14; r >= 0; --r) idx = (sacc[r] == tmax) ? r : idx;     // lowest key of the lane on ties
                    fmax = better ? tmax : fmax;
                    amax = better ? k0 + (idx & 3) + 8 * (idx >> 2) + 4 * hi : amax;
                }
            }
            return;
        }
        // (4b) general case
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float f = sacc[r];
            const bool cand = qvalid & (f >= thr);   // masked keys are -inf; thr is -inf only while mf is
            if (__any(cand)) {
                const int kl = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float pe = (cand && f > -INFINITY) ? __expf(div_T(f) - m) : 0.f;
                l += pe;
                y0 = fmaf(pe, blp[kl], y0);
                y1 = fmaf(pe, blp[CORR_KT + kl], y1);
                y2 = fmaf(pe, blp[2 * CORR_KT + kl], y2);
                if (!WTA) {
                    const bool better = f > fmax;
                    fmax = better ? f : fmax;
                    amax = better ? k0 + kl : amax;
                }
            }
        }
    };

    // Prologue order: first key tile's DMA, then the 128 theta loads (unconditional, from a clamped
    // position: 128 independent loads in flight; lanes of queries beyond P compute on a duplicate of
    // query 0 and are discarded), then wait for the DMA only.  VMEM returns in order and s_waitcnt can
    // express at most vmcnt(63): "at most 63 outstanding" covers the DMA plus the first 65 theta loads, so
    // the chain starts on its first fragments while the rest are still in flight (the compiler waits
    // for each asm statement's operands).
    if (t0 < t1) issue(t0, 0);
    {
        const float* tq = th + (unsigned)(hi * P + (qvalid ? query : 0));
#pragma unroll
        for (int s = 0; s < CORR_C / 2; ++s) qreg[s] = tq[(unsigned)(2 * s * P)];
    }
    if (t0 < t1) commit(0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // vmcnt(63) expcnt(7) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    if (DBG && dbgh) dbgh[1] = __builtin_amdgcn_s_memtime();
    long long* dbgp = nullptr;
    if (DBG && a.dbg && tid == 0)
        dbgp = a.dbg + (long)blockIdx.x * a.dbg_tiles * 4;
    for (int t = t0; t < t1; ++t) {
        const int cur = (t - t0) & 1;
        if (DBG && dbgp && t - t0 < a.dbg_tiles - 1) dbgp[(t - t0) * 4 + 0] = __builtin_amdgcn_s_memtime();
        issue(min(t + 1, t1 - 1), cur ^ 1);  // (the last iteration re-stages its own tile: harmless)

        // S^T tile of THIS key tile: 128 dependent MFMAs (K = 256), one basic block; the A fragments are
        // conflict-free ds_reads the scheduler is free to hoist
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // LDS byte address of this lane's fragment 0: A[i = l31][k = hi] of the tile in buffer `cur`
        const unsigned kaddr = (unsigned)(size_t)(AS3 float*)(smem + cur * CORR_C * CORR_KT + hi * CORR_KT + l31);
        float fa0, fa1, fb0, fb1;
        asm volatile(CORR_RD(a0, a1, 0) CORR_RD(b0, b1, 2)
                     : [a0] "=&v"(fa0), [a1] "=&v"(fa1), [b0] "=&v"(fb0), [b1] "=&v"(fb1)
                     : [addr] "v"(kaddr)
                     : "memory");
        CORR_CHAIN_ALL(CORR_MM1);
        if (DBG && dbgp && t - t0 < a.dbg_tiles - 1) dbgp[(t - t0) * 4 + 1] = __builtin_amdgcn_s_memtime();
        if (!DBG || a.dbg_variant != 1) process_tile(acc, t * CORR_KT, bl + ((t - t0) % 3) * 256);
        if (DBG && dbgp && t - t0 < a.dbg_tiles - 1) dbgp[(t - t0) * 4 + 2] = __builtin_amdgcn_s_memtime();
        commit(cur ^ 1, (t + 1 - t0) % 3);
        __syncthreads();  // next tile landed (DMA drained / stores visible); this tile's reads done
        if (DBG && dbgp && t - t0 < a.dbg_tiles - 1) dbgp[(t - t0) * 4 + 3] = __builtin_amdgcn_s_memtime();
    }
    if (DBG && dbgh) dbgh[2] = __builtin_amdgcn_s_memtime();

    // ---- combine the two key halves of the wave (lanes l and l^32 hold different keys of the SAME query) and write
    // the partial state of this (workgroup, query block) pair: slot = workgroup - first workgroup of the query block.
    // The combination is evaluated by the lower lane as f(state of lanes 0..31, state of lanes 32..63): fixed order.
    {
        const float m_o = __shfl_xor(m, 32), l_o = __shfl_xor(l, 32), y0_o = __shfl_xor(y0, 32), y1_o = __shfl_xor(y1, 32),
                    y2_o = __shfl_xor(y2, 32), f_o = __shfl_xor(fmax, 32);
        const int a_o = __shfl_xor(amax, 32);
        const float M = fmaxf(m, m_o);
        const float c1h = a.invT * 1.44269504088896f;
        const float s_a = (m == -INFINITY) ? 0.f : SOFT == 2 ? __builtin_amdgcn_exp2f((m - M) * c1h) : expf(m - M);
        const float s_b = (m_o == -INFINITY) ? 0.f : SOFT == 2 ? __builtin_amdgcn_exp2f((m_o - M) * c1h) : expf(m_o - M);
        l = fmaf(l_o, s_b, l * s_a);
        y0 = fmaf(y0_o, s_b, y0 * s_a);
        y1 = fmaf(y1_o, s_b, y1 * s_a);
        y2 = fmaf(y2_o, s_b, y2 * s_a);
        m = M;
        const bool better = f_o > fmax || (f_o == fmax && a_o < amax);
        fmax = better ? f_o : fmax;
        amax = better ? a_o : amax;
    }
    if (qvalid && hi == 0) {
        float* pp = a.part + (((long)b * a.nslot + slot0) * CORR_NF) * P + query;
        pp[0] = m;
        pp[(long)P] = l;
        pp[2L * P] = y0;
        pp[3L * P] = y1;
        pp[4L * P] = y2;
        pp[5L * P] = fmax;
        pp[6L * P] = __int_as_float(amax);
    }
    if (DBG && dbgh) {
        dbgh[3] = __builtin_amdgcn_s_memtime();
        dbgh = nullptr;
    }
    u += t1 - t0;
    __syncthreads();  // LDS buffers are reused by the next segment
  }
}

// merge the partial states of each query; write small + x4-upsampled outputs.
// Workgroup = 32 queries x 8 slot groups (this kernel is pure L2 latency: short slot loops, many loads in flight);
// the groups are combined through LDS in a fixed order (deterministic).
#define CORR_MQ 32
#define CORR_MG 8
// The consumer's form (dvc_corr_merge_pack): `out7` != NULL makes this launch ALSO the cat() of models/FrameColor.py:63-64 — the
// merged warped ab and similarity go straight into channels 1..3 of ColorVidNet's 7-channel input (x4 nearest), and the
// workgroup copies the 4x4 pixel blocks of its 32 queries of the four planes that are pure data movement (current L,
// previous L, previous ab) — so neither the warped Lab / similarity maps nor a separate pack launch exist.
struct CorrPackArgs {
    float* out7;             // [7][16 P] or NULL
    const float* IA_l;       // [16 P]
    const float* last_l;     // [16 P]
    const float* last_ab;    // [2][16 P]
};
__global__ __launch_bounds__(256) void corr_merge_kernel(const float* __restrict__ part, int nslot, int P,
                                                         int nqb, int ntiles, long U, long G, float mscale,
                                                         int h, int w, float* __restrict__ y_small,
                                                         float* __restrict__ sim_small,
                                                         float* __restrict__ y_up,
                                                         float* __restrict__ sim_up,
                                                         int* __restrict__ argmax, CorrPackArgs pk) {
    __shared__ float sh[CORR_MG][7][CORR_MQ];
    const int qx_ = threadIdx.x & (CORR_MQ - 1), g = threadIdx.x / CORR_MQ;
    const int q = blockIdx.x * CORR_MQ + qx_;
    const int b = blockIdx.y;
    const bool ok = q < P;
    const float* pb = part + (long)b * nslot * CORR_NF * P + (ok ? q : 0);
    // the partial states of this query block were written by workgroups w_lo..w_hi (one slot each)
    const long qbg = (long)b * nqb + (blockIdx.x * CORR_MQ) / CORR_QB;   // CORR_MQ | CORR_QB: one query block per workgroup
    const int nused = (int)(corr_unit_owner((qbg + 1) * ntiles - 1, U, G) - corr_unit_owner(qbg * ntiles, U, G) + 1);
    // pass 1 (this group's slots): running max of m, best (fmax, argmax)
    float M = -INFINITY, F = -INFINITY;
    int A = 0x7fffffff;
    for (int s = g; s < nused; s += CORR_MG) {
        const float* ps = pb + (long)s * CORR_NF * P;
        M = fmaxf(M, ps[0]);
        float f = ps[5L * P];
        int ai = __float_as_int(ps[6L * P]);
        bool better = f > F || (f == F && ai < A);
        F = better ? f : F;
        A = better ? ai : A;
    }
    sh[g][0][qx_] = M;
    sh[g][5][qx_] = F;
    sh[g][6][qx_] = __int_as_float(A);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CORR_MG; ++k) M = fmaxf(M, sh[k][0][qx_]);
    // pass 2: rescaled sums of this group's slots
    float L = 0.f, Y0 = 0.f, Y1 = 0.f, Y2 = 0.f;
    for (int s = g; s < nused; s += CORR_MG) {
        const float* ps = pb + (long)s * CORR_NF * P;
        float ms = ps[0];
        // (mscale != 0: the states come from the middle-regime instantiation and carry the running maximum in the affinity
        // domain; differences are scaled by log2(e) / T)
        float sc = (ms == -INFINITY) ? 0.f : mscale != 0.f ? __builtin_amdgcn_exp2f((ms - M) * mscale) : expf(ms - M);
        L = fmaf(ps[(long)P], sc, L);
        Y0 = fmaf(ps[2L * P], sc, Y0);
        Y1 = fmaf(ps[3L * P], sc, Y1);
        Y2 = fmaf(ps[4L * P], sc, Y2);
    }
    sh[g][1][qx_] = L;
    sh[g][2][qx_] = Y0;
    sh[g][3][qx_] = Y1;
    sh[g][4][qx_] = Y2;
    __syncthreads();
    const long W4 = 4L * w, HW16 = 16L * P;
    if (pk.out7) {
        // the four copied planes: 4 planes x 32 queries x 4 rows = 512 float4 pieces, two per thread
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = threadIdx.x + it * 256;
            const int plane = item >> 7, qi = (item >> 2) & 31, dy = item & 3;
            const int qq = blockIdx.x * CORR_MQ + qi;
            if (qq < P) {
                const int qy_ = qq / w, qx2 = qq - qy_ * w;
                const long off = (4L * qy_ + dy) * W4 + 4L * qx2;
                const float* src = plane == 0 ? pk.IA_l : plane == 1 ? pk.last_l : pk.last_ab + (plane - 2) * HW16;
                float* dst = pk.out7 + (plane == 0 ? 0L : (long)(plane + 3) * HW16);
                *reinterpret_cast<float4*>(dst + off) = *reinterpret_cast<const float4*>(src + off);
            }
        }
    }
    if (g != 0 || !ok) return;
    L = Y0 = Y1 = Y2 = 0.f;
#pragma unroll
    for (int k = 0; k < CORR_MG; ++k) {   // fixed order
        L += sh[k][1][qx_];
        Y0 += sh[k][2][qx_];
        Y1 += sh[k][3][qx_];
        Y2 += sh[k][4][qx_];
    }
    F = sh[0][5][qx_];
    A = __float_as_int(sh[0][6][qx_]);
    for (int k = 1; k < CORR_MG; ++k) {
        float f = sh[k][5][qx_];
        int ai = __float_as_int(sh[k][6][qx_]);
        bool better = f > F || (f == F && ai < A);
        F = better ? f : F;
        A = better ? ai : A;
    }
    const float yv[3] = {Y0 / L, Y1 / L, Y2 / L};
    if (y_small)
        for (int c = 0; c < 3; ++c) y_small[((long)b * 3 + c) * P + q] = yv[c];
    if (sim_small) sim_small[(long)b * P + q] = F;
    if (argmax) argmax[(long)b * P + q] = A;
    const int qy = q / w, qx = q - qy * w;
    if (pk.out7) {
        const float v3[3] = {yv[1], yv[2], F};
        for (int c = 0; c < 3; ++c) {
            float4 v = make_float4(v3[c], v3[c], v3[c], v3[c]);
            float* o = pk.out7 + (long)(c + 1) * HW16 + (4L * qy) * W4 + 4L * qx;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) *reinterpret_cast<float4*>(o + dy * W4) = v;
        }
    }
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

// decomposition parameters shared by the launch and the workspace query
struct CorrPlan {
    int ntiles, nqb, nslot;
    long U, G;
};
static CorrPlan corr_plan(int B, int P) {
    CorrPlan p;
    p.ntiles = cdiv(P, CORR_KT);
    p.nqb = cdiv(P, CORR_QB);
    p.U = (long)B * p.nqb * p.ntiles;
    p.G = p.U < 512 ? p.U : 512;  // 2 resident workgroups per CU x 256 CUs
    // at most ceil(ntiles / floor(U/G)) + 1 workgroups touch one query block
    long per = p.U / p.G;
    if (per < 1) per = 1;
    p.nslot = (int)((p.ntiles + per - 1) / per + 1);
    return p;
}

extern "C" size_t dvc_corr_workspace_bytes(int32_t B, int32_t P) {
    if (B <= 0 || P <= 0) return 0;
    CorrPlan p = corr_plan(B, P);
    size_t part = (size_t)B * p.nslot * CORR_NF * P * sizeof(float);
    size_t fmax = (size_t)B * P * sizeof(float);
    return part + fmax + 256;
}

// debug hook (not part of the public header): the next dvc_corr_fwd launches record per-tile s_memtime
// stamps of wave 0 of every workgroup into `buf` ([workgroups][max_tiles][4]); pass NULL to switch off.
#ifdef DVC_DEBUG
static long long* g_corr_dbg = nullptr;
static int g_corr_dbg_tiles = 0;
static int g_corr_dbg_variant = 0;
long long* g_corr_bf16_dbg = nullptr;     // (max_tiles < 0: the buffer is for the bf16 pass kernels instead, [2][256] stamps)
extern "C" void dvc_debug_corr_timeline(long long* buf, int max_tiles) {
    if (max_tiles < 0) {
        g_corr_bf16_dbg = buf;
        return;
    }
    g_corr_dbg = buf;
    g_corr_dbg_tiles = max_tiles;
}
extern "C" void dvc_debug_corr_variant(int v) { g_corr_dbg_variant = v; }
#else
static constexpr long long* g_corr_dbg = nullptr;
static constexpr int g_corr_dbg_tiles = 0, g_corr_dbg_variant = 0;
#endif

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
    // every output NULL = the merge is left to dvc_corr_merge_pack, which reads ONE image's single-pass partial states out of
    // `workspace`: a batch would overwrite them image by image, and a WTA call leaves its second pass's states unmerged
    if (!y_small && !sim_small && !y_up && !sim_up && !argmax)
        DVC_REQUIRE(B == 1 && wta_scale == 1.0f,
                    "dvc_corr_fwd: a deferred merge (every output NULL) needs B == 1 and wta_scale == 1 (got B = %d, wta_scale = %g)",
                    B, (double)wta_scale);
    // One image per set of launches, each with the single-image decomposition: an image's result does not depend on the
    // batch it came in (the stream-K unit ranges, hence the order in which partial softmax states are merged, would
    // otherwise change with B).  The launches are 0.13 ms each; nothing is lost.
    if (B > 1) {
        const long PC = (long)CORR_C * P, P16 = 16L * P;
        for (int b = 0; b < B; ++b) {
            const int rc = dvc_corr_fwd(theta + b * PC, phi + b * PC, blab + (long)b * 3 * P, temperature, wta_scale, 1, C, h, w,
                                        y_small ? y_small + (long)b * 3 * P : nullptr, sim_small ? sim_small + (long)b * P : nullptr,
                                        y_up ? y_up + b * 3 * P16 : nullptr, sim_up ? sim_up + b * P16 : nullptr,
                                        argmax ? argmax + (long)b * P : nullptr, workspace, workspace_bytes, stream);
            if (rc != 0) return rc;
        }
        return 0;
    }
    CorrArgs a;
    a.dbg = g_corr_dbg; a.dbg_tiles = g_corr_dbg_tiles; a.dbg_variant = g_corr_dbg_variant;
    a.theta = theta; a.phi = phi; a.blab = blab;
    a.T = temperature; a.invT = 1.0f / temperature; a.wta_scale = wta_scale; a.P = P;
    const CorrPlan pl = corr_plan(B, P);
    a.ntiles = pl.ntiles; a.nqb = pl.nqb; a.nslot = pl.nslot; a.U = pl.U;
    a.part = reinterpret_cast<float*>(workspace);
    float* fmax_buf = a.part + (size_t)B * a.nslot * CORR_NF * P;
    a.fmax_in = fmax_buf;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)pl.G);
    const bool vec4 = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(phi) & 15) == 0);
    dim3 mgrid(cdiv(P, CORR_MQ), B);
    static_assert(CORR_QB % CORR_MQ == 0, "merge kernel assumes one query block per workgroup");
    const bool wta = wta_scale != 1.0f;
    // softmax organisation: 0 = sharp (guard window 120 T < 1e-6: one update per tile, exact division on the rare fallback),
    // 1 = soft (T >= 1e-3, the training-side temperatures), 2 = the middle regime in between (log2 domain around the running
    // maximum kept in the affinity domain)
    const int mode = temperature >= 1e-3f ? 1 : (120.f * temperature >= 1e-6f ? 2 : 0);
    const float mscale = mode == 2 ? a.invT * 1.44269504088896f : 0.f;
    auto launch = [&](bool wta_pass) {
#define CORR_LAUNCH(W, V)                                                                                                  \
        do {                                                                                                               \
            if (mode == 1) hipLaunchKernelGGL((corr_fwd_kernel<W, V, 1>), grid, dim3(256), 0, s, a);                       \
            else if (mode == 2) hipLaunchKernelGGL((corr_fwd_kernel<W, V, 2>), grid, dim3(256), 0, s, a);                  \
            else hipLaunchKernelGGL((corr_fwd_kernel<W, V, 0>), grid, dim3(256), 0, s, a);                                 \
        } while (0)
        if (wta_pass) {
            if (vec4) CORR_LAUNCH(true, true);
            else CORR_LAUNCH(true, false);
        } else {
#ifdef DVC_DEBUG
            if (vec4 && mode == 0 && (a.dbg || a.dbg_variant)) {
                hipLaunchKernelGGL((corr_fwd_kernel<false, true, 0, true>), grid, dim3(256), 0, s, a);
                return;
            }
#endif
            if (vec4) CORR_LAUNCH(false, true);
            else CORR_LAUNCH(false, false);
        }
#undef CORR_LAUNCH
    };
    if (wta) {
        // pass 1: row maxima only (identical MFMA order => `f == rowmax` is exact in pass 2)
        launch(false);
        DVC_CHECK_LAUNCH("dvc_corr_fwd(pass1)");
        hipLaunchKernelGGL(corr_merge_kernel, mgrid, dim3(256), 0, s, a.part, a.nslot, P, pl.nqb, pl.ntiles, pl.U,
                           pl.G, mscale, h, w, (float*)nullptr, fmax_buf, (float*)nullptr, (float*)nullptr, (int*)nullptr,
                           CorrPackArgs{nullptr, nullptr, nullptr, nullptr});
        DVC_CHECK_LAUNCH("dvc_corr_fwd(merge1)");
        launch(true);
    } else {
        launch(false);
    }
    DVC_CHECK_LAUNCH("dvc_corr_fwd");
    // every output NULL: the merge is left to the consumer (dvc_corr_merge_pack) — the partial states stay in `workspace`
    if (!y_small && !sim_small && !y_up && !sim_up && !argmax) return 0;
    hipLaunchKernelGGL(corr_merge_kernel, mgrid, dim3(256), 0, s, a.part, a.nslot, P, pl.nqb, pl.ntiles, pl.U, pl.G,
                       mscale, h, w, y_small, sim_small, y_up, sim_up, argmax, CorrPackArgs{nullptr, nullptr, nullptr, nullptr});
    DVC_CHECK_LAUNCH("dvc_corr_fwd(merge)");
    return 0;
}

extern "C" int dvc_corr_merge_pack(const void* workspace, size_t workspace_bytes, float temperature, int32_t h, int32_t w,
                                   const float* IA_l, const float* last_l, const float* last_ab, float* out7, float* y_up,
                                   float* sim_up, dvcStream stream) {
    DVC_REQUIRE(workspace && IA_l && last_l && last_ab && out7, "dvc_corr_merge_pack: null argument");
    DVC_REQUIRE(h > 0 && w > 0 && temperature > 0.f && std::isfinite(temperature), "dvc_corr_merge_pack: bad shape / temperature");
    const int P = h * w;
    DVC_REQUIRE(workspace_bytes >= dvc_corr_workspace_bytes(1, P), "dvc_corr_merge_pack: workspace too small");
    DVC_REQUIRE(((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(IA_l) | reinterpret_cast<uintptr_t>(last_l) |
                  reinterpret_cast<uintptr_t>(last_ab) | reinterpret_cast<uintptr_t>(out7) | reinterpret_cast<uintptr_t>(y_up) |
                  reinterpret_cast<uintptr_t>(sim_up)) & 15) == 0,
                "dvc_corr_merge_pack: every plane must be 16-byte aligned");
    const CorrPlan pl = corr_plan(1, P);
    const int mode = temperature >= 1e-3f ? 1 : (120.f * temperature >= 1e-6f ? 2 : 0);       // as dvc_corr_fwd
    const float mscale = mode == 2 ? (1.0f / temperature) * 1.44269504088896f : 0.f;
    hipLaunchKernelGGL(corr_merge_kernel, dim3(cdiv(P, CORR_MQ), 1), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(workspace), pl.nslot, P, pl.nqb, pl.ntiles, pl.U, pl.G, mscale, h, w,
                       (float*)nullptr, (float*)nullptr, y_up, sim_up, (int*)nullptr, CorrPackArgs{out7, IA_l, last_l, last_ab});
    DVC_CHECK_LAUNCH("dvc_corr_merge_pack");
    return 0;
}
