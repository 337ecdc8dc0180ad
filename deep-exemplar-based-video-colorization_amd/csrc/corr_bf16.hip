// bf16 mixed-precision correlation (BASELINE.json configs[4]): bf16 MFMA affinities as a CANDIDATE
// FILTER, exact fp32 re-scoring of the candidates.
//
// theta/phi columns are unit vectors (centred + L2-normalised, NonlocalNet.py:469-476), so rounding them
// to bf16 (relative error <= 2^-9 each) perturbs an affinity by at most
//     |f_bf16 - f| <= (2*2^-9 + 2^-18) * sum_c |theta_c||phi_c| <= 2^-8 = 3.9e-3      (Cauchy-Schwarz)
// (the MFMA accumulates in fp32).  Hence the true fp32 row maximum is always within 2*2^-8 of the bf16
// row maximum M: every key with f_bf16 >= M - DELTA (DELTA = 7.9e-3) is a candidate, the true argmax is
// guaranteed to be among them, and the exact fp32 affinities of the candidates reproduce the fp32 path's
// argmax / similarity / one-hot colour.  Keys outside the candidate set have f <= max - 4e-3, so their
// softmax weight is <= exp(-4e-3 / T): negligible (< 4e-18) for T <= 1e-4 — the regime test.py:94 uses
// (T = 1e-10).  For larger temperatures the host falls back to the fp32 kernel.
//
//   pass 1  corr_bf16_kernel<1>   v_mfma_f32_32x32x16_bf16, per-lane running max of its keys (partial maxima per key split)
//   pass 2  corr_bf16_kernel<2>   same MFMAs again (they are ~16x cheaper than fp32 ones); every wave first takes its
//                                 queries' row maxima from pass 1's partial maxima, then keys within DELTA of the row
//                                 maximum are appended to a per-query candidate list
//   (both passes: a ring of four 16 KB key tiles filled by LDS-DMA three tiles ahead, hand-placed vmcnt / lgkmcnt waits —
//   at 16 MFMAs of 32 cycles per tile the kernel is bound by the L2 / fabric latency and bandwidth of the key stream:
//   256 workgroups of 256 queries x (13.5 key tiles x 16 KB + 128 KB of theta) = 87 MB per pass at P = 5184)
//   rescore corr_bf16_rescore_kernel  one wave per query: exact fp32 dot products of the (sorted)
//                                 candidates, then the reference's max / softmax(f/T) / colour gather
//                                 restricted to them; a query whose list overflowed is re-scored
//                                 against ALL keys (exact, just slower).
// Layouts: bf16 and fp32 copies of theta/phi as [P][C] (channel-contiguous): an MFMA fragment is one
// 16-byte load, a candidate's fp32 column is one contiguous 1 KB row.
#include "common.h"

#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CB_C 256
#define CB_KT 32          // keys per LDS tile (32 x 512 B = 16 KB)
#define CB_QB 256         // queries per workgroup: 8 waves x 32 (one workgroup per CU, two waves per SIMD); a key tile
                          // fetched once serves 256 queries — the passes are bound by the key stream, not the matrix pipe
#define CB_CAP 64         // candidate list capacity per query (one per lane of the re-scoring wave)
#define CB_DELTA 7.9e-3f  // 2 * 2^-8 + margin

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);   // inputs are finite, |f| <= 1
    return (unsigned short)(u >> 16);
}

// ------------------------------------------------------------------------------------------------
// centre + normalise (as corr_prepare) but emit [P][C] fp32 and [P][C] bf16
__global__ __launch_bounds__(256) void corr_prepare_t_kernel(const float* __restrict__ t,
                                                             const float* __restrict__ mean, int P, float eps,
                                                             float* __restrict__ out_f32,
                                                             unsigned short* __restrict__ out_bf16) {
    __shared__ float tile[CB_C][33];
    __shared__ float part[8][32];
    const int px = threadIdx.x & 31, g = threadIdx.x >> 5;  // 32 positions x 8 channel groups
    const int p0 = blockIdx.x * 32;
    const int b = blockIdx.y;
    const float* tb = t + (long)b * CB_C * P;
    const float* mb = mean + (long)b * CB_C;
    const int p = p0 + px;
    const bool ok = p < P;
    float s = 0.f;
    for (int c = g; c < CB_C; c += 8) {
        float v = ok ? tb[(long)c * P + p] - mb[c] : 0.f;
        tile[c][px] = v;
        s = fmaf(v, v, s);
    }
    part[g][px] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += part[k][px];
    const float den = sqrtf(tot) + eps;
    for (int c = g; c < CB_C; c += 8) tile[c][px] = tile[c][px] / den;
    __syncthreads();
    // transpose out: thread -> position tid>>3, 32 consecutive channels starting at (tid&7)*32
    const int pp = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 32;
    if (p0 + pp < P) {
        float* of = out_f32 + ((long)b * P + p0 + pp) * CB_C + c0;
        unsigned short* ob = out_bf16 + ((long)b * P + p0 + pp) * CB_C + c0;
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
            float4 v = make_float4(tile[c0 + k][pp], tile[c0 + k + 1][pp], tile[c0 + k + 2][pp], tile[c0 + k + 3][pp]);
            *reinterpret_cast<float4*>(of + k) = v;
            ushort4 h;
            h.x = f32_to_bf16_rne(v.x); h.y = f32_to_bf16_rne(v.y); h.z = f32_to_bf16_rne(v.z); h.w = f32_to_bf16_rne(v.w);
            *reinterpret_cast<ushort4*>(ob + k) = h;
        }
    }
}

__global__ __launch_bounds__(256) void corr_rowmean2_kernel(const float* __restrict__ t, int P,
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

extern "C" int dvc_corr_prepare_bf16(const float* t_raw, int32_t B, int32_t C, int32_t P, float eps,
                                     float* mean_scratch, float* t_f32_pc, void* t_bf16_pc, dvcStream stream) {
    DVC_REQUIRE(t_raw && mean_scratch && t_f32_pc && t_bf16_pc && B > 0 && P > 0, "dvc_corr_prepare_bf16: bad argument");
    DVC_REQUIRE(C == CB_C, "dvc_corr_prepare_bf16: C must be %d", CB_C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(corr_rowmean2_kernel, dim3(B * C), dim3(256), 0, s, t_raw, P, mean_scratch);
    DVC_CHECK_LAUNCH("dvc_corr_prepare_bf16(mean)");
    hipLaunchKernelGGL(corr_prepare_t_kernel, dim3(cdiv(P, 32), B), dim3(256), 0, s, t_raw, mean_scratch, P, eps,
                       t_f32_pc, reinterpret_cast<unsigned short*>(t_bf16_pc));
    DVC_CHECK_LAUNCH("dvc_corr_prepare_bf16(normalise)");
    return 0;
}

// ------------------------------------------------------------------------------------------------
struct CorrBf16Args {
    const unsigned short* theta;  // [B][P][C] bf16
    const unsigned short* phi;    // [B][P][C] bf16
    float* part_max;              // pass 1 out: [B][nslot][P]
    int* cnt;                     // pass 2 out: [B][P]
    int* list;                    // pass 2 out: [B][P][CB_CAP]
    int P, ntiles, tiles_per_split, nslot;
    long long* dbg;               // -DDVC_DEBUG only: s_memtime stamps of wave 0 of workgroup 0 (dvc_debug_corr_timeline)
};

// LDS fragment reads as inline assembly: a ds_read the compiler can see makes it drain the LDS-DMA queue first
// (`s_waitcnt vmcnt(0)`: the reads may alias the DMA destinations), which would serialise the key-tile prefetch with the
// arithmetic — with 16 MFMAs of 32 cycles per tile against ~2000 cycles of L2 / fabric latency per tile fetch, the r02 form
// of this kernel (one tile in flight, compiler-visible reads) spent 8x the matrix time per tile.  The waits are placed by hand.
#define CB_RD8(F, A, S0)                                                                                                 \
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"           \
                 "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"               \
                 : "=&v"(F[S0 + 0]), "=&v"(F[S0 + 1]), "=&v"(F[S0 + 2]), "=&v"(F[S0 + 3]), "=&v"(F[S0 + 4]), "=&v"(F[S0 + 5]),    \
                   "=&v"(F[S0 + 6]), "=&v"(F[S0 + 7])                                                                     \
                 : "v"(A[S0 + 0]), "v"(A[S0 + 1]), "v"(A[S0 + 2]), "v"(A[S0 + 3]), "v"(A[S0 + 4]), "v"(A[S0 + 5]),        \
                   "v"(A[S0 + 6]), "v"(A[S0 + 7])                                                                         \
                 : "memory")
#define CB_WAIT8(F, S0, N)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                              \
                 : "+v"(F[S0 + 0]), "+v"(F[S0 + 1]), "+v"(F[S0 + 2]), "+v"(F[S0 + 3]), "+v"(F[S0 + 4]), "+v"(F[S0 + 5]),  \
                   "+v"(F[S0 + 6]), "+v"(F[S0 + 7])                                                                       \
                 :                                                                                                        \
                 : "memory")

#define CB_NBUF 4         // key tiles in the LDS ring: tile t is consumed while t+1 .. t+3 are in flight

template <int PASS>
__global__ __launch_bounds__(512, 2) void corr_bf16_kernel(CorrBf16Args a) {
    // ring of four key tiles [32 keys][32 x 16 B], 16-byte columns XOR-swizzled with (key & 15) so that the per-lane
    // ds_read_b128 of column 2s+hi over 32 different keys is bank-conflict free; eight such slots (128 KB: one workgroup per
    // CU) so that the prologue can stage every wave's theta block in its own slot
    __shared__ __attribute__((aligned(16))) unsigned short smem[8 * CB_KT * CB_C];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, split = blockIdx.y;
    const int P = a.P;
    const int query = blockIdx.x * CB_QB + wave * 32 + l31;
    const bool qvalid = query < P;
    const unsigned short* th = a.theta + (long)b * P * CB_C;
    const unsigned short* ph = a.phi + (long)b * P * CB_C;
    const int t0 = split * a.tiles_per_split;
    const int t1 = min(a.ntiles, t0 + a.tiles_per_split);
    long long* dbgp = nullptr;
    int dbgi = 0;
    if (kDvcDebug && a.dbg && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) dbgp = a.dbg + (PASS - 1) * 256;
#define CB_STAMP() do { if (kDvcDebug && dbgp && dbgi < 255) dbgp[dbgi++] = __builtin_amdgcn_s_memtime(); } while (0)
    CB_STAMP();

    auto issue = [&](int t, int buf) {  // LDS-DMA: a wave instruction moves 2 keys x 512 B; 2 instructions per wave and tile
        const int k0 = t * CB_KT;
        unsigned short* kb = smem + buf * CB_KT * CB_C;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * 8 + wave;                       // 1 KB chunk = keys 2c, 2c+1
            const int row = 2 * c + (lane >> 5), cp = lane & 31;  // LDS (row, 16-byte column cp)
            const int col = cp ^ (row & 15);                  // ... holds data column col
            const int key = k0 + row < P ? k0 + row : 0;      // beyond P: any valid row (masked later)
            const unsigned short* src = ph + (long)key * CB_C + col * 8;
            __builtin_amdgcn_global_load_lds((const AS1 void*)src, (AS3 void*)(kb + c * 512), 16, 0, 0);
        }
    };
    // LDS byte offsets of the lane's 16 fragments inside a tile (loop-invariant)
    unsigned foff[CB_C / 16];
#pragma unroll
    for (int s = 0; s < CB_C / 16; ++s) foff[s] = (unsigned)(l31 * CB_C * 2 + (((2 * s + hi) ^ (l31 & 15)) * 16));
    const unsigned smem_base = (unsigned)(size_t)(AS3 unsigned short*)smem;

    // query fragments B[k = 16s + 8hi .. +7][j = l31].  A wave's 32 queries are ONE contiguous 16 KB block of the [P][C] array:
    // it is staged like a key tile (16 coalesced 1 KB LDS-DMA pieces into the wave's own slot, same swizzle) and the
    // fragments are read back with the key tiles' conflict-free pattern.  The direct form — every lane fetching its 16-byte
    // pieces at a 512-byte stride — touched 32 different 128-byte lines per wave instruction, four times each over the 16
    // pieces, with 128 KB per workgroup thrashing the 32 KB L1: the theta prologue, not the tile loop, was most of the pass.
    bf16x8 qf[CB_C / 16];
    {
        unsigned short* tbuf = smem + wave * CB_KT * CB_C;
        const int q0 = blockIdx.x * CB_QB + wave * 32;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int row = 2 * c + (lane >> 5), cp = lane & 31;
            const int col = cp ^ (row & 15);
            const int q = q0 + row < P ? q0 + row : 0;        // beyond P: a valid row; such lanes never store / never match
            __builtin_amdgcn_global_load_lds((const AS1 void*)(th + (long)q * CB_C + col * 8), (AS3 void*)(tbuf + c * 512), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the wave reads only what it fetched itself
        unsigned fa[CB_C / 16];
        const unsigned tb = smem_base + (unsigned)wave * (CB_KT * CB_C * 2);
#pragma unroll
        for (int s = 0; s < CB_C / 16; ++s) fa[s] = tb + foff[s];
        u32x4 fr[CB_C / 16];
        CB_RD8(fr, fa, 0);
        CB_RD8(fr, fa, 8);
        CB_WAIT8(fr, 0, 8);
        CB_WAIT8(fr, 8, 0);
#pragma unroll
        for (int s = 0; s < CB_C / 16; ++s) qf[s] = *reinterpret_cast<bf16x8*>(&fr[s]);
    }
    CB_STAMP();
    __builtin_amdgcn_s_barrier();             // every wave has its fragments: slots 0 .. 3 become the key-tile ring
    CB_STAMP();
#pragma unroll
    for (int d = 0; d < CB_NBUF - 1; ++d)
        if (t0 + d < t1) issue(t0 + d, d);

    float lmax = -INFINITY;
    float thr = 0.f;
    // pass 2: the lane's candidate keys (almost always none or one) stay in registers and are appended after the loop — an
    // atomic with return inside the loop makes the wave drain its three tiles in flight and holds the other seven at the barrier
    int ck0 = 0, ck1 = 0, ck2 = 0, ck3 = 0, cn = 0;
    if (PASS == 2) {
        // row maximum over the partial maxima of pass 1, stored [query][slot]: a lane's nslot values are contiguous (one or two
        // cache lines; as [slot][query] the ~24 loads of a lane hit 24 different lines and this prologue took 15 k cycles)
        float rm = -INFINITY;
        if (qvalid) {
            const float* pm = a.part_max + ((long)b * P + query) * a.nslot;
            for (int s = 0; s < a.nslot; ++s) rm = fmaxf(rm, pm[s]);
        }
        thr = (qvalid ? rm : INFINITY) - CB_DELTA;
    }
    // Everything issued so far has to be there before the first tile anyway (the tile fetches are the oldest operations and
    // VMEM returns in order).  The wait is a BUILTIN so that the compiler's own wait-count bookkeeping sees it: with the theta
    // loads still pending in its model it puts `s_waitcnt vmcnt(0)` in front of the loop's first MFMA — on every iteration,
    // draining the three tiles in flight.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) expcnt(7) lgkmcnt(15)
    CB_STAMP();
    for (int t = t0; t < t1; ++t) {
        const int cur = (t - t0) & (CB_NBUF - 1);
        // tile t has landed once at most the 2-instruction fetches of the tiles after it are outstanding
        if (t + 2 < t1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (t + 1 < t1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CB_STAMP();
        __builtin_amdgcn_s_barrier();        // ... for every wave's pieces; and every wave is done reading tile t-1
        CB_STAMP();
        if (t + CB_NBUF - 1 < t1) issue(t + CB_NBUF - 1, (cur + CB_NBUF - 1) & (CB_NBUF - 1));   // into tile t-1's buffer
        unsigned fa[CB_C / 16];
        const unsigned tb = smem_base + (unsigned)cur * (CB_KT * CB_C * 2);
#pragma unroll
        for (int s = 0; s < CB_C / 16; ++s) fa[s] = tb + foff[s];
        u32x4 fr[CB_C / 16];
        CB_RD8(fr, fa, 0);
        CB_RD8(fr, fa, 8);
        // TWO accumulator chains (even / odd channel groups), added at the end: a single chain of 16 dependent MFMAs runs at
        // the instruction's result latency, not its issue rate — PMC of the one-chain form: 1360 of 2880 cycles per tile in
        // issue stalls, matrix pipes 22 % busy.  (The sum order only has to be the same in both passes: it is, same code.)
        f32x16 acc, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
        CB_WAIT8(fr, 0, 8);
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&fr[s]), qf[s], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&fr[s + 1]), qf[s + 1], acc1, 0, 0, 0);
        }
        CB_WAIT8(fr, 8, 0);
#pragma unroll
        for (int s = 8; s < 16; s += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&fr[s]), qf[s], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&fr[s + 1]), qf[s + 1], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
        CB_STAMP();
        const int k0 = t * CB_KT;
        if (k0 + CB_KT > P) {          // partial last tile only (wave-uniform): keys beyond P never win
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                acc[r] = key < P ? acc[r] : -INFINITY;
            }
        }
        const float tm0 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]), tm1 = fmaxf(fmaxf(acc[3], acc[4]), acc[5]);
        const float tm2 = fmaxf(fmaxf(acc[6], acc[7]), acc[8]), tm3 = fmaxf(fmaxf(acc[9], acc[10]), acc[11]);
        const float tm4 = fmaxf(fmaxf(acc[12], acc[13]), acc[14]);
        const float tilemax = fmaxf(fmaxf(fmaxf(tm0, tm1), fmaxf(tm2, tm3)), fmaxf(tm4, acc[15]));
        if (PASS == 1) {
            lmax = fmaxf(lmax, tilemax);
        } else if (tilemax >= thr) {  // rare: some key of this lane is within DELTA of the row maximum
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (acc[r] >= thr) {
                    const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (cn >= 4) {      // a fifth candidate in one lane (clustered exemplar features): appended at once
                        const int pos = atomicAdd(a.cnt + (long)b * P + query, 1);
                        if (pos < CB_CAP) a.list[((long)b * P + query) * CB_CAP + pos] = key;
                    }
                    ck0 = cn == 0 ? key : ck0;
                    ck1 = cn == 1 ? key : ck1;
                    ck2 = cn == 2 ? key : ck2;
                    ck3 = cn == 3 ? key : ck3;
                    ++cn;
                }
            }
        }
    }
    CB_STAMP();
    if (kDvcDebug && dbgp) dbgp[255] = dbgi;
    if (PASS == 2 && cn > 0) {
        int* cq = a.cnt + (long)b * P + query;
        const int nk = cn < 4 ? cn : 4;        // (candidates beyond the fourth were appended inside the loop)
        const int pos = atomicAdd(cq, nk);
        int* lq = a.list + ((long)b * P + query) * CB_CAP;
        if (pos + 0 < CB_CAP) lq[pos + 0] = ck0;
        if (nk > 1 && pos + 1 < CB_CAP) lq[pos + 1] = ck1;
        if (nk > 2 && pos + 2 < CB_CAP) lq[pos + 2] = ck2;
        if (nk > 3 && pos + 3 < CB_CAP) lq[pos + 3] = ck3;
    }
    if (PASS == 1 && qvalid) {
        a.part_max[((long)b * P + query) * a.nslot + split * 2 + hi] = lmax;
        if (split == 0 && hi == 0) a.cnt[(long)b * P + query] = 0;      // the counter pass 2 appends to
    }
}

// one wave per query: exact fp32 re-scoring of the candidates + max / softmax / colour gather
__global__ __launch_bounds__(256) void corr_bf16_rescore_kernel(const float* __restrict__ thT,
                                                                const float* __restrict__ phT,
                                                                const float* __restrict__ blab,
                                                                const int* __restrict__ cnt,
                                                                const int* __restrict__ list, float T, int P,
                                                                int h, int w, float* __restrict__ y_small,
                                                                float* __restrict__ sim_small,
                                                                float* __restrict__ y_up,
                                                                float* __restrict__ sim_up,
                                                                int* __restrict__ argmax) {
    __shared__ __attribute__((aligned(16))) float qs[4][CB_C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    const int b = blockIdx.y;
    const bool qok = q < P;
    const float* tq = thT + ((long)b * P + (qok ? q : 0)) * CB_C;
    *reinterpret_cast<float4*>(&qs[wave][lane * 4]) = *reinterpret_cast<const float4*>(tq + lane * 4);
    __syncthreads();
    if (!qok) return;
    const float* pb = phT + (long)b * P * CB_C;
    const float* bb = blab + (long)b * 3 * P;
    const int n = cnt[(long)b * P + q];
    const bool overflow = n > CB_CAP;

    // running state of this lane over its candidates
    float m = -INFINITY, l = 0.f, y0 = 0.f, y1 = 0.f, y2 = 0.f, fbest = -INFINITY;
    int kbest = 0x7fffffff;
    auto score = [&](int k) {
        const float4* kp = reinterpret_cast<const float4*>(pb + (long)k * CB_C);
        float f = 0.f;
#pragma unroll 8
        for (int c4 = 0; c4 < CB_C / 4; ++c4) {
            const float4 kv = kp[c4];
            const float4 qv = *reinterpret_cast<const float4*>(&qs[wave][c4 * 4]);
            f = fmaf(qv.x, kv.x, f);
            f = fmaf(qv.y, kv.y, f);
            f = fmaf(qv.z, kv.z, f);
            f = fmaf(qv.w, kv.w, f);
        }
        return f;
    };
    auto accumulate = [&](int k, float f) {
        if (f > fbest || (f == fbest && k < kbest)) {
            fbest = f;
            kbest = k;
        }
        const float tt = f / T;
        const float mn = fmaxf(m, tt);
        const float sc = expf(m - mn);  // m == -inf -> 0
        const float pe = expf(tt - mn);
        l = l * sc + pe;
        y0 = y0 * sc + pe * bb[k];
        y1 = y1 * sc + pe * bb[(long)P + k];
        y2 = y2 * sc + pe * bb[2L * P + k];
        m = mn;
    };
    if (!overflow) {
        // sort the (atomically appended) keys so that lane <-> candidate is deterministic
        int key = lane < n ? list[((long)b * P + q) * CB_CAP + lane] : 0x7fffffff;
#pragma unroll
        for (int k2 = 2; k2 <= 64; k2 <<= 1)
#pragma unroll
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                const int other = __shfl_xor(key, j, 64);
                const bool up = ((lane & k2) == 0);
                const bool lower = ((lane & j) == 0);
                key = (lower == up) ? min(key, other) : max(key, other);
            }
        // The exact fp32 affinity of candidate i, by the WHOLE wave: lane c4 multiplies channels 4 c4 .. 4 c4 + 3 (one
        // coalesced 1 KB row of phi, the query's four channels from LDS), then a fixed xor tree adds the 64 partial sums —
        // a handful of candidates per query, each one load deep, instead of one lane walking a 256-channel row on its own
        // (64 dependent uncoalesced loads).  Lane i keeps candidate i's affinity; the softmax below is unchanged.
        const float4 qv = *reinterpret_cast<const float4*>(&qs[wave][lane * 4]);
        float fmine = 0.f;
        for (int i = 0; i < n; ++i) {
            const int k = __shfl(key, i, 64);
            const float4 kv = *reinterpret_cast<const float4*>(pb + (long)k * CB_C + lane * 4);
            float f = fmaf(qv.w, kv.w, fmaf(qv.z, kv.z, fmaf(qv.y, kv.y, qv.x * kv.x)));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) f += __shfl_xor(f, off, 64);
            fmine = lane == i ? f : fmine;
        }
        if (lane < n) accumulate(key, fmine);
    } else {
        for (int k = lane; k < P; k += 64) accumulate(k, score(k));
    }
    // wave reduction (fixed xor tree -> deterministic)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(m, off, 64), ol = __shfl_xor(l, off, 64);
        const float o0 = __shfl_xor(y0, off, 64), o1 = __shfl_xor(y1, off, 64), o2 = __shfl_xor(y2, off, 64);
        const float of = __shfl_xor(fbest, off, 64);
        const int ok = __shfl_xor(kbest, off, 64);
        const float mn = fmaxf(m, om);
        const float sa = (m == -INFINITY) ? 0.f : expf(m - mn), sb = (om == -INFINITY) ? 0.f : expf(om - mn);
        l = l * sa + ol * sb;
        y0 = y0 * sa + o0 * sb;
        y1 = y1 * sa + o1 * sb;
        y2 = y2 * sa + o2 * sb;
        m = mn;
        if (of > fbest || (of == fbest && ok < kbest)) {
            fbest = of;
            kbest = ok;
        }
    }
    if (lane != 0) return;
    const float yv[3] = {y0 / l, y1 / l, y2 / l};
    if (y_small)
        for (int c = 0; c < 3; ++c) y_small[((long)b * 3 + c) * P + q] = yv[c];
    if (sim_small) sim_small[(long)b * P + q] = fbest;
    if (argmax) argmax[(long)b * P + q] = kbest;
    const int qy = q / w, qx = q - qy * w;
    const long W4 = 4L * w, HW16 = 16L * P;
    if (y_up)
        for (int c = 0; c < 3; ++c) {
            float4 v = make_float4(yv[c], yv[c], yv[c], yv[c]);
            float* o = y_up + ((long)b * 3 + c) * HW16 + (4L * qy) * W4 + 4L * qx;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) *reinterpret_cast<float4*>(o + dy * W4) = v;
        }
    if (sim_up) {
        float4 v = make_float4(fbest, fbest, fbest, fbest);
        float* o = sim_up + (long)b * HW16 + (4L * qy) * W4 + 4L * qx;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) *reinterpret_cast<float4*>(o + dy * W4) = v;
    }
}

static void corr_bf16_split(int B, int P, int* ntiles, int* tps, int* nsplit) {
    int qblocks = cdiv(P, CB_QB);
    *ntiles = cdiv(P, CB_KT);
    int want = 256 / (qblocks * B);      // one 8-wave workgroup per CU
    if (want < 1) want = 1;
    if (want > *ntiles) want = *ntiles;
    *tps = cdiv(*ntiles, want);
    *nsplit = cdiv(*ntiles, *tps);
}

extern "C" size_t dvc_corr_bf16_workspace_bytes(int32_t B, int32_t P) {
    if (B <= 0 || P <= 0) return 0;
    int ntiles, tps, nsplit;
    corr_bf16_split(B, P, &ntiles, &tps, &nsplit);
    size_t n = (size_t)B * P;
    return sizeof(float) * n * nsplit * 2 + sizeof(float) * n + sizeof(int) * n + sizeof(int) * n * CB_CAP + 256;
}

extern "C" int dvc_corr_fwd_bf16(const void* theta_bf16_pc, const void* phi_bf16_pc, const float* theta_f32_pc,
                                 const float* phi_f32_pc, const float* blab, float temperature, int32_t B,
                                 int32_t C, int32_t h, int32_t w, float* y_small, float* sim_small, float* y_up,
                                 float* sim_up, int32_t* argmax, void* workspace, size_t workspace_bytes,
                                 dvcStream stream) {
    DVC_REQUIRE(theta_bf16_pc && phi_bf16_pc && theta_f32_pc && phi_f32_pc && blab && workspace,
                "dvc_corr_fwd_bf16: null argument");
    DVC_REQUIRE(C == CB_C, "dvc_corr_fwd_bf16: C must be %d", CB_C);
    DVC_REQUIRE(B > 0 && h > 0 && w > 0, "dvc_corr_fwd_bf16: bad shape");
    DVC_REQUIRE(temperature > 0.f && temperature <= 1e-4f,
                "dvc_corr_fwd_bf16: the bf16 candidate filter is exact only for temperature <= 1e-4 (got %g); "
                "use dvc_corr_fwd", (double)temperature);
    const int P = h * w;
    DVC_REQUIRE(workspace_bytes >= dvc_corr_bf16_workspace_bytes(B, P), "dvc_corr_fwd_bf16: workspace too small");
    DVC_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "dvc_corr_fwd_bf16: workspace must be 16-byte aligned");
    CorrBf16Args a;
    a.theta = reinterpret_cast<const unsigned short*>(theta_bf16_pc);
    a.phi = reinterpret_cast<const unsigned short*>(phi_bf16_pc);
    a.P = P;
    a.dbg = nullptr;
#ifdef DVC_DEBUG
    extern long long* g_corr_bf16_dbg;
    a.dbg = g_corr_bf16_dbg;
#endif
    int nsplit;
    corr_bf16_split(B, P, &a.ntiles, &a.tiles_per_split, &nsplit);
    a.nslot = nsplit * 2;
    const size_t n = (size_t)B * P;
    a.part_max = reinterpret_cast<float*>(workspace);
    a.cnt = reinterpret_cast<int*>(a.part_max + n * a.nslot + n);     // (one unused [B][P] float slot kept: workspace layout of ABI v9)
    a.list = a.cnt + n;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(cdiv(P, CB_QB), nsplit, B);
    hipLaunchKernelGGL(corr_bf16_kernel<1>, grid, dim3(512), 0, s, a);
    DVC_CHECK_LAUNCH("dvc_corr_fwd_bf16(pass1)");
    hipLaunchKernelGGL(corr_bf16_kernel<2>, grid, dim3(512), 0, s, a);      // (takes the row maxima from pass 1's partial maxima itself)
    DVC_CHECK_LAUNCH("dvc_corr_fwd_bf16(pass2)");
    hipLaunchKernelGGL(corr_bf16_rescore_kernel, dim3(cdiv(P, 4), B), dim3(256), 0, s, theta_f32_pc, phi_f32_pc, blab,
                       a.cnt, a.list, temperature, P, h, w, y_small, sim_small, y_up, sim_up, argmax);
    DVC_CHECK_LAUNCH("dvc_corr_fwd_bf16(rescore)");
    return 0;
}
