// Winograd F(2x2, 3x3) form of the 3x3 stride-1 convolutions, on the gfx950 fp32 matrix cores.
//
// Why: the 3x3 layers carry 97 % of the path's FLOPs and the direct implicit GEMM (conv_kernel.h, conv_sk_kernel.h) is
// bound by the fp32 MFMA issue rate plus per-launch costs.  The minimal-filtering form needs 16 instead
// of 36 multiplies per 2x2 output tile and input channel: 2.25x fewer MFMAs for the same result up to fp32 rounding
// (measured 2.3x the rounding error of the direct sum against an fp64 convolution, i.e. ~6e-7 of the output range —
// the same algorithm cuDNN picks for these shapes under the reference's `cudnn.benchmark = True`, test.py:140).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        per 2x2 output tile, d = its 4x4 input patch, g = the 3x3 filter
//
// GEMM view: for each of the 16 positions xi = (i, j) of the 4x4 transform domain
//   M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile]       U = G g G^T (packed once per weight), V = B^T d B
// With v_mfma_f32_32x32x2_f32 the lane (tile = lane & 31, ci parity = lane >> 5) is exactly the lane that holds the B
// operand of its own tile, so every lane reads the 4x4 patch of ITS tile from the staged input image, transforms it in
// registers (the B operands of the MFMAs), and the accumulators of a lane hold transform positions of the same (channel,
// tile) pairs: the inverse transform is per-lane register arithmetic too.  Nothing Winograd-specific ever goes through
// global memory; LDS holds what the direct kernel holds (raw input patch with halo, staged by buffer_load ... lds with
// out-of-range lanes writing the padding zeros) plus the U slice.
//
// POSITION SPLIT — two waves per SIMD.  One wave owning all 16 positions of a (32 channels x 32 tiles) block needs 256
// accumulator registers, i.e. one wave per SIMD, and then nothing covers that wave's barrier waits, LDS latencies, DMA
// issue and MFMA->VALU wait states: that kernel was built first and measured with parts of its K-step removed
// (profiles/r02_conv_wino_one_wave_dbg.txt: 29 % of a 24-GFLOP layer's time was such in-loop overhead).  Here the 16
// positions are shared by a PAIR of waves: wave half ph owns rows i = 2 ph, 2 ph + 1 (8 positions, 128 accumulator
// registers), a CU holds 8 such waves, two per SIMD — one 8-wave workgroup or two 4-wave ones (conv_wino_m1.hip), possibly of
// different launches — and the hardware interleaves them (24-46 % faster on every layer of
// the network, profiles/r02_conv_algo_sweep.txt).  What the split costs:
//   * the input transform V = B^T d B of rows 2 ph, 2 ph + 1 needs three of the four patch rows (48 instead of 64
//     bytes of LDS per lane and K-step, for half the MFMAs) and half the arithmetic: (B^T d) rows 2 ph, 2 ph + 1 and
//     their two column transforms — the VALU work per MFMA is unchanged;
//   * the inverse transform Y = A^T M A is linear in the rows of M: every wave applies it to its two rows, the pair
//     exchanges half of the partial 2x2 outputs through LDS (once per workgroup, the staging buffers are free by
//     then), and each wave finishes and stores 16 of the pair's 32 channels.
// All transform arithmetic is packed (v_pk_add_f32: two values per lane and instruction) — the fp32 MFMA and the fp32
// VALU share the SIMD's lanes, every VALU cycle is taken from the matrix rate whatever the occupancy.
//
// Dilation 2 (ColorVidNet conv5/conv6): the four pixel-parity classes of the output are four independent dilation-1
// problems on the sub-sampled grids x[2u + py][2v + px]; a workgroup works inside one class (`ss` = 2), only the
// DMA address plan and the output index know about it.
// Work decomposition: workgroup = 32*WM channels x WN blocks of 32 tiles (TR x 32/TR tiles each, stacked vertically), i.e.
// WM*WN wave pairs; the shapes in use are 4x1 and 2x2 (8 waves, 111 KB of LDS) and 2x1 (4 waves, 64 KB: two per CU);
// layers that cannot fill the chip are split over input-channel chunks (blockIdx.z), partial OUTPUT tiles (the
// inverse transform is linear) go to the split-K workspace and conv_splitk_reduce_kernel adds them in a fixed order.
#pragma once
#include <type_traits>

#include "conv_kernel.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct ConvWinoArgs {
    ConvKArgs k;        // k.w = U32[Cout/32][Cin][4][32][4] (dvc_winograd_weight_floats / ops.pack_winograd_weight)
    int ss;             // sub-grid step = dilation (1 | 2)
    int blk_y, blk_x;   // tile blocks per parity class
    int gx, gy, gz;     // logical grid: tile blocks x parity classes | channel blocks | images x input-channel splits
    // ceil(2^32 / d) for the divisors of the workgroup-index decode — exact quotients by one s_mul_hi for numerators and
    // divisors < 2^16, which the launcher guarantees (it caps the images per launch): gx, gx * gy, blk_y * blk_x, blk_x, split
    unsigned m_gx, m_gxy, m_cls, m_blkx, m_split;
    // DUAL instantiations (dvc_conv2d_winograd_dual): the reduction runs over the channels of TWO inputs with their own
    // index maps — the first cinA channels of the packed filters belong to k.x (geometry in k), the rest to x2 (geometry
    // below); k.Cin is the total.  One launch for ColorVidNet's `conv8_1(up(n7)) + conv3_3_short(n3)` pairs.
    const float* x2;
    long x2_bs;
    int cinA, H2, W2, VH2, VW2, in_up2, in_sub2;
    // dvc_conv2d_winograd_pool: the 2x2 / stride-2 max pool of the activated output, [N][Cout][OH/2][OW/2] (a lane's 2x2 output
    // tile IS a pooling window).  Unsplit launches write it from the epilogue (and skip y when k.y is NULL); split launches leave
    // it to the reduce kernel.
    float* pool;
    long pool_bs;
};
__host__ __device__ __forceinline__ unsigned wino_magic(long d) {      // (d == 1: 2^32 does not fit; 0 means "quotient = n")
    return d > 1 ? (unsigned)(((1ULL << 32) + (unsigned long long)d - 1) / (unsigned long long)d) : 0u;
}
__device__ __forceinline__ int wino_div(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

// LDS row pitch (floats) of the staged patch: even (8-byte aligned ds_read_b64) and such that the TR tile rows of a
// 32-tile block start in disjoint bank ranges
__host__ __device__ constexpr int wino_pitch(int tr) { return tr == 1 ? 66 : tr == 2 ? 48 : tr == 4 ? 24 : 12; }

// The accumulators live in FIXED accumulation registers a[16k .. 16k+15] (k = 4 (i - 2 ph) + j), touched only by the
// inline assembly below: as compiler-visible f32x16 values carried around the chunk loop, the register allocator permutes
// them between iterations and copies all of them through the VGPR file on every back edge (measured: 2x slower than the
// direct kernel).  Every statement names all 128 as clobbered, so the compiler keeps nothing of its own there.
#define WINO_A10(p) "a" #p "0", "a" #p "1", "a" #p "2", "a" #p "3", "a" #p "4", "a" #p "5", "a" #p "6", "a" #p "7", "a" #p "8", "a" #p "9"
#define WINO_AGPRS                                                                                                     \
    WINO_A10(), WINO_A10(1), WINO_A10(2), WINO_A10(3), WINO_A10(4), WINO_A10(5), WINO_A10(6), WINO_A10(7), WINO_A10(8), \
        WINO_A10(9), WINO_A10(10), WINO_A10(11), "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
// (s_nop 1: wait states between a VALU write of an operand and the MFMA that reads it — the compiler's hazard
// recogniser does not look inside the statement)
#define WINO_MFMA(K, A, B)                                                                \
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 a[%2:%3], %0, %1, a[%2:%3]"            \
                 :                                                                         \
                 : "v"(A), "v"(B), "i"((K) * 16), "i"((K) * 16 + 15)                       \
                 : WINO_AGPRS)
// packed fp32 arithmetic on register pairs (lo, hi)
#define WINO_PK_ADD(D, A, B) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define WINO_PK_SUB(D, A, B) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(D) : "v"(A), "v"(B))
// D = (A.lo + B.hi, A.lo - B.hi)
#define WINO_PK_MID(D, A, B) \
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(D) : "v"(A), "v"(B))

// Operand loads of one K-step (2 input channels: lanes 0-31 the even one, lanes 32-63 the odd one) of one wave half: three
// patch rows (3 x ds_read2_b64: rows of 4 floats at 8-byte alignment) and the 2 x 4 filter values of its two transform
// rows (2 x ds_read_b128).  Inline assembly so that the compiler does not know these touch LDS: it would otherwise put
// `s_waitcnt vmcnt(0)` in front of them (they may alias the LDS-DMA destinations) and serialise the prefetch with the
// arithmetic.  The waits are placed by hand below.
#define WINO_LOADS(D, U, XA, UA, UOFS)                                                                                 \
    asm volatile("ds_read2_b64 %0, %5 offset0:%7 offset1:%8\n\t"                                                        \
                 "ds_read2_b64 %1, %5 offset0:%9 offset1:%10\n\t"                                                       \
                 "ds_read2_b64 %2, %5 offset0:%11 offset1:%12\n\t"                                                      \
                 "ds_read_b128 %3, %6 offset:%13\n\t"                                                                   \
                 "ds_read_b128 %4, %6 offset:%14"                                                                       \
                 : "=&v"(D[0]), "=&v"(D[1]), "=&v"(D[2]), "=&v"(U[0]), "=&v"(U[1])                                      \
                 : "v"(XA), "v"(UA), "i"(0 * (PITCH / 2)), "i"(0 * (PITCH / 2) + 1), "i"(1 * (PITCH / 2)),              \
                   "i"(1 * (PITCH / 2) + 1), "i"(2 * (PITCH / 2)), "i"(2 * (PITCH / 2) + 1), "i"((UOFS) + 0),           \
                   "i"((UOFS) + 512)                                                                                    \
                 : "memory")
// The loads above are asynchronous and the compiler does not know it: every later use of their destination registers must
// depend on this statement (the registers pass through it), otherwise the compiler is free to move a use — or a register
// copy — in front of the wait.
#define WINO_WAIT_LOADS(D, U)                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                        \
                 : "+v"(D[0]), "+v"(D[1]), "+v"(D[2]), "+v"(U[0]), "+v"(U[1])                  \
                 :                                                                             \
                 : "memory")

// scalar forms of the transform arithmetic (VAR 1: A/B against the packed forms, tools/conv_wino_ab.py)
#define WINO_S_ADD(D, A, B) asm volatile("v_add_f32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
#define WINO_S_SUB(D, A, B) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))

__device__ __forceinline__ int wino_stored_offset(int VH, int VW, int W, int in_up, int in_sub, int pad_mode, int vy, int vx) {
    int sy = map_virtual(vy, VH, pad_mode);
    int sx = map_virtual(vx, VW, pad_mode);
    if (sy < 0 || sx < 0) return -1;
    if (in_up == 2) {
        sy >>= 1;
        sx >>= 1;
    } else if (in_sub == 2) {
        sy <<= 1;
        sx <<= 1;
    }
    return sy * W + sx;
}

// LDS of one workgroup (floats): NBUF staged patches + NBUF filter slices (the output exchange reuses the filter buffers)
#define WINO_NBUF 3                             // chunks c+1 and c+2 are in flight under the arithmetic of chunk c
__host__ __device__ constexpr int wino_xs_floats(int wm, int wn, int tr, int kc) {
    return ((kc * (2 * tr * wn + 2) * wino_pitch(tr) + 128 * wm * wn - 1) / (128 * wm * wn)) * (128 * wm * wn);
}
__host__ __device__ constexpr int wino_us_floats(int wm, int kc) { return wm * kc * 128 * 4; }

// The kernel body: workgroup `wlog` (logical index, 0 .. gx * gy * gz - 1) of the launch described by `s`, staging through the
// LDS arrays `xsb` (WINO_NBUF * wino_xs_floats) and `usb` (WINO_NBUF * wino_us_floats).  Shared by the one-layer kernel below
// and by conv_wino_group_kernel (several independent layers in one launch).
// SA = ConvWinoArgs in the address space the caller holds it in: generic (a by-value kernel argument) or the kernel-argument
// segment itself (address space 4, the group kernel's item table: every field read stays a scalar load from constant memory —
// a generic reference to a dynamically selected item makes the compiler copy the whole table to scratch).
template <int WM, int WN, int TR, int KC, int VAR = 0, bool DUAL = false, class SA = ConvWinoArgs>
__device__ __forceinline__ void conv_wino_body(const SA& s, const int wlog, float* const xsb, float* const usb) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto& a = s.k;
    constexpr int NPAIR = WM * WN;              // wave pairs = (32 channels x 32 tiles) blocks of the workgroup
    constexpr int NT = 128 * NPAIR;
    constexpr int TC = 32 / TR;
    constexpr int PR = 2 * TR * WN + 2;
    constexpr int PC = 2 * TC + 2;
    constexpr int PITCH = wino_pitch(TR);
    constexpr int plane = PR * PITCH;
    constexpr int EPT = (KC * plane + NT - 1) / NT;
    constexpr int C_XS = EPT * NT;
    static_assert(C_XS == wino_xs_floats(WM, WN, TR, KC), "LDS size helper out of step");
    constexpr int UQ = WM * KC * 128;
    static_assert(UQ * 4 == wino_us_floats(WM, KC), "LDS size helper out of step");
    static_assert(UQ % NT == 0, "whole staging instructions");
    static_assert(PITCH >= PC && PITCH % 2 == 0 && plane % 2 == 0 && KC % 4 == 0, "aligned patch rows, even K-steps per chunk");
    static_assert(2 * (PITCH / 2) + 1 < 256, "ds_read2_b64 offsets are 8 bits");
    constexpr int WPT = UQ / NT;
    constexpr int NI = EPT + WPT;
    static_assert(2 * NI < 64, "vmcnt is 6 bits");
    constexpr int KS = KC / 2;
    constexpr int NBUF = WINO_NBUF;
    constexpr int OOB = (int)0x80000000;
    static_assert(NBUF * UQ * 4 >= 2 * NPAIR * 32 * 64, "the output exchange fits in the filter buffers");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int ph = wave / NPAIR, pair = wave % NPAIR;      // waves w and w + NPAIR: the two halves of one block
    const int wm = pair / WN, wn = pair % WN;
    const int tr = l31 / TC, tc = l31 % TC;
    const int ss = s.ss;
    // (the decode is all wave-uniform integer division: by multiplication with host-made reciprocals — a plain `/` costs ~35
    // instructions each here, and the prologue is instruction-bound: ~1.5 us of every workgroup's 32)
    const int bz = wino_div(wlog, s.m_gxy), r_xy = wlog - bz * (s.gx * s.gy);
    const int by = wino_div(r_xy, s.m_gx), bx = r_xy - by * s.gx;
    const int per_cls = s.blk_y * s.blk_x;
    const int cls = wino_div(bx, s.m_cls), brem = bx - cls * per_cls;
    const int py = ss == 2 ? cls >> 1 : 0, px = ss == 2 ? cls & 1 : 0;
    const int tyb = wino_div(brem, s.m_blkx);
    const int ty0 = tyb * (TR * WN), tx0 = (brem - tyb * s.blk_x) * TC;
    const int b0 = by * WM;
    const int n = wino_div(bz, s.m_split), ksplit = bz - n * a.split;
    const int HWi = a.H * a.W;

    const int nchunks_all = a.Cin / KC;
    const int c_begin = ksplit * a.chunks_per_split;
    const int c_end = min(nchunks_all, c_begin + a.chunks_per_split);

    // ---- staging plan (the same for every chunk; the chunk advance is the instruction's scalar offset).  The filter half
    // comes first and its DMA for the first chunk is issued at once: it is 8x the bytes of the patch and cold (HBM / Infinity
    // Cache), and its address plan is trivial — the round trip overlaps the index arithmetic of the patch plan below.
    int gofs[EPT], wrel[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int q = tid + i * NT;
        wrel[i] = ((q / (KC * 128)) * a.Cin * 128 + q % (KC * 128)) * 16;
    }
    __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (long)b0 * a.Cin * 512), 0, WM * a.Cin * 2048, 0x00020000);
    auto issue_w = [&](int c, int buf) {
        const int sw = c * KC * 2048;
        float* us = usb + buf * (UQ * 4);
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (CONV_AS3 void*)(us + (i * NT + wave * 64) * 4), 16, wrel[i], sw, 0, 0);
    };
    issue_w(c_begin, 0);
    asm volatile(".set wino_i, 0\n\t.rept 128\n\tv_accvgpr_write_b32 a[wino_i], 0\n\t.set wino_i, wino_i + 1\n\t.endr" ::: WINO_AGPRS);
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const int e = tid + t * NT;
        const int c = e / plane, rem = e - c * plane;
        const int iy = rem / PITCH, ix = rem - iy * PITCH;
        int g = -1;
        if (e < KC * plane && ix < PC) {
            const int o = stored_offset(a, ss * (2 * ty0 - 1 + iy) + py, ss * (2 * tx0 - 1 + ix) + px);
            if (o >= 0) g = c * HWi + o;
        }
        gofs[t] = g >= 0 ? g * 4 : OOB;
    }
    const int cinA = DUAL ? s.cinA : a.Cin;        // channels that come from a.x
    __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long)n * a.x_bs), 0, cinA * HWi * 4, 0x00020000);
    // second input (DUAL): its own patch plan (another stored size / upsampling), same LDS image
    int gofs2[DUAL ? EPT : 1];
    const int HWi2 = DUAL ? s.H2 * s.W2 : 0;
    if (DUAL) {
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            const int e = tid + t * NT;
            const int c = e / plane, rem = e - c * plane;
            const int iy = rem / PITCH, ix = rem - iy * PITCH;
            int g = -1;
            if (e < KC * plane && ix < PC) {
                const int o = wino_stored_offset(s.VH2, s.VW2, s.W2, s.in_up2, s.in_sub2, a.pad_mode, ss * (2 * ty0 - 1 + iy) + py,
                                                 ss * (2 * tx0 - 1 + ix) + px);
                if (o >= 0) g = c * HWi2 + o;
            }
            gofs2[DUAL ? t : 0] = g >= 0 ? g * 4 : OOB;
        }
    }
    __amdgpu_buffer_rsrc_t rs_x2 = rs_x;
    if (DUAL)
        rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(s.x2 + (long)n * s.x2_bs), 0, (a.Cin - cinA) * HWi2 * 4, 0x00020000);
    const int nchA = cinA / KC;
    auto issue_x = [&](int c, int buf) {
        float* xs = xsb + buf * C_XS;
        // (DUAL: descriptor, scalar offset and lane offsets SELECTED, not branched on — the K loop stays one basic block, which
        // is what its hand-placed interleaving of DMA issue and MFMAs relies on)
        const bool first = !DUAL || c < nchA;
        const __amdgpu_buffer_rsrc_t rs = first ? rs_x : rs_x2;
        const int sx = first ? c * KC * HWi * 4 : (c - nchA) * KC * HWi2 * 4;
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            const int go = first ? gofs[t] : gofs2[DUAL ? t : 0];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (CONV_AS3 void*)(xs + t * NT + wave * 64), 4, go, sx, 0, 0);
        }
    };
    auto issue = [&](int c, int buf) {
        issue_x(c, buf);
        issue_w(c, buf);
    };

    // LDS byte addresses of the lane's operands inside buffer 0: patch rows ph .. ph + 2 of its tile, filter rows 2 ph, 2 ph + 1
    const unsigned xlane =
        (unsigned)(size_t)(CONV_AS3 float*)xsb + (hi * plane + (2 * (wn * TR + tr) + ph) * PITCH + 2 * tc) * 4;
    const unsigned ulane = (unsigned)(size_t)(CONV_AS3 float*)usb + (((wm * KC + hi) * 4 + 2 * ph) * 32 + l31) * 16;

    // No control flow at the chunk boundaries: each one issues exactly one more chunk — the range's last chunk again, into a
    // buffer nobody reads any more, once the range is exhausted — so the waits are compile-time constants and the K loop is
    // one basic block (VAR 3 keeps the conditional issues / waits it replaced, for A/B: profiles/r03_conv_wino_branch_free.txt)
    constexpr bool BRANCH_FREE = VAR != 3;
    issue_x(c_begin, 0);      // (the chunk's filter half is already in flight; NI instructions per chunk either way)
    if (BRANCH_FREE) {
        issue(min(c_begin + 1, c_end - 1), 1);
        issue(min(c_begin + 2, c_end - 1), 2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
    } else {
        if (c_begin + 1 < c_end) issue(c_begin + 1, 1);
        if (c_begin + 2 < c_end) issue(c_begin + 2, 2);
        if (c_begin + 2 < c_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        else if (c_begin + 1 < c_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    const float slope = a.act_slope_ptr ? *a.act_slope_ptr : a.act_slope;
    const long OHW = (long)a.OH * a.OW;
    const bool partial = a.split > 1;
    float* yb = partial ? a.part + ((long)ksplit * a.N + n) * a.Cout * OHW : a.y + (long)n * a.y_bs;
    const float* rb = (!partial && a.res) ? a.res + (long)n * a.res_bs : nullptr;

    auto run = [&](auto PHC) {
        constexpr int PH = decltype(PHC)::value;
        // operand sets (ping-pong per K-step).  R0..R2 = patch rows d[PH], d[PH+1], d[PH+2]; each row is two register
        // pairs (columns 0,1 | 2,3).  V03[i] = (V[i][0], V[i][3]), V12[i] = (V[i][1], V[i][2]) for the half's rows i.
        f32x4 D[2][3], U[2][2];
        f32x2 V03[2][2], V12[2][2];
        auto lo = [](const f32x4& v) { return __builtin_shufflevector(v, v, 0, 1); };
        auto hh = [](const f32x4& v) { return __builtin_shufflevector(v, v, 2, 3); };
        // (B^T d) rows 2 PH, 2 PH + 1:  PH 0: d0 - d2, d1 + d2   PH 1: d2 - d1, d1 - d3
        auto transform_a = [&](const f32x4 (&d)[3], f32x2 (&tl)[2], f32x2 (&th)[2]) {
            if (VAR == 1) {
                // the same 16 additions, one lane-wide instruction each (MI355X_MICROARCH.md: a packed fp32 VALU beside
                // MFMAs costs more than the scalar pair)
                if (PH == 0) {
                    WINO_S_SUB(tl[0].x, d[0].x, d[2].x); WINO_S_SUB(tl[0].y, d[0].y, d[2].y);
                    WINO_S_SUB(th[0].x, d[0].z, d[2].z); WINO_S_SUB(th[0].y, d[0].w, d[2].w);
                    WINO_S_ADD(tl[1].x, d[1].x, d[2].x); WINO_S_ADD(tl[1].y, d[1].y, d[2].y);
                    WINO_S_ADD(th[1].x, d[1].z, d[2].z); WINO_S_ADD(th[1].y, d[1].w, d[2].w);
                } else {
                    WINO_S_SUB(tl[0].x, d[1].x, d[0].x); WINO_S_SUB(tl[0].y, d[1].y, d[0].y);
                    WINO_S_SUB(th[0].x, d[1].z, d[0].z); WINO_S_SUB(th[0].y, d[1].w, d[0].w);
                    WINO_S_SUB(tl[1].x, d[0].x, d[2].x); WINO_S_SUB(tl[1].y, d[0].y, d[2].y);
                    WINO_S_SUB(th[1].x, d[0].z, d[2].z); WINO_S_SUB(th[1].y, d[0].w, d[2].w);
                }
                return;
            }
            if (PH == 0) {
                WINO_PK_SUB(tl[0], lo(d[0]), lo(d[2]));
                WINO_PK_SUB(th[0], hh(d[0]), hh(d[2]));
                WINO_PK_ADD(tl[1], lo(d[1]), lo(d[2]));
                WINO_PK_ADD(th[1], hh(d[1]), hh(d[2]));
            } else {
                WINO_PK_SUB(tl[0], lo(d[1]), lo(d[0]));
                WINO_PK_SUB(th[0], hh(d[1]), hh(d[0]));
                WINO_PK_SUB(tl[1], lo(d[0]), lo(d[2]));
                WINO_PK_SUB(th[1], hh(d[0]), hh(d[2]));
            }
        };
        // row i of V = t B:  (x - z, y + z, z - y, y - w) of t = (x, y | z, w)
        auto transform_b = [&](const f32x2& tl, const f32x2& th, f32x2& v03, f32x2& v12) {
            if (VAR == 1) {
                WINO_S_SUB(v03.x, tl.x, th.x); WINO_S_SUB(v03.y, tl.y, th.y);
                WINO_S_ADD(v12.x, th.x, tl.y); WINO_S_SUB(v12.y, th.x, tl.y);
                return;
            }
            WINO_PK_SUB(v03, tl, th);
            WINO_PK_MID(v12, th, tl);
        };
        {
            f32x2 tl[2], th[2];
            WINO_LOADS(D[0], U[0], xlane, ulane, 0);
            WINO_WAIT_LOADS(D[0], U[0]);
            transform_a(D[0], tl, th);
            transform_b(tl[0], th[0], V03[0][0], V12[0][0]);
            transform_b(tl[1], th[1], V03[0][1], V12[0][1]);
        }

        // ---- main loop: one K-step = 8 MFMAs on operand set `cs`; the loads and the transform of the next K-step's
        // operands (set `ns`) are issued among them.  The last K-step of a chunk crosses into the next chunk: wait for that
        // chunk's DMA (issued two chunks ago), barrier (every wave has also finished reading the current buffer: its
        // last reads were awaited one K-step earlier), refill the current buffer with chunk c+3.
        int buf = 0;
        for (int c = c_begin; c < c_end; ++c) {
            const unsigned xb = xlane + buf * (C_XS * 4), ub = ulane + buf * (UQ * 16);
            const int nbuf = buf == NBUF - 1 ? 0 : buf + 1;
            const unsigned xbn = xlane + nbuf * (C_XS * 4), ubn = ulane + nbuf * (UQ * 16);
            const bool more = c + 1 < c_end;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cs = ks & 1, ns = cs ^ 1;
                const bool last = ks == KS - 1;
                f32x2 tl[2], th[2];
                WINO_MFMA(0, U[cs][0].x, V03[cs][0].x);
                if (!last) {
                    WINO_LOADS(D[ns], U[ns], xb + (ks + 1) * (2 * plane * 4), ub, (ks + 1) * 4096);
                } else if (BRANCH_FREE) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
                    __builtin_amdgcn_s_barrier();
                    issue(min(c + 3, c_end - 1), buf);
                    WINO_LOADS(D[ns], U[ns], xbn, ubn, 0);
                } else if (more) {
                    if (c + 2 < c_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (c + 3 < c_end) issue(c + 3, buf);
                    WINO_LOADS(D[ns], U[ns], xbn, ubn, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                WINO_MFMA(1, U[cs][0].y, V12[cs][0].x);
                WINO_MFMA(2, U[cs][0].z, V12[cs][0].y);
                WINO_MFMA(3, U[cs][0].w, V03[cs][0].y);
                WINO_WAIT_LOADS(D[ns], U[ns]);
                __builtin_amdgcn_sched_barrier(0);
                transform_a(D[ns], tl, th);
                __builtin_amdgcn_sched_barrier(0);
                WINO_MFMA(4, U[cs][1].x, V03[cs][1].x);
                transform_b(tl[0], th[0], V03[ns][0], V12[ns][0]);
                __builtin_amdgcn_sched_barrier(0);
                WINO_MFMA(5, U[cs][1].y, V12[cs][1].x);
                transform_b(tl[1], th[1], V03[ns][1], V12[ns][1]);
                __builtin_amdgcn_sched_barrier(0);
                WINO_MFMA(6, U[cs][1].z, V12[cs][1].y);
                WINO_MFMA(7, U[cs][1].w, V03[cs][1].y);
                __builtin_amdgcn_sched_barrier(0);
            }
            buf = nbuf;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // every wave is done with the staging buffers: the exchange may overwrite them

        // ---- inverse transform of this half's rows: A^T M A with A^T = [1 1 1 0; 0 1 -1 -1] restricted to rows
        // 2 PH, 2 PH + 1 of M (m_a, m_b):  PH 0: s0 = m_a + m_b, s1 = m_b   PH 1: s0 = m_a, s1 = -m_a - m_b.
        // Register r of every accumulator = channel co(r) of the lane's tile; half PH finishes registers 8 PH .. 8 PH + 7
        // and hands its partial outputs of the other eight to its partner.
        float* ex = usb;                                   // [wave][32 values][64 lanes]
        asm volatile("s_nop 15\n\ts_nop 7" ::: WINO_AGPRS);
        auto partial_out = [&](auto RC, float (&o)[4]) {
            constexpr int r = decltype(RC)::value;
            float m[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(m[k]) : "i"(k * 16 + r) : WINO_AGPRS);
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (PH == 0) {
                    s0[j] = m[j] + m[4 + j];
                    s1[j] = m[4 + j];
                } else {
                    s0[j] = m[j];
                    s1[j] = -m[j] - m[4 + j];
                }
            }
            o[0] = s0[0] + s0[1] + s0[2];
            o[1] = s0[1] - s0[2] - s0[3];
            o[2] = s1[0] + s1[1] + s1[2];
            o[3] = s1[1] - s1[2] - s1[3];
        };
        auto for8 = [&](auto f) {
            f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{});
            f(std::integral_constant<int, 3>{}); f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{});
            f(std::integral_constant<int, 6>{}); f(std::integral_constant<int, 7>{});
        };
        {
            float* mine = ex + (wave * 32) * 64 + lane;
            for8([&](auto IC) {
                constexpr int i = decltype(IC)::value;
                float o[4];
                partial_out(std::integral_constant<int, 8 * (1 - PH) + i>{}, o);
#pragma unroll
                for (int q = 0; q < 4; ++q) mine[(i * 4 + q) * 64] = o[q];
            });
        }
        __syncthreads();
        const float* theirs = ex + ((PH == 0 ? wave + NPAIR : wave - NPAIR) * 32) * 64 + lane;
        const int u0 = 2 * (ty0 + wn * TR + tr), x0 = 2 * (tx0 + tc);
        const int oy0 = ss * u0 + py, oy1 = ss * (u0 + 1) + py;
        const int ox0 = ss * x0 + px, ox1 = ss * (x0 + 1) + px;
        const bool okx0 = ox0 < a.OW, okx1 = ox1 < a.OW;
        const bool pairst = ss == 1 && okx1 && (a.OW & 1) == 0;      // the two columns are one aligned float2
        const bool pool_here = !partial && s.pool;                   // (wave-uniform; the launcher guarantees ss == 1 then)
        float* pb = pool_here ? s.pool + (long)n * s.pool_bs : nullptr;
        const long PHW = (long)(a.OH >> 1) * (a.OW >> 1);
        for8([&](auto IC) {
            constexpr int i = decltype(IC)::value;
            constexpr int r = 8 * PH + i;
            float o[4];
            partial_out(std::integral_constant<int, r>{}, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = theirs[(i * 4 + q) * 64];
                o[q] = PH == 0 ? o[q] + t : t + o[q];      // half 0's term first, whoever adds
            }
            const int co = (b0 + wm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float bias = (!partial && a.bias) ? a.bias[co] : 0.f;
            float pooled = 0.f;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int oy = p ? oy1 : oy0;
                if (oy >= a.OH) continue;
                const long row = (long)co * OHW + (long)oy * a.OW;
                float v0 = o[2 * p] + bias, v1 = o[2 * p + 1] + bias;
                if (rb) {
                    if (okx0) v0 += rb[row + ox0];
                    if (okx1) v1 += rb[row + ox1];
                }
                if (!partial) {
                    v0 = apply_act(v0, a.act, slope);
                    v1 = apply_act(v1, a.act, slope);
                }
                if (pool_here) {
                    const float m = fmaxf(v0, v1);
                    pooled = p ? fmaxf(pooled, m) : m;
                    if (p && okx1) pb[(long)co * PHW + (long)(oy0 >> 1) * (a.OW >> 1) + (ox0 >> 1)] = pooled;   // (oy1 < OH here)
                    if (!a.y) continue;
                }
                if (pairst && VAR == 2) {
                    // write-through (sc1) stores: the lines leave the XCD's L2 while the kernel runs instead of in the
                    // end-of-kernel release (A/B only, tools/conv_wino_ab.py)
                    union { float2 f; unsigned long long u; } cv;
                    cv.f = make_float2(v0, v1);
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(yb + row + ox0), cv.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (pairst) {
                    *reinterpret_cast<float2*>(yb + row + ox0) = make_float2(v0, v1);
                } else {
                    if (okx0) yb[row + ox0] = v0;
                    if (okx1) yb[row + ox1] = v1;
                }
            }
        });
    };
    if (ph) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});
#endif
}

template <int WM, int WN, int TR, int KC, int VAR = 0, bool DUAL = false>
__global__ __launch_bounds__(128 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino_kernel(ConvWinoArgs s) {
    __shared__ __attribute__((aligned(16))) float xsb[WINO_NBUF * wino_xs_floats(WM, WN, TR, KC)];
    __shared__ __attribute__((aligned(16))) float usb[WINO_NBUF * wino_us_floats(WM, KC)];
    // XCD-aware order (speed only): the dispatcher places workgroup b of the 1-D launch on XCD b % 8 and every XCD has
    // its own L2.  The workgroups of one XCD get CONSECUTIVE logical indices, tile blocks fastest: they share one (channel
    // block, input-channel split) filter slice, which then comes from HBM / the Infinity Cache once per XCD that uses it
    // instead of once per XCD (512 -> 512 channels at 27x48: 145 -> ~40 MB per launch).  Bijective for any grid size.
    const int G = s.gx * s.gy * s.gz;
    const int xq = G / 8, xr = G % 8, xcd = blockIdx.x % 8, xi = blockIdx.x / 8;
    const int wlog = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    conv_wino_body<WM, WN, TR, KC, VAR, DUAL>(s, wlog, xsb, usb);
}

// ---- several INDEPENDENT layers in one launch (dvc_conv2d_winograd_group): WarpNet's four heads (NonlocalNet.py:451-458) are
// mutually independent, each fills the chip for about one round of workgroups with a prologue, a K loop and an epilogue that
// all of its workgroups run in lock step, and the small ones (13x24, 27x48 maps) cannot fill it at all.  As ONE grid the
// workgroups of the layers stream through the CUs back to back: different layers' workgroups have different lengths, so the
// prologue / epilogue of one overlaps the K loop of its CU neighbour, the small layers ride in the gaps, and three launch
// ramps and tails disappear.  Every item keeps the plan (tile-block shape TR, split over input channels) it would get alone
// and runs the same body: results are bit-identical to the per-layer launches.
// Grid: every item's workgroup count rounded up to a multiple of 8; XCD x (blockIdx.x % 8) walks item 0's x-th eighth, then
// item 1's, ...: each layer is spread over all XCDs (an XCD-contiguous split of the whole grid would give XCD 0 nothing but the
// longest layer), and within an XCD consecutive workgroups still share a filter slice.
#define WINO_GROUP_MAX 4
struct ConvWinoGroupArgs {
    ConvWinoArgs item[WINO_GROUP_MAX];
    int per_xcd[WINO_GROUP_MAX + 1];   // prefix sums of ceil(G_i / 8)
    int tr[WINO_GROUP_MAX];
    int n;
};
typedef __attribute__((address_space(4))) ConvWinoArgs ConvWinoArgsK;     // ... as it lies in the kernel-argument segment
template <int WM, int WN, int KC>
__global__ __launch_bounds__(128 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino_group_kernel(ConvWinoGroupArgs g) {
    constexpr int XS = wino_xs_floats(WM, WN, 1, KC) > wino_xs_floats(WM, WN, 8, KC) ? wino_xs_floats(WM, WN, 1, KC) : wino_xs_floats(WM, WN, 8, KC);
    static_assert(XS >= wino_xs_floats(WM, WN, 2, KC) && XS >= wino_xs_floats(WM, WN, 4, KC), "largest patch buffer");
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float xsb[WINO_NBUF * XS];
    __shared__ __attribute__((aligned(16))) float usb[WINO_NBUF * wino_us_floats(WM, KC)];
    const int xcd = blockIdx.x % 8, xi = blockIdx.x / 8;
    // (compile-time indices only: a dynamic index into the by-value argument block would move it to scratch)
    int it = 0, base = 0, q = g.per_xcd[1], tr = g.tr[0];
#pragma unroll
    for (int i = 1; i < WINO_GROUP_MAX; ++i)
        if (i < g.n && xi >= g.per_xcd[i]) {
            it = i;
            base = g.per_xcd[i];
            q = g.per_xcd[i + 1] - g.per_xcd[i];
            tr = g.tr[i];
        }
    const int wlog = xcd * q + (xi - base);
    // item `it` where it lies in the kernel-argument segment (item[] is the first member of the only argument)
    static_assert(__builtin_offsetof(ConvWinoGroupArgs, item) == 0, "item table at the start of the argument block");
    const ConvWinoArgsK& s = ((const ConvWinoArgsK*)__builtin_amdgcn_kernarg_segment_ptr())[it];
    if (wlog >= s.gx * s.gy * s.gz) return;       // padding workgroup of this item's last eighth
    switch (tr) {
        case 1: conv_wino_body<WM, WN, 1, KC, 0, false, ConvWinoArgsK>(s, wlog, xsb, usb); break;
        case 2: conv_wino_body<WM, WN, 2, KC, 0, false, ConvWinoArgsK>(s, wlog, xsb, usb); break;
        case 4: conv_wino_body<WM, WN, 4, KC, 0, false, ConvWinoArgsK>(s, wlog, xsb, usb); break;
        default: conv_wino_body<WM, WN, 8, KC, 0, false, ConvWinoArgsK>(s, wlog, xsb, usb); break;
    }
#endif
}

template <int WM, int WN, int KC>
static void conv_wino_launch_shape(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s) {
    constexpr int NT = 128 * WM * WN;
#ifdef DVC_DEBUG
    if (s.k.dbg & 8) {      // dvc_debug_conv_variant(8): conditional chunk issues / waits in the K loop (A/B only)
        switch (tr) {
            case 1: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 1, KC, 3>), grid, dim3(NT), 0, st, s); break;
            case 2: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 2, KC, 3>), grid, dim3(NT), 0, st, s); break;
            case 4: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 4, KC, 3>), grid, dim3(NT), 0, st, s); break;
            default: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 8, KC, 3>), grid, dim3(NT), 0, st, s); break;
        }
        return;
    }
    if (s.k.dbg & 32) {     // dvc_debug_conv_variant(32): write-through output stores (A/B only)
        switch (tr) {
            case 1: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 1, KC, 2>), grid, dim3(NT), 0, st, s); break;
            case 2: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 2, KC, 2>), grid, dim3(NT), 0, st, s); break;
            case 4: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 4, KC, 2>), grid, dim3(NT), 0, st, s); break;
            default: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 8, KC, 2>), grid, dim3(NT), 0, st, s); break;
        }
        return;
    }
    if (s.k.dbg & 16) {     // dvc_debug_conv_variant(16): scalar transform arithmetic (A/B only)
        switch (tr) {
            case 1: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 1, KC, 1>), grid, dim3(NT), 0, st, s); break;
            case 2: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 2, KC, 1>), grid, dim3(NT), 0, st, s); break;
            case 4: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 4, KC, 1>), grid, dim3(NT), 0, st, s); break;
            default: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 8, KC, 1>), grid, dim3(NT), 0, st, s); break;
        }
        return;
    }
#endif
    // (dvc_debug_conv_variant(512), debug build: pad the launch with dynamic LDS so that only ONE such workgroup fits a CU next
    // to a 61 KB one — tools/bg_split_probe.py asks whether background launches that leave a slot per CU free help the chain)
    const unsigned dyn = (kDvcDebug && (s.k.dbg & 512)) ? 36u * 1024u : 0u;
    switch (tr) {
        case 1: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 1, KC>), grid, dim3(NT), dyn, st, s); break;
        case 2: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 2, KC>), grid, dim3(NT), dyn, st, s); break;
        case 4: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 4, KC>), grid, dim3(NT), dyn, st, s); break;
        default: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 8, KC>), grid, dim3(NT), dyn, st, s); break;
    }
}

template <int WM, int WN, int KC>
static void conv_wino_launch_shape_dual(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s) {
    constexpr int NT = 128 * WM * WN;
    switch (tr) {
        case 1: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 1, KC, 0, true>), grid, dim3(NT), 0, st, s); break;
        case 2: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 2, KC, 0, true>), grid, dim3(NT), 0, st, s); break;
        case 4: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 4, KC, 0, true>), grid, dim3(NT), 0, st, s); break;
        default: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 8, KC, 0, true>), grid, dim3(NT), 0, st, s); break;
    }
}

void conv_wino_launch_m1_dual(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s);   // the two-input form, 64 x 32 shape only
void conv_wino_launch_m4(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s);   // 128 channels x 32 tiles, 8 waves
void conv_wino_launch_m2(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s);   // 64 channels x 64 tiles, 8 waves
void conv_wino_launch_m1(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s);   // 64 channels x 32 tiles, 4 waves, two per CU
void conv_wino_launch_m0(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s);   // 32 channels x 32 tiles, 2 waves, four per CU
void conv_wino_launch_group_m1(dim3 grid, hipStream_t st, const ConvWinoGroupArgs& g);  // several layers, 64 x 32 shape (conv_wino_group.hip)
