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
// One wave owns 32 output channels x 32 tiles x ALL 16 positions: 16 independent accumulators (256 registers, one
// wave per SIMD).  With v_mfma_f32_32x32x2_f32 the lane (tile = lane & 31, ci parity = lane >> 5) is exactly the lane
// that holds the B operand of its own tile, so every lane reads the 4x4 patch of ITS tile from the staged input image
// (8 ds_read_b64), transforms it in registers (32 adds -> the B operands of 16 MFMAs), and the 16 accumulators of a lane
// hold all 16 positions of the same (channel, tile) pairs: the inverse transform is per-lane register arithmetic too.
// Nothing Winograd-specific ever goes through memory; LDS holds what the direct kernel holds (raw input patch with
// halo, staged by buffer_load ... lds with out-of-range lanes writing the padding zeros) plus the U slice.
//
// Dilation 2 (ColorVidNet conv5/conv6): the four pixel-parity classes of the output are four independent dilation-1
// problems on the sub-sampled grids x[2u + py][2v + px]; a workgroup works inside one class (`ss` = 2), only the
// DMA address plan and the output index know about it.
// Work decomposition: workgroup = 32*WM channels x WN blocks of 32 tiles (TR x 32/TR tiles each, stacked vertically);
// layers that cannot fill the chip are split over input-channel chunks (blockIdx.z), partial OUTPUT tiles (the
// inverse transform is linear) go to the split-K workspace and conv_splitk_reduce_kernel adds them in a fixed order.
#pragma once
#include "conv_kernel.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvWinoArgs {
    ConvKArgs k;        // k.w = U32[Cout/32][Cin][4][32][4] (dvc_winograd_weight_floats / ops.pack_winograd_weight)
    int ss;             // sub-grid step = dilation (1 | 2)
    int blk_y, blk_x;   // tile blocks per parity class
};

// LDS row pitch (floats) of the staged patch: even (8-byte aligned ds_read_b64) and such that the TR tile rows of a
// 32-tile block start in disjoint bank ranges
__host__ __device__ constexpr int wino_pitch(int tr) { return tr == 1 ? 66 : tr == 2 ? 48 : tr == 4 ? 24 : 12; }

// The 16 accumulators live in FIXED accumulation registers a[16k .. 16k+15] (k = 4i + j), touched only by the inline
// assembly below: as compiler-visible f32x16 values carried around the chunk loop, the register allocator permutes them
// between iterations and copies all 256 through the VGPR file on every back edge (measured: 2x slower than the direct
// kernel).  Every statement names all 256 as clobbered, so the compiler keeps nothing of its own there.
#define WINO_A10(p) "a" #p "0", "a" #p "1", "a" #p "2", "a" #p "3", "a" #p "4", "a" #p "5", "a" #p "6", "a" #p "7", "a" #p "8", "a" #p "9"
#define WINO_AGPRS                                                                                                      \
    WINO_A10(), WINO_A10(1), WINO_A10(2), WINO_A10(3), WINO_A10(4), WINO_A10(5), WINO_A10(6), WINO_A10(7), WINO_A10(8), \
        WINO_A10(9), WINO_A10(10), WINO_A10(11), WINO_A10(12), WINO_A10(13), WINO_A10(14), WINO_A10(15), WINO_A10(16),  \
        WINO_A10(17), WINO_A10(18), WINO_A10(19), WINO_A10(20), WINO_A10(21), WINO_A10(22), WINO_A10(23), WINO_A10(24), \
        "a250", "a251", "a252", "a253", "a254", "a255"
// (s_nop 1: wait states between a VALU write of an operand and the MFMA that reads it — the compiler's hazard
// recogniser does not look inside the statement)
#define WINO_MFMA(K, A, B)                                                                 \
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 a[%2:%3], %0, %1, a[%2:%3]"            \
                 :                                                                         \
                 : "v"(A), "v"(B), "i"((K) * 16), "i"((K) * 16 + 15)                       \
                 : WINO_AGPRS)

// Operand loads of one K-step (2 input channels: lanes 0-31 the even one, lanes 32-63 the odd one): the lane's 4x4 patch
// (4 x ds_read2_b64: rows of 4 floats at 8-byte alignment) and its 4 x 4 transform-domain filter values (4 x ds_read_b128).
// Inline assembly so that the compiler does not know these touch LDS: it would otherwise put `s_waitcnt vmcnt(0)` in front
// of them (they may alias the LDS-DMA destinations) and serialise the prefetch with the arithmetic.  The waits are
// placed by hand below.
#define WINO_LOADS(D, U, XA, UA, UOFS)                                                                                       \
    asm volatile("ds_read2_b64 %0, %8 offset0:%10 offset1:%11\n\t"                                                           \
                 "ds_read2_b64 %1, %8 offset0:%12 offset1:%13\n\t"                                                           \
                 "ds_read2_b64 %2, %8 offset0:%14 offset1:%15\n\t"                                                           \
                 "ds_read2_b64 %3, %8 offset0:%16 offset1:%17\n\t"                                                           \
                 "ds_read_b128 %4, %9 offset:%18\n\t"                                                                        \
                 "ds_read_b128 %5, %9 offset:%19\n\t"                                                                        \
                 "ds_read_b128 %6, %9 offset:%20\n\t"                                                                        \
                 "ds_read_b128 %7, %9 offset:%21"                                                                            \
                 : "=&v"(D[0]), "=&v"(D[1]), "=&v"(D[2]), "=&v"(D[3]), "=&v"(U[0]), "=&v"(U[1]), "=&v"(U[2]), "=&v"(U[3])    \
                 : "v"(XA), "v"(UA), "i"(0 * (PITCH / 2)), "i"(0 * (PITCH / 2) + 1), "i"(1 * (PITCH / 2)),                   \
                   "i"(1 * (PITCH / 2) + 1), "i"(2 * (PITCH / 2)), "i"(2 * (PITCH / 2) + 1), "i"(3 * (PITCH / 2)),           \
                   "i"(3 * (PITCH / 2) + 1), "i"((UOFS) + 0), "i"((UOFS) + 512), "i"((UOFS) + 1024), "i"((UOFS) + 1536)      \
                 : "memory")

// The loads above are asynchronous and the compiler does not know it: every later use of their destination registers must
// depend on this statement (the registers pass through it), otherwise the compiler is free to move a use — or a register
// copy — in front of the wait.
#define WINO_WAIT_LOADS(D, U)                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                 : "+v"(D[0]), "+v"(D[1]), "+v"(D[2]), "+v"(D[3]), "+v"(U[0]), "+v"(U[1]), "+v"(U[2]), "+v"(U[3])       \
                 :                                                                                                      \
                 : "memory")

// DBG (timing experiments only, wrong results): 4 = no patch transform (VALU), 8 = no operand loads (LDS), 16 = no barrier
template <int WM, int WN, int TR, int KC, int DBG = 0>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_wino_kernel(ConvWinoArgs s) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ConvKArgs& a = s.k;
    constexpr int NT = 64 * WM * WN;
    constexpr int TC = 32 / TR;                 // tiles per block row
    constexpr int PR = 2 * TR * WN + 2;         // patch rows (2 per tile row + halo)
    constexpr int PC = 2 * TC + 2;              // patch columns in use
    constexpr int PITCH = wino_pitch(TR);
    constexpr int plane = PR * PITCH;
    constexpr int EPT = (KC * plane + NT - 1) / NT;
    constexpr int C_XS = EPT * NT;              // every lane of every staging load owns an LDS cell
    constexpr int UQ = WM * KC * 128;           // float4 pieces of one weight chunk: [WM][KC][4][32] x float4
    static_assert(UQ % NT == 0, "whole staging instructions");
    static_assert(PITCH >= PC && PITCH % 2 == 0 && plane % 2 == 0 && KC % 4 == 0, "aligned patch rows, even K-steps per chunk");
    static_assert(3 * (PITCH / 2) + 1 < 256, "ds_read2_b64 offsets are 8 bits");
    constexpr int WPT = UQ / NT;
    constexpr int NI = EPT + WPT;               // staging instructions per thread and chunk
    static_assert(2 * NI < 64, "vmcnt is 6 bits");
    constexpr int KS = KC / 2;                  // K-steps per chunk
    constexpr int NBUF = 3;                     // chunks c+1 and c+2 are in flight under the arithmetic of chunk c
    constexpr int OOB = (int)0x80000000;
    __shared__ __attribute__((aligned(16))) float xsb[NBUF * C_XS];
    __shared__ __attribute__((aligned(16))) float usb[NBUF * UQ * 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tr = l31 / TC, tc = l31 % TC;
    const int ss = s.ss;
    const int per_cls = s.blk_y * s.blk_x;
    const int cls = blockIdx.x / per_cls, brem = blockIdx.x % per_cls;
    const int py = cls / ss, px = cls % ss;
    const int ty0 = (brem / s.blk_x) * (TR * WN), tx0 = (brem % s.blk_x) * TC;   // first tile (sub-grid tile coordinates)
    const int b0 = blockIdx.y * WM;                                                // first 32-channel block
    const int n = blockIdx.z / a.split, ksplit = blockIdx.z % a.split;
    const int HWi = a.H * a.W;

    const int nchunks_all = a.Cin / KC;
    const int c_begin = ksplit * a.chunks_per_split;
    const int c_end = min(nchunks_all, c_begin + a.chunks_per_split);

    // ---- staging plan (the same for every chunk; the chunk advance is the instruction's scalar offset)
    int gofs[EPT], wrel[WPT];
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
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int q = tid + i * NT;
        wrel[i] = ((q / (KC * 128)) * a.Cin * 128 + q % (KC * 128)) * 16;
    }
    __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (long)b0 * a.Cin * 512), 0, WM * a.Cin * 2048, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long)n * a.x_bs), 0, a.Cin * HWi * 4, 0x00020000);
    // (a.dbg: timing experiments only, dvc_debug_conv_variant — bit 0: no patch DMA after the prologue, bit 1: no filter
    // DMA after the prologue; results are wrong)
    const bool dbg_nox = a.dbg & 1, dbg_nou = a.dbg & 2;
    auto issue = [&](int c, int buf) {
        const int sx = c * KC * HWi * 4, sw = c * KC * 2048;
        float* xs = xsb + buf * C_XS;
        float* us = usb + buf * (UQ * 4);
        if (!(dbg_nox && c > c_begin + 2)) {
#pragma unroll
            for (int t = 0; t < EPT; ++t)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (CONV_AS3 void*)(xs + t * NT + wave * 64), 4, gofs[t], sx, 0, 0);
        }
        if (!(dbg_nou && c > c_begin + 2)) {
#pragma unroll
            for (int i = 0; i < WPT; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (CONV_AS3 void*)(us + (i * NT + wave * 64) * 4), 16, wrel[i], sw, 0, 0);
        }
    };

    asm volatile(".set wino_i, 0\n\t.rept 256\n\tv_accvgpr_write_b32 a[wino_i], 0\n\t.set wino_i, wino_i + 1\n\t.endr" ::: WINO_AGPRS);

    // LDS byte addresses of the lane's operands inside buffer 0
    const unsigned xlane = (unsigned)(size_t)(CONV_AS3 float*)xsb + (hi * plane + 2 * (wn * TR + tr) * PITCH + 2 * tc) * 4;
    const unsigned ulane = (unsigned)(size_t)(CONV_AS3 float*)usb + (((wm * KC + hi) * 4) * 32 + l31) * 16;

    // ---- prologue: up to three chunks in flight, the first one awaited
    issue(c_begin, 0);
    if (c_begin + 1 < c_end) issue(c_begin + 1, 1);
    if (c_begin + 2 < c_end) issue(c_begin + 2, 2);
    if (c_begin + 2 < c_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
    else if (c_begin + 1 < c_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // operand sets (ping-pong per K-step): patch rows, filter values, transformed patch
    f32x4 D[2][4], U[2][4];       // (native vector type: the struct float4 cannot be an in/out operand of an asm statement)
    float V[2][16];
    auto transform_a = [&](const f32x4 (&d)[4], f32x4 (&t)[4], int r) {      // row r of t = B^T d
        if (r == 0) t[0] = d[0] - d[2];
        if (r == 1) t[1] = d[1] + d[2];
        if (r == 2) t[2] = d[2] - d[1];
        if (r == 3) t[3] = d[1] - d[3];
        asm volatile("" : "+v"(t[r]));
    };
    auto transform_b = [&](const f32x4 (&t)[4], float (&v)[16], int i) {      // row i of V = t B
        v[i * 4 + 0] = t[i].x - t[i].z;
        v[i * 4 + 1] = t[i].y + t[i].z;
        v[i * 4 + 2] = t[i].z - t[i].y;
        v[i * 4 + 3] = t[i].y - t[i].w;
        // (pins the four operations here: without a use in this basic block the compiler sinks them behind the branch
        // at the end of the K-step, out of the MFMAs' shadow)
        asm volatile("" : "+v"(v[i * 4 + 0]), "+v"(v[i * 4 + 1]), "+v"(v[i * 4 + 2]), "+v"(v[i * 4 + 3]));
    };
    {
        f32x4 t[4];
        WINO_LOADS(D[0], U[0], xlane, ulane, 0);
        WINO_WAIT_LOADS(D[0], U[0]);
#pragma unroll
        for (int r = 0; r < 4; ++r) transform_a(D[0], t, r);
#pragma unroll
        for (int i = 0; i < 4; ++i) transform_b(t, V[0], i);
    }

    // ---- main loop.  One K-step = 16 MFMAs (1024 cycles of the matrix pipe) on operand set `cs`, with the loads and the
    // transform of the next K-step's operands (set `ns`) issued in their shadow.  The last K-step of a chunk crosses into
    // the next chunk: wait for that chunk's DMA (issued two chunks ago), barrier (every wave has also finished reading
    // the current buffer: its last reads were awaited one K-step earlier), refill the current buffer with chunk c+3.
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
            f32x4 t[4];
            WINO_MFMA(0, U[cs][0].x, V[cs][0]);
            WINO_MFMA(1, U[cs][0].y, V[cs][1]);
            if (!last) {
                if (!(DBG & 8)) WINO_LOADS(D[ns], U[ns], xb + (ks + 1) * (2 * plane * 4), ub, (ks + 1) * 4096);
            } else if (more) {
                if (c + 2 < c_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(DBG & 16)) __builtin_amdgcn_s_barrier();
                if (c + 3 < c_end) issue(c + 3, buf);
                if (!(DBG & 8)) WINO_LOADS(D[ns], U[ns], xbn, ubn, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            WINO_MFMA(2, U[cs][0].z, V[cs][2]);
            WINO_MFMA(3, U[cs][0].w, V[cs][3]);
            WINO_MFMA(4, U[cs][1].x, V[cs][4]);
            WINO_MFMA(5, U[cs][1].y, V[cs][5]);
            WINO_WAIT_LOADS(D[ns], U[ns]);
            __builtin_amdgcn_sched_barrier(0);
            if (!(DBG & 4)) transform_a(D[ns], t, 0);
            __builtin_amdgcn_sched_barrier(0);
            WINO_MFMA(6, U[cs][1].z, V[cs][6]);
            if (!(DBG & 4)) transform_a(D[ns], t, 1);
            __builtin_amdgcn_sched_barrier(0);
            WINO_MFMA(7, U[cs][1].w, V[cs][7]);
            if (!(DBG & 4)) transform_a(D[ns], t, 2);
            __builtin_amdgcn_sched_barrier(0);
            WINO_MFMA(8, U[cs][2].x, V[cs][8]);
            if (!(DBG & 4)) transform_a(D[ns], t, 3);
            __builtin_amdgcn_sched_barrier(0);
            WINO_MFMA(9, U[cs][2].y, V[cs][9]);
            if (!(DBG & 4)) transform_b(t, V[ns], 0);
            __builtin_amdgcn_sched_barrier(0);
            WINO_MFMA(10, U[cs][2].z, V[cs][10]);
            if (!(DBG & 4)) transform_b(t, V[ns], 1);
            __builtin_amdgcn_sched_barrier(0);
            WINO_MFMA(11, U[cs][2].w, V[cs][11]);
            if (!(DBG & 4)) transform_b(t, V[ns], 2);
            __builtin_amdgcn_sched_barrier(0);
            WINO_MFMA(12, U[cs][3].x, V[cs][12]);
            if (!(DBG & 4)) transform_b(t, V[ns], 3);
            __builtin_amdgcn_sched_barrier(0);
            WINO_MFMA(13, U[cs][3].y, V[cs][13]);
            WINO_MFMA(14, U[cs][3].z, V[cs][14]);
            WINO_MFMA(15, U[cs][3].w, V[cs][15]);
            __builtin_amdgcn_sched_barrier(0);
        }
        buf = nbuf;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- inverse transform Y = A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]) and epilogue; register r of every accumulator
    // belongs to output channel co(r) of the lane's tile
    const float slope = a.act_slope_ptr ? *a.act_slope_ptr : a.act_slope;
    const long OHW = (long)a.OH * a.OW;
    const bool partial = a.split > 1;
    float* yb = partial ? a.part + ((long)ksplit * a.N + n) * a.Cout * OHW : a.y + (long)n * a.y_bs;
    const float* rb = (!partial && a.res) ? a.res + (long)n * a.res_bs : nullptr;
    const int u0 = 2 * (ty0 + wn * TR + tr), x0 = 2 * (tx0 + tc);
    const int oy0 = ss * u0 + py, oy1 = ss * (u0 + 1) + py;
    const int ox0 = ss * x0 + px, ox1 = ss * (x0 + 1) + px;
    const bool okx0 = ox0 < a.OW, okx1 = ox1 < a.OW;
    const bool pair = ss == 1 && okx1 && (a.OW & 1) == 0;      // the two columns are one aligned float2
    asm volatile("s_nop 15\n\ts_nop 7" ::: WINO_AGPRS);     // the last MFMAs' results before the first accumulator read
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float m[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(m[k]) : "i"(k * 16 + r) : WINO_AGPRS);
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = m[0 + j] + m[4 + j] + m[8 + j];
            s1[j] = m[4 + j] - m[8 + j] - m[12 + j];
        }
        float o[2][2];
        o[0][0] = s0[0] + s0[1] + s0[2];
        o[0][1] = s0[1] - s0[2] - s0[3];
        o[1][0] = s1[0] + s1[1] + s1[2];
        o[1][1] = s1[1] - s1[2] - s1[3];
        const int co = (b0 + wm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float bias = (!partial && a.bias) ? a.bias[co] : 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int oy = p ? oy1 : oy0;
            if (oy >= a.OH) continue;
            const long row = (long)co * OHW + (long)oy * a.OW;
            float v0 = o[p][0] + bias, v1 = o[p][1] + bias;
            if (rb) {
                if (okx0) v0 += rb[row + ox0];
                if (okx1) v1 += rb[row + ox1];
            }
            if (!partial) {
                v0 = apply_act(v0, a.act, slope);
                v1 = apply_act(v1, a.act, slope);
            }
            if (pair) {
                *reinterpret_cast<float2*>(yb + row + ox0) = make_float2(v0, v1);
            } else {
                if (okx0) yb[row + ox0] = v0;
                if (okx1) yb[row + ox1] = v1;
            }
        }
    }
#endif
}

// ---- launchers (one translation unit per workgroup shape)
template <int WM, int WN, int KC>
static void conv_wino_launch_shape(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s) {
    constexpr int NT = 64 * WM * WN;
    switch (tr) {
        case 1: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 1, KC>), grid, dim3(NT), 0, st, s); break;
        case 2: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 2, KC>), grid, dim3(NT), 0, st, s); break;
        case 4: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 4, KC>), grid, dim3(NT), 0, st, s); break;
        default: hipLaunchKernelGGL((conv_wino_kernel<WM, WN, 8, KC>), grid, dim3(NT), 0, st, s); break;
    }
}

void conv_wino_launch_m4(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s);   // 128 channels x 32 tiles
void conv_wino_launch_m2(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s);   // 64 channels x 64 tiles
void conv_wino_launch_m4_dbg(int variant, dim3 grid, hipStream_t st, const ConvWinoArgs& s);   // 1x32-tile blocks only
