// Stream-K form of the implicit-GEMM convolution (conv_kernel.h) for the LDS-DMA ("plain") layers.
//
// Why: the layers of this network are small next to the chip.  A 256->256 3x3 layer at 54x96 is 1296
// 32x32 output tiles of 1152 dependent MFMAs each, for 1024 SIMDs: with one tile per wave some SIMDs get
// two tiles and the rest one, and the launch takes two tile-times whatever the kernel does inside
// (profiles/r01_conv_layer_sweep.json: 67-76 us against a 39 us MFMA floor; split-K only trades that for a
// second launch and a half-empty tail round).  Here the work is cut in units of one LDS chunk (CK input
// channels of one 64x64 output tile = 36 MFMAs per wave), ordered tile after tile, and every workgroup of a
// grid that exactly fills the chip (k workgroups per CU) walks an EQUAL contiguous range of units:
//     unit u = tile * NC + chunk,   tile = (n * co_blocks + co_block) * px_tiles + px_tile
//     workgroup w owns [w*U/G, (w+1)*U/G)
// A range crosses tile boundaries; the accumulator is flushed at each boundary — straight to y (bias, skip,
// activation) when the workgroup covered the whole tile, otherwise as a raw register image into a
// per-workgroup slot; a second small kernel adds the slots of every split tile in ascending-K order
// (deterministic) and applies the epilogue.  The chunk pipeline (LDS-DMA of unit u+1 under the MFMAs of
// unit u, one barrier per unit) runs THROUGH the tile boundaries: only the DMA address plan is recomputed.
// Tiles are ordered output-channel-block major and consecutive ranges go to workgroups of the same XCD
// (block b runs on XCD b % 8), so an XCD's L2 holds the weight slices of one or two channel blocks.
#pragma once
#include "conv_kernel.h"

struct ConvSkArgs {
    ConvKArgs k;     // tensors + geometry, as for the tile-per-workgroup kernel (split fields unused)
    int tiles_x;     // pixel tiles per row
    int px_tiles;    // pixel tiles per image
    int co_blocks;   // Cout / MT
    int NC;          // chunks (units) per tile = Cin / CK
    long U;          // units = N * co_blocks * px_tiles * NC
    float* part;     // [G][2][RM*RN*4][NT] float4: raw accumulator images (thread-major 16-byte pieces)
};

__host__ __device__ __forceinline__ long sk_unit_start(long w, long U, long G) { return w * U / G; }
__host__ __device__ __forceinline__ long sk_unit_owner(long u, long U, long G) { return ((u + 1) * G - 1) / U; }

struct SkTile {
    int n, m0, ox0, oy0;
};
__device__ __forceinline__ SkTile sk_decode(const ConvSkArgs& s, int tile, int MT, int TW, int PH) {
    const int pt = tile % s.px_tiles, r = tile / s.px_tiles;
    const int cb = r % s.co_blocks, n = r / s.co_blocks;
    SkTile t;
    t.n = n;
    t.m0 = cb * MT;
    t.ox0 = (pt % s.tiles_x) * TW;
    t.oy0 = (pt / s.tiles_x) * PH;
    return t;
}

// XCD-aware logical workgroup index (speed only; bijective for any G)
__device__ __forceinline__ long sk_logical_wg(long b, long G) {
    const long xq = G / 8, xr = G % 8, xcd = b % 8, xi = b / 8;
    return (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
}

// Sum of the slots of one split tile in ascending-K (workgroup) order + bias / skip / activation -> y.  Executed by all
// NT threads of one workgroup for the accumulator registers [k0, k0 + KPB) of every thread (thread <-> element map of
// the main kernel).  slot 0 = the segment at the START of a workgroup's range, slot 1 = the one at its end: only the
// first contributor can hold the tile at the end of its range; every later contributor's range starts inside the tile.
template <int WM, int WN, int RM, int RN, int TW, int K0, int KPB>
__device__ __forceinline__ void sk_fix_tile(const ConvSkArgs& s, long G, int tile, int tid) {
    const ConvKArgs& a = s.k;
    constexpr int NT = 64 * WM * WN;
    constexpr int MT = 32 * WM * RM;
    constexpr int RPT = 32 / TW;
    constexpr int PH = WN * RN * RPT;
    constexpr int NQ = RM * RN * 4;          // float4 pieces per thread and slot
    static_assert(K0 % 4 == 0 && KPB % 4 == 0, "whole float4 pieces");
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const long uf = (long)tile * s.NC, ul = uf + s.NC - 1;
    const long wf = sk_unit_owner(uf, s.U, G), wl = sk_unit_owner(ul, s.U, G);
    const int slot_f = (sk_unit_start(wf, s.U, G) / s.NC == tile) ? 0 : 1;
    float4 v[KPB / 4];
    {
        const float4* pp = reinterpret_cast<const float4*>(s.part) + ((wf * 2 + slot_f) * NQ + K0 / 4) * NT + tid;
#pragma unroll
        for (int q = 0; q < KPB / 4; ++q) v[q] = pp[(long)q * NT];
    }
    for (long w = wf + 1; w <= wl; ++w) {
        const float4* pp = reinterpret_cast<const float4*>(s.part) + ((w * 2) * NQ + K0 / 4) * NT + tid;
#pragma unroll
        for (int q = 0; q < KPB / 4; ++q) {
            const float4 t = pp[(long)q * NT];
            v[q].x += t.x; v[q].y += t.y; v[q].z += t.z; v[q].w += t.w;
        }
    }
    const SkTile t_ = sk_decode(s, tile, MT, TW, PH);
    const float slope = a.act_slope_ptr ? *a.act_slope_ptr : a.act_slope;
    const long OHW = (long)a.OH * a.OW;
    float* yn = a.y + (long)t_.n * a.y_bs;
    const float* rn_ = a.res ? a.res + (long)t_.n * a.res_bs : nullptr;
    const int pr = l31 / TW, pc = l31 % TW;
#pragma unroll
    for (int k = 0; k < KPB; ++k) {
        const int kk = K0 + k, ij = kk / 16, r = kk % 16, i = ij / RN, j = ij % RN;
        const int oy = t_.oy0 + (wn * RN + j) * RPT + pr, ox = t_.ox0 + pc;
        if (oy >= a.OH || ox >= a.OW) continue;
        const long pix = (long)oy * a.OW + ox;
        const int co = t_.m0 + (wm * RM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float4 q4 = v[k / 4];
        float o = (k % 4 == 0) ? q4.x : (k % 4 == 1) ? q4.y : (k % 4 == 2) ? q4.z : q4.w;
        if (a.bias) o += a.bias[co];
        if (rn_) o += rn_[(long)co * OHW + pix];
        yn[(long)co * OHW + pix] = apply_act(o, a.act, slope);
    }
}

// Staging is by `buffer_load ... offen lds` through buffer descriptors (not `global_load_lds` with per-lane pointers):
// the per-chunk advance is the instruction's scalar offset (no VALU), and a lane whose tap reads padding (or has no
// element) carries the offset 0x80000000, which the descriptor's range check turns into a zero written to LDS (checked
// on the hardware: tools/hwprobe/buffer_lds_oob.hip) — no exec-mask branch per load, no zeroing of padding cells at tile
// switches: the staging of a unit is ~35 straight-line instructions.
// Measured and dropped (profiles/r02_conv_experiments.md): two accumulator chains per tile, operands double-buffered in
// registers, double-length LDS chunks — none moved the time once the staging was this cheap; what is left above the MFMA
// time is per-launch cost (ramp, first DMA, last flush, fixup), not the unit loop.
template <int WM, int WN, int RM, int RN, int TW, int KS, int DIL>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_sk_kernel(ConvSkArgs s) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the buffer-descriptor type and builtins exist in the device pass only; the host
                                      // pass needs nothing but the signature for its launch stub)
    const ConvKArgs& a = s.k;
    constexpr int NT = 64 * WM * WN;
    constexpr int MT = 32 * WM * RM;
    constexpr int RPT = 32 / TW;
    constexpr int PH = WN * RN * RPT;
    constexpr int KK = KS * KS;
    constexpr int CK = conv_ck(KS, RM * RN, false);
    constexpr int ROW4 = MT / 4;
    constexpr int WPT = (CK * KK * ROW4 + NT - 1) / NT;
    constexpr int IH_T = PH + DIL * (KS - 1);
    constexpr int IW_T = TW + DIL * (KS - 1);
    constexpr int IW_P = conv_pitch(TW, IW_T, 1, true);
    constexpr int plane = IH_T * IW_P;
    constexpr int EPT = (CK * plane + NT - 1) / NT;
    // every lane of every staging load writes its LDS slot (padding lanes write zeros): the patch buffer covers all slots
    constexpr int C_XS = EPT * NT;
    constexpr int C_WS = conv_ws_floats(CK, KK, MT);
    constexpr int NACC = RM * RN * 16;
    constexpr int OOB = (int)0x80000000;
    // four distinct static arrays (see conv_kernel.h: alias analysis needs them distinct)
    __shared__ __attribute__((aligned(16))) float xsb0[C_XS];
    __shared__ __attribute__((aligned(16))) float xsb1[C_XS];
    __shared__ __attribute__((aligned(16))) float wsb0[C_WS];
    __shared__ __attribute__((aligned(16))) float wsb1[C_WS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const long G = gridDim.x;
    const long wlog = sk_logical_wg(blockIdx.x, G);
    long u = sk_unit_start(wlog, s.U, G);
    const long u1 = sk_unit_start(wlog + 1, s.U, G);
    if (u >= u1) return;
    const int HWi = a.H * a.W;
    const int NC = s.NC;

    // ---- tile-independent staging plan
    // patch element e = tid + t*NT of the pitched image [CK][IH_T][IW_P] (LDS float offset == e):
    //   ecy[t] = (c << 20) | (iy << 10) | ix, or -1 when the element does not exist
    // weight float4 q = tid + i*NT of the [CK*KK][MT] slice: wrel[i] = row*Cout + col, or -1
    int ecy[EPT], gofs[EPT], wrel[WPT];
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const int e = tid + t * NT;
        const int c = e / plane, rem = e - c * plane;
        const int iy = rem / IW_P, ix = rem - iy * IW_P;
        ecy[t] = (e < CK * plane && ix < IW_T) ? ((c << 20) | (iy << 10) | ix) : -1;
    }
    constexpr int nq = CK * KK * ROW4;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int q = tid + i * NT;
        wrel[i] = q < nq ? (q / ROW4) * a.Cout + (q % ROW4) * 4 : -1;
        wrel[i] = wrel[i] >= 0 ? wrel[i] * 4 : OOB;      // byte offset into the weight buffer
    }
    // buffer descriptors: the whole weight tensor; one image of the input (rebuilt when the staged image changes)
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cin * KK * a.Cout * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.Cin * HWi * 4, 0x00020000);
    int rs_n = 0;
    // DMA address plan of one tile: gofs[t] = offset from the chunk's first channel plane, -1 = reads zero
    auto plan = [&](const SkTile& t_) {
        const int vy0 = t_.oy0 - a.pad, vx0 = t_.ox0 - a.pad;
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            int g = -1;
            if (ecy[t] >= 0) {
                const int c = ecy[t] >> 20, iy = (ecy[t] >> 10) & 1023, ix = ecy[t] & 1023;
                const int o = stored_offset(a, vy0 + iy, vx0 + ix);
                g = o >= 0 ? c * HWi + o : -1;
            }
            gofs[t] = g >= 0 ? g * 4 : OOB;
        }
        if (t_.n != rs_n) {
            rs_n = t_.n;
            rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long)t_.n * a.x_bs), 0, a.Cin * HWi * 4, 0x00020000);
        }
    };
    auto issue = [&](int c, float* xs, float* ws, int m0) {
        const int sx = c * CK * HWi * 4, sw = (c * CK * KK * a.Cout + m0) * 4;
#pragma unroll
        for (int t = 0; t < EPT; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (CONV_AS3 void*)(xs + t * NT + wave * 64), 4, gofs[t], sx, 0, 0);
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (CONV_AS3 void*)(ws + (i * NT + wave * 64) * 4), 16, wrel[i], sw, 0, 0);
    };

    // ---- accumulators: tot = running sum of the current tile segment, acc = one chunk's chain
    f32x16 tot[RM][RN], acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
    int xoff[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int t = wn * RN + j;
        xoff[j] = hi * plane + (t * RPT + l31 / TW) * IW_P + l31 % TW;
    }
    const int woff = hi * KK * MT + wm * RM * 32 + l31;
    const float slope = a.act_slope_ptr ? *a.act_slope_ptr : a.act_slope;
    const long OHW = (long)a.OH * a.OW;

    // flush the finished segment of tile `t_`: y (full tile) or this workgroup's slot (partial)
    auto flush = [&](const SkTile& t_, bool full, int slot) {
        if (!full) {
            float4* pp = reinterpret_cast<float4*>(s.part) + ((long)(wlog * 2 + slot) * (NACC / 4)) * NT + tid;
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        pp[(long)((i * RN + j) * 4 + r4) * NT] = make_float4(tot[i][j][r4 * 4 + 0], tot[i][j][r4 * 4 + 1],
                                                                              tot[i][j][r4 * 4 + 2], tot[i][j][r4 * 4 + 3]);
            return;
        }
        float* yn = a.y + (long)t_.n * a.y_bs;
        const float* rn_ = a.res ? a.res + (long)t_.n * a.res_bs : nullptr;
        const int pr = l31 / TW, pc = l31 % TW;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int oy = t_.oy0 + (wn * RN + j) * RPT + pr, ox = t_.ox0 + pc;
            if (oy >= a.OH || ox >= a.OW) continue;
            const long pix = (long)oy * a.OW + ox;
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = t_.m0 + (wm * RM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float v = tot[i][j][r];
                    if (a.bias) v += a.bias[co];
                    if (rn_) v += rn_[(long)co * OHW + pix];
                    yn[(long)co * OHW + pix] = apply_act(v, a.act, slope);
                }
        }
    };

    // ---- prologue: stage the first unit
    int tile = (int)(u / NC), c = (int)(u - (long)tile * NC);
    const int first_tile = tile;
    int seg_c0 = c;                      // first chunk of the current segment
    SkTile cur = sk_decode(s, tile, MT, TW, PH);
    SkTile stg = cur;                    // tile of the unit being staged
    int stg_tile = tile, stg_c = c;
    plan(cur);
    issue(c, xsb0, wsb0, cur.m0);
    __syncthreads();   // (drains the staging loads: vmcnt(0) before the barrier)

    // One unit: stage unit u+1 into the other buffer, run unit u's MFMA chain, barrier, flush on a tile
    // boundary.  Everything that decides control flow is wave-uniform (derived from blockIdx).
    auto step = [&](const float* xs, const float* ws, float* xs_next, float* ws_next) {
        const bool has_next = u + 1 < u1;
        const bool tile_done = c + 1 == NC;
        if (has_next) {
            if (tile_done) {
                stg_tile = tile + 1;
                stg_c = 0;
                stg = sk_decode(s, stg_tile, MT, TW, PH);
                plan(stg);
            } else {
                stg_c = c + 1;
            }
            issue(stg_c, xs_next, ws_next, stg.m0);
        }
        const float* wb = ws + woff;
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int toff = (tap / KS) * DIL * IW_P + (tap % KS) * DIL;
#pragma unroll
            for (int kk = 0; kk < CK; kk += 2) {
                float av[RM], bv[RN];
#pragma unroll
                for (int i = 0; i < RM; ++i) av[i] = wb[(kk * KK + tap) * MT + i * 32];
#pragma unroll
                for (int j = 0; j < RN; ++j) bv[j] = xs[xoff[j] + kk * plane + toff];
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < KK * (CK / 2) * RM * RN; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
        }
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) tot[i][j] += acc[i][j];
        __syncthreads();  // unit u+1 landed in LDS; everyone is done reading unit u
        if (tile_done || !has_next) {
            flush(cur, seg_c0 == 0 && tile_done, tile == first_tile ? 0 : 1);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
            seg_c0 = 0;
        }
        if (tile_done) {
            cur = stg;
            tile = stg_tile;
        }
        c = stg_c;
        ++u;
    };
    while (u < u1) {
        step(xsb0, wsb0, xsb1, wsb1);
        if (u < u1) step(xsb1, wsb1, xsb0, wsb0);
    }
#endif
}

// Second stage: every tile that was split over several workgroups = sum of their slots in ascending-K order
// (workgroup order) + bias / skip / activation.  One workgroup per tile, same thread <-> element map as above.
// Second stage: one launch over the tiles; each split tile is handled by SK_FIX_SPLIT workgroups (blockIdx.y), a quarter of
// the accumulator registers each (this kernel is pure latency).  An in-kernel form of this step — the last contributor of a
// tile to finish adds the slots up, found through per-tile arrival counters, write-through slot stores and one agent-scope
// acquire (no spinning) — was built, verified bit-identical under load, and measured SLOWER on the 6-GFLOP layers (69.0 vs
// 63.8 us at 256->256 54x96, 66.2 vs 60.7 at 512->512 27x48; profiles/r02_conv_inkernel_fixup_probe.txt): the drain +
// atomic + acquire + dependent loads at the end of every workgroup cost more than this 5 us launch.  Removed.
#define SK_FIX_SPLIT 4
template <int WM, int WN, int RM, int RN, int TW>
__global__ __launch_bounds__(64 * WM * WN) void conv_sk_fixup_kernel(ConvSkArgs s, long G) {
    constexpr int NACC = RM * RN * 16;
    constexpr int KPB = NACC / SK_FIX_SPLIT;
    const int tile = blockIdx.x;
    const long uf = (long)tile * s.NC, ul = uf + s.NC - 1;
    if (sk_unit_owner(uf, s.U, G) == sk_unit_owner(ul, s.U, G)) return;   // computed by one workgroup: already in y
    switch (blockIdx.y) {
        case 0: sk_fix_tile<WM, WN, RM, RN, TW, 0 * KPB, KPB>(s, G, tile, threadIdx.x); break;
        case 1: sk_fix_tile<WM, WN, RM, RN, TW, 1 * KPB, KPB>(s, G, tile, threadIdx.x); break;
        case 2: sk_fix_tile<WM, WN, RM, RN, TW, 2 * KPB, KPB>(s, G, tile, threadIdx.x); break;
        default: sk_fix_tile<WM, WN, RM, RN, TW, 3 * KPB, KPB>(s, G, tile, threadIdx.x); break;
    }
}

// ---- per-variant launchers (one translation unit each)
struct ConvSkLaunch {
    dim3 grid, fix_grid;
    bool need_fixup;
};

template <int KS, int DIL, int WM, int WN, int RM, int RN, int TW>
static void conv_sk_launch_one(const ConvSkLaunch& L, hipStream_t st, const ConvSkArgs& s) {
    constexpr int NT = 64 * WM * WN;
    hipLaunchKernelGGL((conv_sk_kernel<WM, WN, RM, RN, TW, KS, DIL>), L.grid, dim3(NT), 0, st, s);
    if (L.need_fixup)
        hipLaunchKernelGGL((conv_sk_fixup_kernel<WM, WN, RM, RN, TW>), L.fix_grid, dim3(NT), 0, st, s, (long)L.grid.x);
}

template <int KS, int DIL, int WM, int WN, int RM, int RN>
static void conv_sk_launch_tw(int tw, const ConvSkLaunch& L, hipStream_t st, const ConvSkArgs& s) {
    switch (tw) {
        case 32: conv_sk_launch_one<KS, DIL, WM, WN, RM, RN, 32>(L, st, s); break;
        case 16: conv_sk_launch_one<KS, DIL, WM, WN, RM, RN, 16>(L, st, s); break;
        default: conv_sk_launch_one<KS, DIL, WM, WN, RM, RN, 8>(L, st, s); break;
    }
}

// tile configurations as in conv_kernel.h (kConvCfgs): 2 = 64 co x 4 N-tiles, 3 = 32 co x 4 N-tiles, 4 = 64 co x 2 N-tiles
template <int KS, int DIL>
static void conv_sk_launch_variant(int cfg, int tw, const ConvSkLaunch& L, hipStream_t st, const ConvSkArgs& s) {
    switch (cfg) {
        case 2: conv_sk_launch_tw<KS, DIL, 1, 4, 2, 1>(tw, L, st, s); break;
        case 3: conv_sk_launch_tw<KS, DIL, 1, 4, 1, 1>(tw, L, st, s); break;
        default: conv_sk_launch_tw<KS, DIL, 2, 2, 1, 1>(tw, L, st, s); break;
    }
}

void conv_sk_launch_k3d1(int cfg, int tw, const ConvSkLaunch& L, hipStream_t st, const ConvSkArgs& s);
void conv_sk_launch_k3d2(int cfg, int tw, const ConvSkLaunch& L, hipStream_t st, const ConvSkArgs& s);
void conv_sk_launch_k1(int cfg, int tw, const ConvSkLaunch& L, hipStream_t st, const ConvSkArgs& s);
