// Weights-in-registers direct convolution (r06): 3x3, stride 1, pad 1 (zero), Cin = 32 / 64 / 128, on v_mfma_f32_32x32x2_f32.
//
// Why: the layers of ColorVidNet's encoder that must stay on the direct engine (arch.DIRECT_LAYERS: Winograd's rounding there
// puts the whole path above the reference's own fp32 error) are large in space and small in channels (32 -> 64 and 64 -> 64 at
// 216x384, 64 -> 128 at 108x192).  The general engine (conv_kernel.h) streams a workgroup's filter slice through LDS once per
// 64-pixel tile — for 64 -> 64 at 216x384 that is the WHOLE 147 KB filter set 1296 times, one barrier and one LDS round per
// 8-channel chunk — and sits at 0.55 of the fp32 MFMA peak there.  Here the roles are those of the correlation kernel
// (corr.hip): the operand that is reused lives in REGISTERS for the workgroup's whole life, the other one streams through LDS.
//   * a wave holds the filters of 32 output channels x 32 input channels x 9 taps as 144 MFMA A fragments
//     (A[i = lane & 31][k = lane >> 5]: one VGPR per MFMA), loaded once (36 coalesced 16-byte loads per lane from a copy packed
//     in fragment order, dvc_conv2d_ws_pack_weight);
//   * a workgroup owns a 32-pixel-wide column strip and walks down its rows: the staged input is a RING of rows
//     [slot][channel][34 floats] filled by buffer-descriptor LDS-DMA one step ahead (out-of-range lanes write the padding
//     zeros), so every input row is fetched once per strip — no halo re-reads along y, 34/32 along x;
//   * a row's 32 pixels x 32 output channels are a chain of 144 MFMAs per wave (B fragment = one ds_read_b32 with an immediate
//     offset per MFMA, two pairs ahead in fixed registers: conv_ws_chain.inc), restarted from zero every 36 (8 channels x 9
//     taps) and added to a running total — the direct engine's blocked summation, the same chain lengths;
//   * Cin = 64 (128): two (four) waves split the input channels (KH = 2, 4) and combine their totals through LDS once per row;
//     Cin = 32: one wave holds all of K and the workgroup's second wave pair takes every other row (PS = 2);
//   * ONE barrier per row step; no filter traffic after the prologue; 2 workgroups per CU (<= 256 VGPRs).
// fp32 throughout: exact products, fp32 accumulation, chain lengths 72 + a short tree — the direct engine's error class.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#include "conv_ws_chain.inc"

#define WS_AS3 __attribute__((address_space(3)))

struct ConvWsArgs {
    const float* x;
    const float* w;        // [Cout/32][KH][36][64][4]  (dvc_conv2d_ws_pack_weight)
    const float* bias;
    const float* act_slope_ptr;
    float* y;
    int N, H, W, Cout;
    int strips, chunks, rpw;   // column strips of 32 pixels, row chunks per strip, rows per workgroup
    int act;
    float act_slope;
    long x_bs, y_bs;
};

__device__ __forceinline__ float ws_act(float v, int act, float slope) {
    switch (act) {
        case DVC_ACT_RELU: return v > 0.f ? v : 0.f;
        case DVC_ACT_PRELU:
        case DVC_ACT_LEAKY: return v >= 0.f ? v : v * slope;
        default: return v;
    }
}

// KH: waves that split the 32 * KH input channels; CT: 32-channel output tiles per workgroup; PS: rows per step.
// KH * CT * PS = 4 waves (two workgroups per CU) or 8 (Cin = 128: KH = 4, CT = 2 — one workgroup per CU): two waves per SIMD either way.
template <int KH, int CT, int PS>
__global__ __launch_bounds__(64 * KH * CT * PS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_ws_kernel(ConvWsArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(KH * CT * PS == 4 || KH * CT * PS == 8, "four or eight waves");
    constexpr int NT = 64 * KH * CT * PS;
    constexpr int CIN = 32 * KH;
    constexpr int ROWF = 34;                              // staged floats per channel row (32 pixels + halo)
    constexpr int EPT = (CIN * ROWF + NT - 1) / NT;       // DMA instructions per thread and row
    constexpr int SLOT = EPT * NT;                        // floats per ring slot (the tail is padding the DMA zero-fills)
    constexpr int R = 2 * PS + 2;                         // ring: PS + 2 rows in use, PS rows in flight
    constexpr int OOB = (int)0x80000000;
    constexpr int XW = (KH - 1) * CT * PS;                // waves that hand a partial total over
    __shared__ __attribute__((aligned(16))) float ring[R * SLOT];
    __shared__ __attribute__((aligned(16))) float xbuf[XW > 0 ? 2 * XW * 16 * 64 : 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int kh = wave % KH, ct = (wave / KH) % CT, ps = wave / (KH * CT);
    const int strip = blockIdx.x % a.strips, chunk = blockIdx.x / a.strips;
    const int x0 = strip * 32;
    const int y_begin = chunk * a.rpw, y_end = min(a.H, y_begin + a.rpw);
    const int cot = blockIdx.y * CT + ct;                 // 32-channel output tile of this wave
    const int n = blockIdx.z;
    const int HW = a.H * a.W;

    // ---- DMA plan of one input row (the same for every row: the row advance is the instruction's scalar offset)
    int gofs[EPT];
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const int e = t * NT + tid;
        const int c = e / ROWF, col = e - c * ROWF;
        const int gx = x0 - 1 + col;
        gofs[t] = (c < CIN && gx >= 0 && gx < a.W) ? (c * HW + gx) * 4 : OOB;
    }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long)n * a.x_bs), 0, CIN * HW * 4, 0x00020000);
    auto issue_row = [&](int row) {      // input row `row` -> slot row mod R (rows outside the image: zeros)
        const bool ok = row >= 0 && row < a.H;
        const int slot = ((row % R) + R) % R;
        const int so = ok ? row * a.W * 4 : 0;
        float* dst = ring + slot * SLOT;
#pragma unroll
        for (int t = 0; t < EPT; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (WS_AS3 void*)(dst + t * NT + wave * 64), 4, ok ? gofs[t] : OOB, so, 0, 0);
    };
    // this lane's B operand inside a slot: channel kh * 32 + hi of its pixel (tap column 0)
    const unsigned lanepart = (unsigned)(size_t)(WS_AS3 float*)ring + (unsigned)(((kh * 32 + hi) * ROWF + l31) * 4);

    const float slope = a.act_slope_ptr ? *a.act_slope_ptr : a.act_slope;
    const float* biasp = a.bias ? a.bias + cot * 32 + 4 * hi : nullptr;     // (read at store time: 16 VGPRs are worth more)
    float* yn = a.y + (long)n * a.y_bs;

    // rows y_begin - 1 .. y_begin + PS of the first step (issued before the filter loads: the two round trips overlap)
#pragma unroll
    for (int k = -1; k <= PS; ++k) issue_row(y_begin + k);
    asm volatile("" ::: "memory");      // (the filter loads below must stay BEHIND the row DMAs: the vmcnt(36) further down counts on it)
    // ---- filters: 144 A fragments, 36 x 16 bytes per lane
    float A[144];
    {
        const f32x4v* wp = reinterpret_cast<const f32x4v*>(a.w) + ((long)(cot * KH + kh) * 36) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            const f32x4v v = wp[i * 64];
            A[4 * i + 0] = v.x; A[4 * i + 1] = v.y; A[4 * i + 2] = v.z; A[4 * i + 3] = v.w;
        }
    }
    // the rows were issued BEFORE the 36 filter loads: wait for them only (memory operations complete in order per counter) —
    // the filters keep arriving under the barrier and the first chain blocks (the compiler waits for each fragment at its use)
    asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int par = 0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int y = y_begin; y < y_end; y += PS) {
        // the next step's new rows (their slots held rows y - 1 - PS .. y - 2: nobody reads those any more)
#pragma unroll
        for (int k = 0; k < PS; ++k) issue_row(y + PS + 1 + k);
        const int yy = y + ps;                            // this wave's output row
        unsigned b0, b1, b2;
        {
            const int s0 = (((yy - 1) % R) + R) % R, s1 = (yy % R), s2 = ((yy + 1) % R);
            b0 = lanepart + (unsigned)(s0 * SLOT * 4);
            b1 = lanepart + (unsigned)(s1 * SLOT * 4);
            b2 = lanepart + (unsigned)(s2 * SLOT * 4);
        }
        f32x16 tot;
        float fa0, fa1, fb0, fb1;
        CONV_WS_PREFETCH(fa0, fa1, fb0, fb1, b0, b1, b2);
        CONV_WS_BLOCK_0(acc, fa0, fa1, fb0, fb1, b0, b1, b2, A);
        CONV_WS_BLOCK_1(acc, fa0, fa1, fb0, fb1, b0, b1, b2, A);
        tot = acc;
        CONV_WS_BLOCK_2(acc, fa0, fa1, fb0, fb1, b0, b1, b2, A);
        CONV_WS_BLOCK_3(acc, fa0, fa1, fb0, fb1, b0, b1, b2, A);
        tot += acc;
        CONV_WS_BLOCK_4(acc, fa0, fa1, fb0, fb1, b0, b1, b2, A);
        CONV_WS_BLOCK_5(acc, fa0, fa1, fb0, fb1, b0, b1, b2, A);
        tot += acc;
        CONV_WS_BLOCK_6(acc, fa0, fa1, fb0, fb1, b0, b1, b2, A);
        CONV_WS_BLOCK_7(acc, fa0, fa1, fb0, fb1, b0, b1, b2, A);
        tot += acc;
        // partial totals of the upper channel halves -> LDS (two buffers: the reader of step y may still be at it at y + PS).
        // (Rotating the finishing wave with the step, so that the epilogues spread over the SIMDs, was measured: 65 -> 68 us.)
        if (XW > 0 && kh > 0) {
            float* mine = xbuf + ((par * XW + ((kh - 1) * CT + ct) * PS + ps) * 16) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r * 64] = tot[r];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the next rows has landed
        __syncthreads();
        if (kh == 0 && yy < y_end) {
#pragma unroll
            for (int k = 1; k < KH; ++k) {                     // ascending channel halves: a fixed order
                const float* theirs = xbuf + ((par * XW + ((k - 1) * CT + ct) * PS + ps) * 16) * 64 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[r] += theirs[r * 64];
            }
            const int ox = x0 + l31;
            if (ox < a.W) {
                float* yp = yn + (long)yy * a.W + ox + (long)(cot * 32 + 4 * hi) * HW;
                float bv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bv[r] = 0.f;
                if (biasp) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) bv[r] = biasp[(r & 3) + 8 * (r >> 2)];
                }
                // (the activation is wave-uniform: one straight store loop per kind, no per-element branches)
                if (a.act == DVC_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = tot[r] + bv[r];
                        yp[(long)((r & 3) + 8 * (r >> 2)) * HW] = v > 0.f ? v : 0.f;
                    }
                } else {
                    const float ns = a.act == DVC_ACT_NONE ? 1.f : slope;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = tot[r] + bv[r];
                        yp[(long)((r & 3) + 8 * (r >> 2)) * HW] = v >= 0.f ? v : v * ns;
                    }
                }
            }
        }
        par ^= 1;
    }
#endif
}

// ------------------------------------------------------------------------------------------------ host side
// filters in fragment order: u[cot][kh][s4 = 9 q + t][lane = 32 hi + i][p] = w[32 cot + i][32 kh + 8 q + 2 p + hi][t]
__global__ __launch_bounds__(256) void conv_ws_pack_kernel(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ u) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)Cout * Cin * 9;
    if (e >= total) return;
    const int p = (int)(e & 3), lane = (int)((e >> 2) & 63);
    long r = e >> 8;
    const int s4 = (int)(r % 36);
    r /= 36;
    const int KH = Cin / 32;
    const int kh = (int)(r % KH), cot = (int)(r / KH);
    const int q = s4 / 9, t = s4 % 9, hi = lane >> 5, i = lane & 31;
    const int co = 32 * cot + i, ci = 32 * kh + 8 * q + 2 * p + hi;
    u[e] = w[((long)co * Cin + ci) * 9 + t];
}

extern "C" int dvc_conv2d_ws_eligible(const DvcConvDesc* d) {
    if (!d) return 0;
    return d->ksize == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1 && d->pad_mode == DVC_PAD_ZERO && d->in_up == 1 &&
           d->in_sub == 1 && !d->in_prelu && (d->Cin == 32 || d->Cin == 64 || d->Cin == 128) && d->Cout % 64 == 0 && d->N > 0 && d->H > 0 &&
           d->W > 0 && (long)d->Cin * d->H * d->W * 4 < (1L << 31) &&
           (d->act == DVC_ACT_NONE || d->act == DVC_ACT_RELU || d->act == DVC_ACT_PRELU || d->act == DVC_ACT_LEAKY);
}

extern "C" int dvc_conv2d_ws_pack_weight(const float* w, int32_t Cout, int32_t Cin, float* u_packed, dvcStream stream) {
    DVC_REQUIRE(w && u_packed, "dvc_conv2d_ws_pack_weight: null argument");
    DVC_REQUIRE(Cout > 0 && Cout % 32 == 0 && (Cin == 32 || Cin == 64 || Cin == 128), "dvc_conv2d_ws_pack_weight: needs Cout %% 32 == 0 and Cin 32, 64 or 128 (got %d, %d)",
                Cout, Cin);
    DVC_REQUIRE((reinterpret_cast<uintptr_t>(u_packed) & 15) == 0, "dvc_conv2d_ws_pack_weight: destination must be 16-byte aligned");
    const long n = (long)Cout * Cin * 9;
    hipLaunchKernelGGL(conv_ws_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, u_packed);
    DVC_CHECK_LAUNCH("dvc_conv2d_ws_pack_weight");
    return 0;
}

static int ws_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return n;
}

extern "C" int dvc_conv2d_ws(const DvcConvDesc* d, const float* x, const float* u_packed, const float* bias, const float* act_slope_ptr,
                             float* y, dvcStream stream) {
    DVC_REQUIRE(d && x && u_packed && y, "dvc_conv2d_ws: null argument");
    DVC_REQUIRE(dvc_conv2d_ws_eligible(d), "dvc_conv2d_ws: needs a 3x3 stride-1 pad-1 zero-padded layer with 32, 64 or 128 input channels, "
                                           "Cout %% 64 == 0 and no fused input transform");
    DVC_REQUIRE((reinterpret_cast<uintptr_t>(u_packed) & 15) == 0, "dvc_conv2d_ws: weights must be 16-byte aligned");
    ConvWsArgs a;
    a.x = x; a.w = u_packed; a.bias = bias; a.act_slope_ptr = act_slope_ptr; a.y = y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
    a.act = d->act; a.act_slope = d->act_slope;
    a.x_bs = d->x_batch_stride ? d->x_batch_stride : (long)d->Cin * d->H * d->W;
    a.y_bs = d->y_batch_stride ? d->y_batch_stride : (long)d->Cout * d->H * d->W;
    a.strips = cdiv(d->W, 32);
    const int ps = d->Cin == 32 ? 2 : 1;
    const int coblk = d->Cout / 64;
    // rows per workgroup: as many workgroups as the chip holds at once (two four-wave ones per CU, or one of eight waves), per
    // image — never a function of the batch
    const int slots = (d->Cin == 128 ? 1 : 2) * ws_num_cus();
    int nchunk = slots / (a.strips * coblk);
    if (nchunk < 1) nchunk = 1;
    int rpw = cdiv(d->H, nchunk);
    rpw = cdiv(rpw, ps) * ps;
    a.rpw = rpw;
    a.chunks = cdiv(d->H, rpw);
    dim3 grid((unsigned)(a.strips * a.chunks), (unsigned)coblk, (unsigned)d->N);
    hipStream_t st = (hipStream_t)stream;
    if (d->Cin == 32) hipLaunchKernelGGL((conv_ws_kernel<1, 2, 2>), grid, dim3(256), 0, st, a);
    else if (d->Cin == 64) hipLaunchKernelGGL((conv_ws_kernel<2, 2, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_ws_kernel<4, 2, 1>), grid, dim3(512), 0, st, a);
    DVC_CHECK_LAUNCH("dvc_conv2d_ws");
    return 0;
}
