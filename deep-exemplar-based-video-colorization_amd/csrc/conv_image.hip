// The two convolutions that read an IMAGE: VGG19 conv1_1 (3 -> 64, /root/reference/models/NonlocalNet.py:235, with
// vgg_preprocess folded in as a per-channel affine) and ColorVidNet conv1_1[0] (7 -> 32, models/ColorVidNet.py:98), both 3x3 /
// stride 1 / pad 1 at full frame size.
//
// Why a kernel of their own: the general engine (conv_kernel.h) walks input channels in LDS chunks of 32 per filter tap — with
// 3 (7) channels that is 9 K-steps of 32 of which 3 (7) carry data, i.e. 10.7x (4.6x) the MFMAs the layer needs: 34.7 us and
// 21.5 us for 0.29 / 0.33 GFLOP (profiles/r03_conv_algo_sweep.txt) where the output write alone (21 / 10.6 MB) is ~5 / ~3 us of
// HBM time.  Here the reduction index is k = ci * 9 + ky * 3 + kx itself: K = 27 (63) padded to 28 (64), 14 (32) steps of
// v_mfma_f32_32x32x2_f32 per (32 output channels x 32 pixels) block — the same exact-fp32 products and fp32 accumulation as
// everywhere else on the path, in one chain over k.
//
// Workgroup = 4 waves = an 8 x 32 pixel tile of one image, all output channels.  The (8 + 2) x (32 + 2) x Cin input patch is
// staged in LDS by ordinary loads (padding zeros / reflection and the input affine applied on the way: padding is zero AFTER the
// affine, as in the reference where the affine precedes the convolution's own padding); a lane is pixel (lane & 31) of its
// wave's current row and supplies the B operand of k = 2 step + (lane >> 5) with one ds_read; the filter values a lane
// supplies as A operand (channel lane & 31 of each 32-channel block, same k) are the same for every pixel block, so each wave
// loads them ONCE into registers: w_packed is [Cin][9][Cout] = [k][Cout], the lane's values are a strided column of it.
#include <type_traits>

#include "conv_kernel.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CIN, int NBLK>
__global__ __launch_bounds__(256) void conv_image_kernel(ConvKArgs a, int tiles_x) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int K = CIN * 9, KSTEPS = (K + 1) / 2;
    constexpr int TH = 8, TW = 32, PR = TH + 2, PITCH = TW + 2 + 1, PLANE = PR * PITCH;      // (odd pitch: the two k of a step hit different banks)
    __shared__ float patch[CIN * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int n = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const int Cout = NBLK * 32;

    // the lane's filter column(s): issued first, they arrive while the patch is staged
    float wreg[NBLK][KSTEPS];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int k = 2 * s + hi;
            wreg[b][s] = k < K ? a.w[k * Cout + b * 32 + l31] : 0.f;
        }

    // the lane's 16 output channels per block are (r & 3) + 8 (r >> 2) + 4 hi: their bias values too are loaded once
    float breg[NBLK][16];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) breg[b][r] = a.bias ? a.bias[b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] : 0.f;

    // patch staging: the index arithmetic of a patch position is shared by the CIN planes, all their loads are in flight together
    const float* xn = a.x + (long)n * a.x_bs;
    const int HW = a.H * a.W;
    float sc[CIN], sh[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
        sc[c] = a.in_scale ? a.in_scale[n * CIN + c] : 1.f;      // (x * 1 + 0 == x exactly)
        sh[c] = a.in_scale ? a.in_shift[n * CIN + c] : 0.f;
    }
    constexpr int PE = PR * (TW + 2);
#pragma unroll
    for (int it = 0; it < (PE + 255) / 256; ++it) {
        const int e = tid + it * 256;
        const int iy = e / (TW + 2), ix = e - iy * (TW + 2);
        const int o = e < PE ? stored_offset(a, ty0 - 1 + iy, tx0 - 1 + ix) : -1;
        float v[CIN];
        if (a.gray) {
            // DVC_CONV_GRAY_INPUT: every virtual channel is gray2rgb_batch's value of the ONE stored plane, with that kernel's
            // own expression (csrc/color.hip: bit-identical to running it first)
            const float g = (xn[max(o, 0)] * 1.0f + 50.0f) / 100.0f;
#pragma unroll
            for (int c = 0; c < CIN; ++c) v[c] = g;
        } else {
#pragma unroll
            for (int c = 0; c < CIN; ++c) v[c] = xn[c * HW + max(o, 0)];
        }
        if (e < PE) {
#pragma unroll
            for (int c = 0; c < CIN; ++c) patch[c * PLANE + iy * PITCH + ix] = o >= 0 ? v[c] * sc[c] + sh[c] : 0.f;
        }
    }
    __syncthreads();

    const float slope = a.act_slope_ptr ? *a.act_slope_ptr : a.act_slope;
    const int OHW = a.OH * a.OW;                            // (the launcher checks Cout * OH * OW < 2^31)
    float* yn = a.y + (long)n * a.y_bs;
    const int ox = tx0 + l31;
    // activation without control flow in the store loop: v > 0 ? v : v * neg (ReLU: neg = 0 and the product is replaced by 0;
    // none: neg = 1); tanh (nothing on the path uses it here) takes the general function
    const bool relu = a.act == DVC_ACT_RELU, general_act = a.act == DVC_ACT_TANH128;
    const float neg = (a.act == DVC_ACT_PRELU || a.act == DVC_ACT_LEAKY) ? slope : 1.f;
#pragma unroll
    for (int rr = 0; rr < TH / 4; ++rr) {
        const int row = wave + 4 * rr;                      // the wave's pixel block: row `row` of the tile, 32 pixels
        const float* pb = patch + row * PITCH + l31;
        f32x16 acc[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            // k = 2 s + hi -> (ci, ky, kx); a padded k reads element 0 of the patch against a zero filter value
            int off[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 2 * s + h;
                off[h] = k < K ? (k / 9) * PLANE + ((k % 9) / 3) * PITCH + (k % 3) : 0;
            }
            const float bv = pb[hi ? off[1] : off[0]];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[b][s], bv, acc[b], 0, 0, 0);
        }
        const int oy = ty0 + row;
        if (oy < a.OH && ox < a.OW) {
            float* yp = yn + (4 * hi) * OHW + oy * a.OW + ox;
            auto store_all = [&](auto GENERAL) {
#pragma unroll
                for (int b = 0; b < NBLK; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[b][r] + breg[b][r];
                        if (decltype(GENERAL)::value) v = apply_act(v, a.act, slope);
                        else v = v > 0.f ? v : (relu ? 0.f : v * neg);
                        yp[(b * 32 + (r & 3) + 8 * (r >> 2)) * OHW] = v;
                    }
            };
            if (general_act) store_all(std::true_type{});
            else store_all(std::false_type{});
        }
    }
#endif
}

// (3 -> 64) and (7 -> 32): the two image-input layers of the path.  Returns false when the layer is not one of them.
bool conv_image_launch(const ConvKArgs& a, hipStream_t st) {
    if (a.ks != 3 || a.stride != 1 || a.dil != 1 || a.pad != 1 || a.in_up != 1 || a.in_sub != 1 || a.in_prelu || a.res) return false;
    if (a.gray && !(a.Cin == 3 && a.Cout == 64)) return false;
    const int tiles_x = (a.OW + 31) / 32, tiles_y = (a.OH + 7) / 8;
    if ((long)tiles_x * tiles_y >= (1L << 31) || a.N > 65535 || (long)a.Cout * a.OH * a.OW >= (1L << 31)) return false;
    const dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)a.N);
    if (a.Cin == 3 && a.Cout == 64)
        hipLaunchKernelGGL((conv_image_kernel<3, 2>), grid, dim3(256), 0, st, a, tiles_x);
    else if (a.Cin == 7 && a.Cout == 32)
        hipLaunchKernelGGL((conv_image_kernel<7, 1>), grid, dim3(256), 0, st, a, tiles_x);
    else
        return false;
    return true;
}
