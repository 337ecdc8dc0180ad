// Hardware probe (MI355X / gfx950): can fp32 VALU work hide beside a chain of v_mfma_f32_32x32x2_f32?
// The fused correlation (csrc/corr.hip) runs, per key tile, 128 DEPENDENT fp32 MFMAs and then ~60-100 VALU instructions of
// online softmax on the chain's result.  DESIGN.md §4.1 assumes that fp32 MFMA and fp32 VALU share the SIMD's fp32 lanes, so the
// softmax could not be hidden.  This probe measures it: a wave loops over "tiles" of 128 dependent MFMAs with F filler
// v_fma_f32 per tile, either
//   AFTER  the chain, dependent on its result (what the kernel does today), or
//   INSIDE the chain, independent of it (what a software-pipelined kernel would do: softmax of tile t-1 beside the MFMAs of t),
// at 1 and 2 waves per SIMD.  If INSIDE is free and AFTER is not, pipelining the softmax one tile behind pays.
//   build: hipcc --offload-arch=gfx950 -O2 mfma_f32_valu_fill.hip -o mfma_f32_valu_fill ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F, bool INSIDE>     // F fillers per 128-MFMA tile (multiple of 64, or 0)
__global__ __launch_bounds__(256, 2) void tiles(const float* in, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float a = in[lane], b = in[64 + lane], c = 0.999f;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float x0 = a, x1 = b, x2 = a + 1.f, x3 = b + 1.f, x4 = a - 1.f, x5 = b - 1.f, x6 = a * 2.f, x7 = b * 2.f;
#define FILL8()                                                                                                              \
    asm volatile("v_fma_f32 %0, %0, %8, %0\n\tv_fma_f32 %1, %1, %8, %1\n\tv_fma_f32 %2, %2, %8, %2\n\tv_fma_f32 %3, %3, %8, %3\n\t" \
                 "v_fma_f32 %4, %4, %8, %4\n\tv_fma_f32 %5, %5, %8, %5\n\tv_fma_f32 %6, %6, %8, %6\n\tv_fma_f32 %7, %7, %8, %7"   \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)                             \
                 : "v"(c))
#define MFMA16()                                                                                                            \
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t"                        \
                 "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t"                        \
                 "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t"                        \
                 "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t"                        \
                 "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t"                        \
                 "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t"                        \
                 "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t"                        \
                 "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0"                            \
                 : "+v"(acc)                                                                                                \
                 : "v"(a), "v"(b))
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            MFMA16();
            if (INSIDE) {       // F / 8 fillers per 16-MFMA block, independent of the chain
#pragma unroll
                for (int f = 0; f < F / 64; ++f) FILL8();
            }
        }
        if (!INSIDE && F > 0) {
            // the softmax reads the chain's result: wait states of an MFMA result before a VALU read, then the fillers
            asm volatile("s_nop 15\n\ts_nop 7\n\tv_add_f32 %0, %0, %1" : "+v"(x0) : "v"(acc[0]));
#pragma unroll
            for (int f = 0; f < F / 8; ++f) FILL8();
        }
        // (the next tile's chain starts from a fresh accumulator, as in the kernel: one v_mov per register)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = (float)it * 1e-30f;
    }
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int F, bool INSIDE>
static double run(int wgs, int iters, const float* in, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((tiles<F, INSIDE>), dim3(wgs), dim3(256), 0, 0, in, out, iters);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((tiles<F, INSIDE>), dim3(wgs), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 100.0;   // us per launch
}

template <int F>
static void row(int wgs, int iters, const float* in, float* out) {
    const double flop = (double)wgs * 4 * iters * 128 * 4096.0;
    const double t_after = run<F, false>(wgs, iters, in, out), t_inside = run<F, true>(wgs, iters, in, out);
    printf("%4d WGs x %3d tiles, %3d fp32 VALU per 128-MFMA tile: after the chain %7.1f us (%5.1f %% of the MFMA peak) | inside the chain "
           "%7.1f us (%5.1f %%)\n", wgs, iters, F, t_after, flop / t_after * 1e-6 / 157.3 * 100, t_inside, flop / t_inside * 1e-6 / 157.3 * 100);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 128 * sizeof(float));
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    std::vector<float> h(128);
    for (int i = 0; i < 128; ++i) h[i] = 0.37f + 0.013f * (float)((i * 7919) % 97) - 0.6f;
    hipMemcpy(in, h.data(), 128 * sizeof(float), hipMemcpyHostToDevice);
    for (int wgs : {512, 256})          // 2 waves / 1 wave per SIMD
        for (int iters : {13, 104}) {
            row<0>(wgs, iters, in, out);
            row<64>(wgs, iters, in, out);
            row<128>(wgs, iters, in, out);
            row<256>(wgs, iters, in, out);
            row<512>(wgs, iters, in, out);
        }
    return 0;
}
