// Hardware probe (MI355X / gfx950): what does a bare v_mfma_f32_32x32x2_f32 loop sustain chip-wide, as a function of
// (a) waves per SIMD, (b) dependent chain vs two alternating accumulators, (c) a drain + 16 adds every 36 MFMAs (the
// per-chunk flush of the convolution engine), (d) launch length (1440 MFMAs per wave = one 6-GFLOP layer, and 10x that).
//   build: hipcc --offload-arch=gfx950 -O2 mfma_f32_chain.hip -o mfma_f32_chain ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool FLUSH>
__global__ __launch_bounds__(256, 2) void chain(const float* in, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float a = in[lane], b = in[64 + lane];
    f32x16 acc[NACC], tot;
    for (int r = 0; r < 16; ++r) tot[r] = 0.f;
    for (int p = 0; p < NACC; ++p)
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (FLUSH)
            for (int p = 0; p < NACC; ++p)
                for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
#pragma unroll
        for (int m = 0; m < 36; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m % NACC], 0, 0, 0);
        if (FLUSH) {
            for (int p = 0; p < NACC; ++p) tot += acc[p];
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    if (!FLUSH)
        for (int p = 0; p < NACC; ++p) tot += acc[p];
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += tot[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool FLUSH>
static void run(const char* name, int wgs, int iters, const float* in, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((chain<NACC, FLUSH>), dim3(wgs), dim3(256), 0, 0, in, out, iters);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((chain<NACC, FLUSH>), dim3(wgs), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us = best * 100.0;   // per launch
    const double flop = (double)wgs * 4 * iters * 36 * 4096.0;
    printf("%-44s %4d WGs x %4d iters: %8.1f us/launch  %6.1f TFLOP/s (%.1f %% of 157.3)\n", name, wgs, iters, us, flop / us * 1e-6,
           flop / us * 1e-6 / 157.3 * 100);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 128 * sizeof(float));
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    std::vector<float> h(128);
    for (int i = 0; i < 128; ++i) h[i] = 0.37f + 0.013f * (float)((i * 7919) % 97) - 0.6f;
    hipMemcpy(in, h.data(), 128 * sizeof(float), hipMemcpyHostToDevice);
    for (int len : {40, 400}) {
        run<1, false>("1 wave/SIMD, one chain, no flush", 256, len, in, out);
        run<1, true>("1 wave/SIMD, one chain, flush every 36", 256, len, in, out);
        run<2, false>("1 wave/SIMD, two chains, no flush", 256, len, in, out);
        run<2, true>("1 wave/SIMD, two chains, flush every 36", 256, len, in, out);
        run<1, false>("2 waves/SIMD, one chain, no flush", 512, len / 2, in, out);
        run<1, true>("2 waves/SIMD, one chain, flush every 36", 512, len / 2, in, out);
        run<2, true>("2 waves/SIMD, two chains, flush every 36", 512, len / 2, in, out);
    }
    // zero operands: the clock the chip allows itself when the multipliers toggle nothing
    hipMemset(in, 0, 128 * sizeof(float));
    run<1, false>("1 wave/SIMD, one chain, ZERO operands", 256, 400, in, out);
    run<1, true>("2 waves/SIMD, flush, ZERO operands", 512, 200, in, out);
    return 0;
}
