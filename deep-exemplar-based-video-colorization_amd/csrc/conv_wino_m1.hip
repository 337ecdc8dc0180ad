// Instantiations of the Winograd conv kernel template: 64 output channels x 32 tiles per workgroup (4 waves), 4-channel chunks,
// 64 KB of LDS: TWO workgroups per CU, each with one wave per SIMD — the prologue, the accumulator flush / exchange / stores
// and the per-chunk barriers of one overlap the K loop of the other.
#include "conv_wino_kernel.h"

void conv_wino_launch_m1(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s) {
    conv_wino_launch_shape<2, 1, 4>(tr, grid, st, s);
}

void conv_wino_launch_m1_dual(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s) {
    conv_wino_launch_shape_dual<2, 1, 4>(tr, grid, st, s);
}
