// Instantiations of the Winograd conv kernel template: 64 output channels x 64 tiles per workgroup (8 waves), 8-channel chunks.
#include "conv_wino_kernel.h"

void conv_wino_launch_m2(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s) {
    conv_wino_launch_shape<2, 2, 8>(tr, grid, st, s);
}
