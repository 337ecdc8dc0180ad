// Instantiations of the Winograd conv kernel template: 128 output channels x 32 tiles per workgroup (8 waves), 4-channel chunks.
#include "conv_wino_kernel.h"

void conv_wino_launch_m4(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s) {
    conv_wino_launch_shape<4, 1, 4>(tr, grid, st, s);
}
