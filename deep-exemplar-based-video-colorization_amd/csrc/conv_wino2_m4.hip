// Instantiations of the position-split Winograd kernel: 128 output channels x 32 tiles per workgroup (8 waves), 4-channel chunks.
#include "conv_wino2_kernel.h"

void conv_wino2_launch_m4(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s) {
    conv_wino2_launch_shape<4, 1, 4>(tr, grid, st, s);
}
