// Instantiations of the Winograd conv kernel template: 128 output channels x 32 tiles per workgroup, 4-channel chunks.
#include "conv_wino_kernel.h"

void conv_wino_launch_m4(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s) {
    conv_wino_launch_shape<4, 1, 4>(tr, grid, st, s);
}

// timing experiments (dvc_debug_conv_variant >= 4): 1x32-tile blocks with parts of the K-step removed
void conv_wino_launch_m4_dbg(int variant, dim3 grid, hipStream_t st, const ConvWinoArgs& s) {
    switch (variant & ~3) {
        case 4: hipLaunchKernelGGL((conv_wino_kernel<4, 1, 1, 4, 4>), grid, dim3(256), 0, st, s); break;
        case 8: hipLaunchKernelGGL((conv_wino_kernel<4, 1, 1, 4, 8>), grid, dim3(256), 0, st, s); break;
        case 12: hipLaunchKernelGGL((conv_wino_kernel<4, 1, 1, 4, 12>), grid, dim3(256), 0, st, s); break;
        case 16: hipLaunchKernelGGL((conv_wino_kernel<4, 1, 1, 4, 16>), grid, dim3(256), 0, st, s); break;
        default: hipLaunchKernelGGL((conv_wino_kernel<4, 1, 1, 4, 28>), grid, dim3(256), 0, st, s); break;
    }
}
