// The grouped form of the Winograd conv kernel (several independent layers in one launch, conv_wino_kernel.h): 64 output
// channels x 32 tiles per workgroup (4 waves, two workgroups per CU), the shape the planner gives every layer of WarpNet's heads.
#include "conv_wino_kernel.h"

void conv_wino_launch_group_m1(dim3 grid, hipStream_t st, const ConvWinoGroupArgs& g) {
    hipLaunchKernelGGL((conv_wino_group_kernel<2, 1, 4>), grid, dim3(256), 0, st, g);
}
