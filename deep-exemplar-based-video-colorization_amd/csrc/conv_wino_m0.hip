// Instantiations of the Winograd conv kernel template: 32 output channels x 32 tiles per workgroup — ONE wave pair (2 waves),
// 4-channel chunks, 37 KB of LDS: four workgroups per CU.  For the layers that otherwise need a split over input channels to
// fill the chip (256 -> 256 at 54 x 96: 328 workgroups of this shape run the whole K in one go: no partial sums, no reduce).
#include "conv_wino_kernel.h"

void conv_wino_launch_m0(int tr, dim3 grid, hipStream_t st, const ConvWinoArgs& s) {
    conv_wino_launch_shape<1, 1, 4>(tr, grid, st, s);
}
