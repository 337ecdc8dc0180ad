// Instantiations of the conv kernel template: variant k3d2 (KS, DIL, GEN = 3, 2, false).
#include "conv_kernel.h"

void conv_launch_k3d2(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a) {
    conv_launch_variant<3, 2, false>(cfg, tw, grid, lds, s, a);
}
