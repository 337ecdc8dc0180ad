// Instantiations of the conv kernel template: variant k3d1 (KS, DIL, GEN = 3, 1, false).
#include "conv_kernel.h"

void conv_launch_k3d1(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a) {
    conv_launch_variant<3, 1, false>(cfg, tw, grid, lds, s, a);
}
