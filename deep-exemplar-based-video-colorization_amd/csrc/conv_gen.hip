// Instantiations of the conv kernel template: variant gen (KS, DIL, GEN = 3, 1, true).
#include "conv_kernel.h"

void conv_launch_gen(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a) {
    conv_launch_variant<3, 1, true>(cfg, tw, grid, lds, s, a);
}
