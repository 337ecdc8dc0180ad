// Instantiations of the conv kernel template: variant k3d1 with LDS-DMA staging (KS, DIL, GEN, DMA = 3, 1, false, true).
#include "conv_kernel.h"

void conv_launch_k3d1_dma(int cfg, int tw, dim3 grid, size_t lds, hipStream_t s, const ConvKArgs& a) {
    conv_launch_variant<3, 1, false, true>(cfg, tw, grid, lds, s, a);
}
