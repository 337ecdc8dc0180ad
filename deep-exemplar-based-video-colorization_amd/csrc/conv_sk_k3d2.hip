// Instantiations of the stream-K conv kernel template: variant k3d2 (KS, DIL = 3, 2).
#include "conv_sk_kernel.h"

void conv_sk_launch_k3d2(int cfg, int tw, const ConvSkLaunch& L, hipStream_t st, const ConvSkArgs& s) {
    conv_sk_launch_variant<3, 2>(cfg, tw, L, st, s);
}
