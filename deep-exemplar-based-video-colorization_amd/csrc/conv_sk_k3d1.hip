// Instantiations of the stream-K conv kernel template: variant k3d1 (KS, DIL = 3, 1).
#include "conv_sk_kernel.h"

void conv_sk_launch_k3d1(int cfg, int tw, const ConvSkLaunch& L, hipStream_t st, const ConvSkArgs& s) {
    conv_sk_launch_variant<3, 1>(cfg, tw, L, st, s);
}
