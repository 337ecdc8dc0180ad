// Instantiations of the stream-K conv kernel template: variant k1 (KS, DIL = 1, 1).
#include "conv_sk_kernel.h"

void conv_sk_launch_k1(int cfg, int tw, const ConvSkLaunch& L, hipStream_t st, const ConvSkArgs& s) {
    conv_sk_launch_variant<1, 1>(cfg, tw, L, st, s);
}
