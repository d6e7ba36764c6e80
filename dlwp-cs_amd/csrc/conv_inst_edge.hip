// Instantiation unit of the forward / data-gradient kernel (conv_launch.h): the bf16 data gradient in gather form (EDGE).
#include "conv_launch.h"

namespace dlwpcs {

int dispatch_conv_edge(const ConvKParams &P, const Work &W, hipStream_t s) {
    if (P.m0 || P.m1) return launch_conv<bf16_t, 3, 8, MODE_HALO, false, true, true>(P, W, s);
    return launch_conv<bf16_t, 3, 8, MODE_HALO, false, false, true>(P, W, s);
}

}  // namespace dlwpcs
