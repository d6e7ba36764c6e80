// Instantiation unit of the forward / data-gradient kernel (conv_launch.h): bf16 kernels (forward, padded-grid data gradient, TAIL8 input layer).
#include "conv_launch.h"

namespace dlwpcs {

// forward: MODE_HALO / MODE_DIRECT without mask; data gradient: MODE_ZERO (k=3) or MODE_DIRECT (k=1) with/without mask
template <typename T>
static int dispatch_conv_t(int KS, int vw, const ConvKParams &P, const Work &W, hipStream_t s) {
    const bool mask = P.ymask != nullptr;
    if (KS == 3) {
        if (P.mode == MODE_HALO) return dispatch_vw<T, 3, MODE_HALO, false>(vw, P, W, s);
        if (P.mode == MODE_DIRECT) return dispatch_vw<T, 3, MODE_DIRECT, false>(vw, P, W, s);
        if constexpr (sizeof(T) == 2) {
            // pre-masked gradients: the direct-store epilogue multiplies by act'(source) (full 16-B vectors only; other shapes
            // leave the masks to the routing kernels / the caller, see mask_done)
            if (!mask && vw == 8 && (P.m0 || P.m1)) return launch_conv<T, 3, 8, MODE_ZERO, false, true>(P, W, s);
        }
        return mask ? dispatch_vw<T, 3, MODE_ZERO, true>(vw, P, W, s) : dispatch_vw<T, 3, MODE_ZERO, false>(vw, P, W, s);
    }
    return mask ? dispatch_vw<T, 1, MODE_DIRECT, true>(vw, P, W, s) : dispatch_vw<T, 1, MODE_DIRECT, false>(vw, P, W, s);
}
int dispatch_conv_bf16(int KS, int vw, const ConvKParams &P, const Work &W, hipStream_t s) { return dispatch_conv_t<bf16_t>(KS, vw, P, W, s); }

// the network's input layer (14 or 26 channels): 16-B vectors with a shifted tail instead of 4-B loads (conv_ws.h, TAIL8)
int dispatch_conv_tail8(int kc, int NTtot, const ConvKParams &P, const Work &W, hipStream_t s) {
    if (kc == 16) {
        if (NTtot == 1) return launch_conv_cfg<bf16_t, 3, 16, 3, 1, 4, 1, 8, MODE_HALO, false, true>(P, W, s);
        return launch_conv_cfg<bf16_t, 3, 16, 3, 1, 2, 2, 8, MODE_HALO, false, true>(P, W, s);
    }
    if (NTtot == 1) return launch_conv_cfg<bf16_t, 3, 32, 3, 1, 4, 1, 8, MODE_HALO, false, true>(P, W, s);
    return launch_conv_cfg<bf16_t, 3, 32, 3, 1, 2, 2, 8, MODE_HALO, false, true>(P, W, s);
}

}  // namespace dlwpcs
