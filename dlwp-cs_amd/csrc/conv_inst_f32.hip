// Instantiation unit of the forward / data-gradient kernel (conv_launch.h): exact-fp32 kernels.
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
int dispatch_conv_f32(int KS, int vw, const ConvKParams &P, const Work &W, hipStream_t s) { return dispatch_conv_t<float>(KS, vw, P, W, s); }

}  // namespace dlwpcs
