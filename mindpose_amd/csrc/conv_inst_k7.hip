// Instantiations of the direct MFMA convolution for 7x7 stride-2 kernels (ResNet stem, Cout = 64): only the
// two 192-pixel tile variants are built.
#include "conv_mfma.h"

namespace mp {

int launch_conv_k7(const ConvKParams& p, int stride, int variant, size_t lds_bytes, hipStream_t s) {
    if (stride != 2) return MP_ERR_UNSUPPORTED;
    if (variant == V_CT64_PT192)
        return p.vec ? launch_variant<7, 2, 3, 4, 4, 1, true, false>(p, lds_bytes, s)
                     : launch_variant<7, 2, 3, 4, 4, 1, false, false>(p, lds_bytes, s);
    if (variant == V_CT32_PT192_H)
        return p.vec ? launch_variant<7, 2, 3, 2, 4, 1, true, false>(p, lds_bytes, s)
                     : launch_variant<7, 2, 3, 2, 4, 1, false, false>(p, lds_bytes, s);
    return MP_ERR_UNSUPPORTED;
}

}  // namespace mp
