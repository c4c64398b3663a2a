// Instantiations of the direct MFMA convolution for 7x7 kernels (own translation unit so the
// template variants compile in parallel).
#include "conv_mfma.h"

namespace mp {

int launch_conv_k7(const ConvKParams& p, int stride, int variant, size_t lds_bytes, hipStream_t s) {
    if (stride == 2) return launch_ks<7, 2>(p, variant, lds_bytes, s);
    return MP_ERR_UNSUPPORTED;
}

}  // namespace mp
