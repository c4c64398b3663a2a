// Instantiations of the direct MFMA convolution for 1x1 kernels (own translation unit so the
// template variants compile in parallel).
#include "conv_mfma.h"

namespace mp {

int launch_conv_k1(const ConvKParams& p, int stride, int variant, size_t lds_bytes, hipStream_t s) {
    if (stride == 1) return launch_ks<1, 1>(p, variant, lds_bytes, s);
    if (stride == 2) return launch_ks<1, 2>(p, variant, lds_bytes, s);
    return MP_ERR_UNSUPPORTED;
}

}  // namespace mp
