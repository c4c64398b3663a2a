// Instantiations of the direct MFMA convolution for 2x2 kernels (own translation unit so the
// template variants compile in parallel).
#include "conv_mfma.h"

namespace mp {

int launch_conv_k2(const ConvKParams& p, int stride, int variant, size_t lds_bytes, hipStream_t s) {
    if (stride == 1) return launch_ks<2, 1>(p, variant, lds_bytes, s);
    return MP_ERR_UNSUPPORTED;
}

}  // namespace mp
