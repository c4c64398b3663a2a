// Instantiations of the direct MFMA convolution for 3x3 stride-2 kernels (own translation unit so the
// template variants compile in parallel).
#include "conv_mfma.h"

namespace mp {

int launch_conv_k3s2(const ConvKParams& p, int variant, size_t lds_bytes, hipStream_t s) {
    return launch_ks<3, 2>(p, variant, lds_bytes, s);
}

}  // namespace mp
