// Instantiations of the direct MFMA convolution for 3x3 stride-1 kernels (own translation unit so the
// template variants compile in parallel).
#include "conv_mfma.h"

namespace mp {

int launch_conv_k3s1(const ConvKParams& p, int variant, size_t lds_bytes, hipStream_t s) {
    return launch_ks<3, 1>(p, variant, lds_bytes, s);
}

}  // namespace mp
