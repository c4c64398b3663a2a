// fp32 first conv of the network as a streaming kernel (stem_f32.hip): launch record shared with the plan replay (conv_api.hip).
#pragma once
#include "common.h"

namespace mp {

struct StemF32Params {
    const float* x;      // [N][3][H][W]
    const float* w;      // [64][3][3][3]
    const float* scale;  // [64]
    const float* shift;
    float* out;          // [N][64][H/2][W/2]
    int N, H, W, Ho, Wo, pitch, tiles_y, total_blocks, relu;
    unsigned magic_upr;  // / (W / 4)
};
struct StemF32Launch {
    StemF32Params p;
    size_t lds_bytes;
};
int stemf32_build(const float* x, const float* w, const float* scale, const float* shift, int relu, float* out, int n, int h, int wd,
                  StemF32Launch& L);
int stemf32_launch(const StemF32Launch& L, hipStream_t s);

}  // namespace mp
