// fp32 expand + reduce 1x1 chain of HRNet's stage 1 (pwchain_f32.hip): launch record shared with the conv API / plan replay (conv_api.hip).
#pragma once
#include "common.h"

namespace mp {

struct PwChainF32Params {
    const float* mid;     // [N][64][HW]   input of the expand conv (the Bottleneck's 3x3 output)
    const float* res;     // [N][256][HW]  the block's identity, or null:
    const float* x0;      // [N][64][HW]   the block's INPUT - the residual is its down-sample conv, computed in the launch
    const float* wd;      // down-sample weights [64][256], scale_d / shift_d [256] (no ReLU)
    const float* scale_d;
    const float* shift_d;
    const float* w3;      // expand weights, the direct kernel's packing of a 1x1 weight: [64][256] (k-major, cout contiguous)
    const float* scale3;  // [256] folded BatchNorm
    const float* shift3;
    const float* w1;      // reduce weights of the NEXT block: [256][64]; null = the expand conv alone (z null)
    const float* scale1;  // [64]
    const float* shift1;
    float* y;             // [N][256][HW]  relu(conv3(mid) * scale3 + shift3 + res)
    float* z;             // [N][64][HW]   relu(conv1'(y) * scale1 + shift1)
    int N, HW;
    int tiles_per_img;    // HW / 64
    int total_tiles;      // N * tiles_per_img
};

struct PwChainF32Launch {
    PwChainF32Params p;
    bool ds, red;  // the instantiation: down-sample residual computed here / reduce conv follows
    int form;      // 4: four waves, 64-pixel tiles; 8: eight waves (pixel half x cout quarter); 2: four waves, 32-pixel tiles, two workgroups per CU
    int grid;
    size_t lds_bytes;
};

int pwchain32_build(const float* mid, const float* res, const float* x0, const float* packed_wd, const float* scale_d, const float* shift_d,
                    const float* packed_w3, const float* scale3, const float* shift3, const float* packed_w1, const float* scale1,
                    const float* shift1, float* y, float* z, int n, int cm, int ce, int cr, int h, int w,
                    PwChainF32Launch& L);  // MP_OK / MP_ERR_UNSUPPORTED
int pwchain32_launch(const PwChainF32Launch& L, hipStream_t s);

}  // namespace mp
