// Streaming 1x1 fp32 convolution (conv_pw_f32.hip): launch record shared with the conv API / plan replay (conv_api.hip).
#pragma once
#include "common.h"

namespace mp {

struct PwParams {
    const float* x;
    const float* wp;  // the direct kernel's packing of a 1x1 weight: [Cin/4][4][Cout_pad16]
    const float* scale;
    const float* shift;
    const float* res1;
    float* out;
    int N, Cin, Cout, Cout_pad16, HW;
    int tiles, tiles_per_wg;  // 64-pixel tiles in all / per persistent workgroup
    int relu;
};

struct PwLaunch {
    PwParams p;
    int kq, cbw;  // Cin / 4, cout blocks per wave: the instantiation
    int grid;
    size_t lds_bytes;
};

int pw_configure(const mp_conv_desc* d, PwLaunch& L);  // MP_OK / MP_ERR_UNSUPPORTED; pointers left null
int pw_launch(const PwLaunch& L, hipStream_t s);

}  // namespace mp
