// fp32 1x1 convolution as a blocked GEMM (conv_gemm_f32.hip): launch record shared with the conv API / plan replay (conv_api.hip).
#pragma once
#include "common.h"

namespace mp {

struct GemmParams {
    const float* x;
    const float* wp;  // the direct kernel's packing of a 1x1 weight: [Cin_pad4][Cout_pad16] (cout contiguous)
    const float* scale;
    const float* shift;
    const float* res1;
    float* out;
    int N, Cin, Cin_pad4, Cout, Cout_pad16;
    int HWi;        // pixels of an input plane
    int Wi;         // input row length (stride 2: the gather walks rows)
    int HWo, Wo;    // pixels / row length of an output plane
    int cols;       // N * HWo: the GEMM's column count
    int n_ct;       // 128-channel cout tiles
    int n_chunks;   // ceil(Cin / 16)
    int relu;
    unsigned magic_hwo, magic_wo;
};

struct GemmLaunch {
    GemmParams p;
    int stride;  // 1 or 2 and
    int ni;      // 32-column blocks per wave (2: 128-column tiles, 1: 64-column tiles): the instantiation
    int grid;
    size_t lds_bytes;
};

int gemm_configure(const mp_conv_desc* d, GemmLaunch& L);  // MP_OK / MP_ERR_UNSUPPORTED; pointers left null
int gemm_launch(const GemmLaunch& L, hipStream_t s);

}  // namespace mp
