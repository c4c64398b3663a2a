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
    const float* res2;  // second residual tensor (the exchange unit's running sum + identity), same geometry as the output
    float* out;
    int N, Cin, Cin_pad4, Cout, Cout_pad16;
    int HWi, Hi, Wi;        // input plane
    int HWo, Wo;            // the convolution's own output grid (columns of the GEMM: N * HWo)
    int OHW, OW;            // the output tensor's plane (sub-pixel phases write every out_mul-th pixel of it)
    int out_mul, off_y, off_x;
    int stride, pad_top, pad_left;
    int T, kw;      // taps (1, 4 = 2x2, 9 = 3x3) and the kernel's width
    int cols;       // N * HWo: the GEMM's column count
    int n_ct;       // cout tiles (128 or 64 channels)
    int n_iters;    // (Cin / 16) * T k-loop steps, walked as (cin chunk, tap) pairs
    int relu;
    int phases;                 // 1, or 4 = all sub-pixel phases of a transposed convolution in one launch (grid.y)
    unsigned wp_phase_floats;   // floats per phase slice of the packed weights
};

struct GemmLaunch {
    GemmParams p;
    int stride, taps, phases;
    bool gather; // columns fetched one by one (stride 2, 2x2 phases) and
    int ni, mi;  // 32-column / 32-cout blocks per wave (2: 128-wide tiles, 1: 64-wide): the instantiation
    int grid;
    size_t lds_bytes;
};

int gemm_configure(const mp_conv_desc* d, GemmLaunch& L);  // MP_OK / MP_ERR_UNSUPPORTED; pointers left null
int gemm_configure_deconv(const mp_conv_desc* phase00, GemmLaunch& L);  // all four phases of Conv2dTranspose(k=4, s=2, p=1)
int gemm_launch(const GemmLaunch& L, hipStream_t s);

}  // namespace mp
