// Winograd F(2x2, 3x3) fp32 convolution (conv_wino_f32.hip): launch record shared with the plan replay (conv_api.hip).
#pragma once
#include "common.h"

namespace mp {

struct WinoParams {
    const float* x;
    const float* u;  // transformed weights [Cin_pad4][Cout_pad16][16]
    const float* scale;
    const float* shift;
    const float* res1;
    const float* res2;
    float* out;
    int N, Cin, Cout, Cout_pad16, H, W;
    int TW, TR, M;   // tiles per row, tile rows per workgroup, tiles per workgroup (<= 48, even)
    int G, tpi, img_plane;  // image-grouped bands (W % 4 != 0 maps): images per workgroup, tiles per image, LDS floats per image
    int R, Rin, Wp, cin_plane;
    int n_chunks, n_ct, bands, total_blocks;
    int upr, upc;    // float4 staging units per input row / per input channel
    int relu;
    unsigned magic_upr, magic_upc, magic_tw, magic_pairs, magic_tpi, magic_w;
    unsigned long long* dbg;  // diagnostic builds (MP_CONV_STAMPS) only: 8 x u64 per workgroup
};

struct WinoLaunch {
    WinoParams p;
    size_t lds_bytes;
    int ni;
    bool group;  // image-grouped bands
};

int wino_configure(const mp_conv_desc* d, WinoLaunch& L);  // MP_OK / MP_ERR_UNSUPPORTED; pointers left null
int wino_launch(const WinoLaunch& L, hipStream_t s);
unsigned long long* conv_stamp_buffer(size_t need_bytes);  // conv_api.hip: mp_debug_set_stamp_buffer's buffer when large enough, else null

}  // namespace mp
