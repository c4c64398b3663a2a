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
    int n_chunks, n_ct, bands, total_blocks;  // total_blocks = tiles; a workgroup takes tiles_per_wg consecutive ones
    int tiles_per_wg;
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
    int teams;   // 1 or 2 four-wave teams per workgroup (2: a 64-channel cout tile on one shared input transform)
};

// U = G g G^T of one (cout co, cin ci) pair -> out[16] (xi = 4 row + col).  `transposed` = 0: w is [cout][cin][3][3] (forward);
// != 0: the data-gradient form of a forward weight [cin][cout][3][3] in THIS convolution's terms (roles swapped, taps mirrored -
// mode 2 of mp_conv_pack_weight).  Callers have checked co < cout && ci < cin.
__device__ __forceinline__ void wino_transform_weight(const float* __restrict__ w, int cout, int cin, int transposed, int co, int ci,
                                                      float (&u)[16]) {
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            g[a][c] = transposed ? w[(((size_t)ci * cout + co) * 3 + (2 - a)) * 3 + (2 - c)] : w[(((size_t)co * cin + ci) * 3 + a) * 3 + c];
    float t[4][3];  // G g
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t[0][c] = g[0][c];
        t[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
        t[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
        t[3][c] = g[2][c];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        u[a * 4 + 0] = t[a][0];
        u[a * 4 + 1] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
        u[a * 4 + 2] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
        u[a * 4 + 3] = t[a][2];
    }
}

int wino_pack_launch(const float* w, float* packed, int cout, int cin, int transposed, hipStream_t s);
int wino_configure(const mp_conv_desc* d, WinoLaunch& L);  // MP_OK / MP_ERR_UNSUPPORTED; pointers left null
int wino_launch(const WinoLaunch& L, hipStream_t s);
unsigned long long* conv_stamp_buffer(size_t need_bytes);  // conv_api.hip: mp_debug_set_stamp_buffer's buffer when large enough, else null

}  // namespace mp
