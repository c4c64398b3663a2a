// fp32 3x3 (stride 1 / 2) and 1x1 convolution for SMALL problems (conv_small_f32.hip): launch record shared with the conv API / plan replay (conv_api.hip).
#pragma once
#include "common.h"

namespace mp {

struct SmallParams {
    const float* x;      // [N][Cin][H][W]
    const float* wp;     // the direct kernel's packing: [Cin_pad4 / 4][9][4][Cout_pad16]
    const float* scale;  // [Cout_pad16] folded BatchNorm (or 1 / bias)
    const float* shift;
    const float* res1;   // optional residual tensors in the output geometry
    const float* res2;
    float* out;          // [N][Cout][H][W]
    int N, Cin, Cin_pad4, Cout, Cout_pad16, H, W;   // H x W: the OUTPUT map (= the input map for stride 1)
    int Hin, Win, HWin;  // the input map
    int ks, stride, pad; // 3x3 stride 1 / 2 pad 1, or 1x1 stride 1 pad 0
    int HW, Wp;          // Wp: a staged input row with its halo columns (stride * (W - 1) + ks)
    int rows;            // staged input rows per tile: what the output rows 16 consecutive pixels can span need
    int plane;           // floats per staged channel plane (rows * Wp, padded to 16 mod 32)
    int pt;              // 16-pixel tiles per workgroup (1; the wide form: 3 or 4)
    int tiles_img;       // workgroup tiles (16 pt pixels) per image
    int n_ct;            // 16-cout tiles
    int kq;              // Cin_pad4 / 4: k-steps of one tap
    int relu;
    unsigned magic_w, magic_wp, magic_tiles, magic_nct, magic_per_plane;
};

struct SmallLaunch {
    SmallParams p;
    int grid;
    size_t lds_bytes;
};

int small_configure(const mp_conv_desc* d, SmallLaunch& L, int wide);  // wide: 48 / 64 pixels per workgroup;  // MP_OK / MP_ERR_UNSUPPORTED; pointers left null
int small_launch(const SmallLaunch& L, hipStream_t s);

}  // namespace mp
