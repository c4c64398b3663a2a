// fp16-MFMA convolution family (amp level O2/O3 inference): declarations shared by conv_f16.hip and the launch plan.
//
// Activations live in HBM channel-blocked, [N][ceil(C/8)][H][W][8] halfs ("c8"): one pixel of one 8-channel block is a
// 16-byte element, which is exactly the per-lane operand of v_mfma_f32_16x16x32_f16 (8 consecutive k), so global ->
// LDS staging is plain 16-byte copies and every MFMA operand is ONE ds_read_b128 at any tap offset or stride.
// Padding channels of the last block are kept zero by every producer.
#pragma once
#include "common.h"

namespace mp {

struct ConvF16Params {
    const void* x;       // c8 halfs
    const void* wp;      // packed weights [Cin_pad32/32][T][4][Cout_pad16][8] halfs
    const float* scale;  // [Cout_pad16] fp32 (zero beyond Cout)
    const float* shift;  // [Cout_pad16]
    const void* res1;    // c8 halfs, output geometry
    const void* res2;
    void* out;           // c8 halfs
    int N, C8in, H, W;
    int Cout, Cout_pad16, C8out;
    int Ho, Wo, pad_t, pad_l;
    int R, G, Rin, Wp, img_plane;
    int plane;       // LDS elements (16 B) per 8-channel plane: == 0 (mod 16) for stride 1, odd for stride 2
    int PK;          // planes per chunk (multiple of 4 = one MFMA k-step of 32 channels)
    int PKs;         // planes per chunk that are staged (PK, or C8in when the single chunk is zero padded)
    int n_chunks, n_ct, tiles_y, tiles_n;
    int ncols, upc;  // staged columns per row, staging units per plane (G * Rin * ncols)
    int in_buf;      // LDS elements per input buffer (PK * plane)
    int w_buf;       // LDS elements per weight buffer (PK * T * CT)
    int nbuf;
    int relu;
    int out_h, out_w, out_mul, off_y, off_x;  // output mapping: conv pixel (y, x) -> (y*out_mul + off_y, x*out_mul + off_x)
    unsigned magic_upc, magic_ncols, magic_rin, magic_rwo, magic_wo;
    int RWo, total_blocks;
    int phases;              // 4: MP_CONV_PHASES4 - blockIdx.y = phase 2 py + px (its output offset, its weight slice); else 1
    unsigned w_phase_bytes;  // bytes per weight slice
    // persistent multi-tile kernel only: a workgroup keeps its weight slice in LDS and walks tiles_per_wg pixel tiles
    int tiles_total, tiles_per_wg, n_groups;
    int ni_used, nw_used;  // staging slots (of the kernel's NI / NW) that carry data for this shape: the rest are skipped
    unsigned magic_rows;       // / (G * Rin)
    int step_rows, step_cols;  // 256 staging units = step_rows whole rows + step_cols columns (slot-to-slot stepping)
    // BatchNorm statistics from the epilogue (training; conv_f16_dev.h): 0 = off, 1 = forward sums of the output, 2 = backward sums
    // of the masked gradient (the stored tensor is then the masked gradient)
    int st_mode, st_nparts, st_relu;
    float* st_part;        // [C8out][st_nparts][8][2] fp32
    const void* st_z;      // mode 2: the BatchNorm's input (output geometry, c8 halfs)
    const void* st_y;      // mode 2, st_relu == 1: the BatchNorm's output (mask = y > 0)
    // BatchNorm apply on the INPUT operand (training, mode 1, weights-in-registers kernel): the tensor at x is the raw conv output z
    // of the layer below; the staged tile becomes act(z * pre_scale[c] + pre_shift[c]) in LDS (padding stays zero) and the tile's own
    // rows are also written to pre_out - the activation tensor the backward pass reads - so no separate apply pass runs
    const float* pre_scale;  // [PK * 8] (zeros behind Cin); null: off
    const float* pre_shift;
    void* pre_out;           // may be null
    int pre_relu;
};

struct ConvF16Launch {
    ConvF16Params p;
    int ks, stride, variant;
    size_t lds_bytes;
};

// 0..4 tile shapes (cout tile x pixel tile), 5..9 their light builds, 10..14 / 15..19 the persistent multi-tile kernel
// (weights resident in LDS, input tiles double-buffered) compiled for two / one workgroup per CU
enum ConvF16Variant { F_CT32_PT192 = 0, F_CT64_PT192 = 1, F_CT48_PT192 = 2, F_CT64_PT96 = 3, F_CT32_PT96 = 4,
                      F_CT32_PT192_L = 5, F_CT64_PT192_L = 6, F_CT48_PT192_L = 7, F_CT64_PT96_L = 8, F_CT32_PT96_L = 9,
                      F_MT2_BASE = 10, F_MT1_BASE = 15,
                      // 16-cout tiles (twice the workgroups: latency hiding for the large-K small-map layers)
                      F_CT16_PT192 = 20, F_CT16_PT192_L = 21, F_CT16_PT192_MT2 = 22, F_CT16_PT192_MT1 = 23,
                      // 384-pixel tiles for the small-K layers: a workgroup's fixed set-up (a third of its life at K = 288)
                      // and its weight loads are spread over twice the MFMA work
                      F_CT32_PT384 = 24,
                      // "weights in registers" kernel (conv_f16_wreg.hip): P<pixel tiles of 16>C<cout tiles of 16 per wave, x 4 waves>
                      // _W<n>: n waves split the pixel tiles (4 / n split the couts); no suffix = 1
                      F_WREG_P6C2 = 25, F_WREG_P3C2 = 26, F_WREG_P6C1 = 27, F_WREG_P6C3 = 28, F_WREG_P3C3 = 29, F_WREG_P3C4 = 30,
                      F_WREG_P6C2_W2 = 31, F_WREG_P6C3_W2 = 32, F_WREG_P3C2_W2 = 33, F_WREG_P6C2_W4 = 34, F_WREG_P6C3_W4 = 35,
                      F_WREG_P3C2_W4 = 36,
                      // weight-stationary persistent kernel (conv_f16_ws.hip): shapes in kWsShapes
                      F_WS_BASE = 37, F_WS_COUNT = 8,
                      // round 4, weights-in-registers shapes whose pixel tile makes W48's deep layers ONE round of 256 workgroups:
                      // 112 px x 192 couts (192 -> 192 @24x18, N = 64: 6 rows of 18 = 4 tiles per image x 64) and 64 px x 192 couts
                      // (384 -> 384 @12x9: 7 rows of 9 = 2 tiles per image x 64 x 2 cout slices); P5C4 = 80 px x 256 couts
                      F_WREG_P7C3 = 45, F_WREG_P4C3 = 46, F_WREG_P5C4 = 47, F_COUNT = 48 };
inline bool f16_variant_wreg(int v) { return (v >= F_WREG_P6C2 && v <= F_WREG_P3C2_W4) || (v >= F_WREG_P7C3 && v <= F_WREG_P5C4); }
inline int f16_wreg_index(int v) { return v <= F_WREG_P3C2_W4 ? v - F_WREG_P6C2 : 12 + v - F_WREG_P7C3; }
inline bool f16_variant_ws(int v) { return v >= F_WS_BASE && v < F_WS_BASE + F_WS_COUNT; }
inline bool f16_variant_mt(int v) { return (v >= F_MT2_BASE && v < F_CT16_PT192) || v == F_CT16_PT192_MT2 || v == F_CT16_PT192_MT1; }
inline int f16_variant_mt_occ(int v) { return (v >= F_MT1_BASE && v < F_CT16_PT192) || v == F_CT16_PT192_MT1 ? 1 : 2; }
bool f16_variant_light(int v);

// fused BasicBlock of the 32-channel branch (basicblock_f16.hip)
struct BlockF16Params {
    const void* x;
    const void* w1;
    const void* w2;
    const float* scale1;
    const float* shift1;
    const float* scale2;
    const float* shift2;
    void* out;
    int N, H, W;
    int R, Wp, plane_in, plane_mid;  // LDS planes in 16-byte units (multiples of 16: conflict-free ds_read_b128)
    int M1, M2;                      // intermediate / output pixels of a tile
    int in_units;                    // staged 16-byte units of the input tile (4 planes x (R+4) rows x W)
    int tiles_y, tiles_total, tiles_per_wg, total_blocks, ni_used;
    unsigned magic_w, magic_rw;      // / W, / ((R+4) * W)   (second structure: / (W + 1), / plane_in)
    unsigned magic_wo;               // second structure: / W
    unsigned long long* dbg;         // diagnostic builds (-DMP_BLOCK_STAMPS=1) only: 16 x u64 per (workgroup, wave half)
};

struct BlockF16Launch {
    BlockF16Params p;
    int small;  // 1: the <5,3> pixel-tile build suffices, 0: <6,5>
    size_t lds_bytes;
};
int blockf16_build(const void* x, const void* w1, const float* scale1, const float* shift1, const void* w2, const float* scale2,
                   const float* shift2, void* out, int n, int c, int h, int w, int rows, BlockF16Launch& L);
int blockf16_launch(const BlockF16Launch& L, hipStream_t s);
// second structure (basicblock_f16_v2.hip): weights in registers, LDS-DMA bands, 512-thread workgroups; L.small == 2 marks it
bool blockf16_v2_build(const void* x, const void* w1, const float* scale1, const float* shift1, const void* w2, const float* scale2,
                       const float* shift2, void* out, int n, int c, int h, int w, int rows, BlockF16Launch& L);
int blockf16_v2_launch(const BlockF16Launch& L, hipStream_t s);
// 64-channel block (basicblock_f16_c64.hip): one band per workgroup, cout tile per wave; L.small == 4 marks it
bool blockf16_c64_build(const void* x, const void* w1, const float* scale1, const float* shift1, const void* w2, const float* scale2,
                        const float* shift2, void* out, int n, int c, int h, int w, int rows, BlockF16Launch& L);
int blockf16_c64_launch(const BlockF16Launch& L, hipStream_t s);

// two chained 1x1 convs of HRNet's stage 1 in one launch (pwchain_f16.hip): expand conv of Bottleneck i + reduce conv of i + 1
struct PwChainParams {
    const void* mid;   // [N][CM/8][HW] x 16 B: the 3x3 conv's output
    const void* res;   // [N][CE/8][HW]: the identity of the expand conv
    const void* w3;    // packed 1x1 weights CM -> CE
    const float* scale3;
    const float* shift3;
    const void* w1;    // packed 1x1 weights CE -> CR
    const float* scale1;
    const float* shift1;
    void* y;           // [N][CE/8][HW]
    void* z;           // [N][CR/8][HW]
    int N, HW, tiles_per_img, total_blocks, relu3, relu1;
    // down-sample form: the identity is conv1x1(x0; wd) * scale_d + shift_d (no ReLU), computed in the launch (res unused)
    const void* x0;    // [N][CM/8][HW]: the block's input
    const void* wd;    // packed 1x1 weights CM -> CE
    const float* scale_d;
    const float* shift_d;
};
struct PwChainLaunch {
    PwChainParams p;
    int cm, ce, cr, h, w;
    bool dual;  // both convs read `mid` (the first Bottleneck's down-sample + reduce convs): res == mid marks it
    bool ds;    // the identity is the block's down-sample conv of x0, computed in the launch
    size_t lds_bytes;
};
int pwchain_build(const void* mid, const void* res, const void* w3, const float* scale3, const float* shift3, int relu3, const void* w1,
                  const float* scale1, const float* shift1, int relu1, void* y, void* z, int n, int cm, int ce, int cr, int h, int w,
                  PwChainLaunch& L, const void* x0 = nullptr, const void* wd = nullptr, const float* scale_d = nullptr,
                  const float* shift_d = nullptr);
int pwchain_launch(const PwChainLaunch& L, hipStream_t s);

// first conv of the network straight from the fp32 NCHW image (stem_f16.hip): 3x3 stride 2 padding 1, 3 -> 64 channels
struct StemF16Params {
    const float* x;      // [N][3][H][W] fp32
    const float* w;      // [64][3][3][3] fp32
    const float* scale;  // [64]
    const float* shift;
    void* out;           // [N][8][H/2][W/2] x 16 B
    int N, H, W, Ho, Wo, pitch, tiles_y, total_blocks, relu;
    unsigned magic_upr;  // / (W / 4)
};
struct StemF16Launch {
    StemF16Params p;
    size_t lds_bytes;
};
int stemf16_build(const float* x, const float* w, const float* scale, const float* shift, int relu, void* out, int n, int h, int wd,
                  StemF16Launch& L);
int stemf16_launch(const StemF16Launch& L, hipStream_t s);

int f16_build_launch(const mp_conv_desc* desc, int variant, const void* x, const void* w, const float* scale,
                     const float* shift, const void* res1, const void* res2, void* out, ConvF16Launch& L);
int f16_launch(const ConvF16Launch& L, hipStream_t s);
// partial-sum slots (per channel) a launch with epilogue statistics writes: one per pixel-tile workgroup / persistent workgroup
int f16_stats_parts(const ConvF16Launch& L);
void f16_variant_dims(int v, int& ct, int& pt);
int f16_mt_launch(const ConvF16Launch& L, hipStream_t s);
bool f16_configure_wreg(const mp_conv_desc& d, int variant, ConvF16Launch& L);
int f16_wreg_launch(const ConvF16Launch& L, hipStream_t s);
void f16_wreg_dims(int v, int& ps, int& csw, int& waves_p);
int f16_mt_ni(int occ);
bool f16_configure_ws(const mp_conv_desc& d, int variant, ConvF16Launch& L);
int f16_ws_launch(const ConvF16Launch& L, hipStream_t s);
void f16_ws_dims(int v, int& ps, int& csw, int& waves_p);

}  // namespace mp
