// First convolution of the network under amp O2 (hrnet.py:377-385: 3x3 stride 2 padding 1, 3 -> 64 channels, BatchNorm, ReLU) straight
// from the fp32 NCHW image to the channel-blocked fp16 activation - round 4.
//
// Why: the two launches it replaces are the layout pass (mp_f16_to_c8: read 75 MB of fp32, write 100 MB of 8-channel blocks of which
// 5 channels are zeros; 31 us at N = 128) and the general fp16 conv on those blocks (106 us): with 3 real channels in a 32-deep
// k-step the matrix pipe multiplies zeros 29 times out of 32 - nine k-steps (one per tap) of it - and the launch is MFMA-bound on
// padding.  Here the k axis is (tap, channel): 9 x 3 = 27 of the 32 positions of ONE k-step carry data, i.e. one MFMA per 16 pixels x
// 16 output channels instead of nine, and the image is read once:
//   * a workgroup takes R output rows of one image: the 2 R + 1 input rows of the three planes go global (16-byte loads) -> fp16 ->
//     LDS [channel][row][column + 1] (column 0 and rows outside the image are the zero padding);
//   * a lane's B fragment (pixel lr of the tile, k = 8 lq .. 8 lq + 7) is eight 2-byte LDS reads at offsets fixed per lane
//     (k -> (tap, channel) -> (row, column, plane) offset), the same for every pixel tile;
//   * the weights [64][3][3][3] are 1728 numbers: every lane builds its four A fragments (k order as above, zeros behind k = 26)
//     from the fp32 tensor itself, once - no packing pass;
//   * epilogue as everywhere: scale / shift, ReLU, one rounding, 16-byte stores of whole channel blocks (cout-tile pairing).
// HBM: 75 MB in, 201 MB out (N = 128, 256x192): the launch is a streaming kernel.  Sums of 27 products in another order than the
// nine-k-step kernel: equal to it within one fp16 rounding of the output, not bit for bit (tests/test_gpu_f16.py::test_stem_conv_*).
#include "conv_f16.h"
#include "conv_f16_dev.h"

namespace mp {

namespace {

constexpr int kStemR = 8;  // output rows per workgroup (17 input rows: 6 % of halo re-reads)

__global__ __launch_bounds__(256) void stem_conv_f16_kernel(const StemF16Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds_x[];  // [3][2 R + 1][pitch] fp16 bit patterns; + 8 zeros behind
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    constexpr int RIN = 2 * kStemR + 1, CS = 4;
    const int pitch = p.pitch;
    int b = blockIdx.x;
    {
        const int nb = p.total_blocks, q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, j = b >> 3;
        b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const int n = b / p.tiles_y, y0 = (b - n * p.tiles_y) * kStemR;

    // ---- this lane's k positions: k = 8 lq + j -> (tap, channel) = (k / 3, k % 3) -> element offset in the tile; k >= 27: the zero slot
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * lq + j, tap = k / 3, c = k - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
        koff[j] = k < 27 ? (c * RIN + ky) * pitch + kx : -1;
    }
    // ---- A fragments: row lr of cout tile cs (paired order), k as above, from the fp32 weights [64][3][3][3]
    u32x4 A[CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        const int co = f16_a_row<CS>(cs, lr);
        f16x8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * lq + j, tap = k / 3, c = k - 3 * tap;
            a[j] = k < 27 ? (_Float16)p.w[(co * 3 + c) * 9 + tap] : (_Float16)0.f;
        }
        A[cs] = __builtin_bit_cast(u32x4, a);
    }
    f32x4 sc[CS], sh[CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        const int co = f16_d_cout<CS>(cs, lq);
        sc[cs] = *reinterpret_cast<const f32x4*>(p.scale + co);
        sh[cs] = *reinterpret_cast<const f32x4*>(p.shift + co);
    }
    // ---- input rows 2 y0 - 1 ... 2 y0 + 2 R - 1 of the three planes: float4 units, converted on the way
    {
        const int upr = p.W >> 2, units = 3 * RIN * upr;  // W % 4 == 0
        const float* img = p.x + (size_t)n * 3 * p.H * p.W;
        for (int u = tid; u < units; u += 256) {
            const int cr = (int)__umulhi((unsigned)u, p.magic_upr), xu = u - cr * upr;  // u / upr (a runtime division is ~40 instructions)
            const int c = cr / RIN, r = cr - c * RIN;
            const int yin = 2 * y0 - 1 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yin >= 0 && yin < p.H) v = *reinterpret_cast<const float4*>(img + ((size_t)c * p.H + yin) * p.W + 4 * xu);
            const f16x4 h = (f16x4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            const u32x2 hb = __builtin_bit_cast(u32x2, h);
            // column ix lives at index ix + 1 (index 0 = the left padding): 4 xu + 1 is odd - two 2-byte and one 4-byte store
            unsigned short* dst = lds_x + (c * RIN + r) * pitch + 4 * xu + 1;
            dst[0] = (unsigned short)(hb.x & 0xFFFFu);
            *reinterpret_cast<unsigned*>(dst + 1) = (hb.x >> 16) | (hb.y << 16);
            dst[3] = (unsigned short)(hb.y >> 16);
        }
        for (int i = tid; i < 3 * RIN; i += 256) lds_x[i * pitch] = 0;  // left padding column
        if (tid < 8) lds_x[3 * RIN * pitch + tid] = 0;                  // the slot the padding k positions read
    }
    __syncthreads();

    const int tiles_row = p.Wo >> 4, n_tiles = kStemR * tiles_row;  // Wo % 16 == 0: a pixel tile never straddles rows
    const size_t plane_o = (size_t)p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, (size_t)p.N * 8 * plane_o * 16);
    const int zero_slot = 3 * RIN * pitch;
    for (int t = wave; t < n_tiles; t += 4) {
        const int ry = t / tiles_row, ox = (t - ry * tiles_row) * 16 + lr;
        const int base = 2 * ry * pitch + 2 * ox;  // window origin: input row 2 ry (tile row 0 = image row 2 y0 - 1), index 2 ox
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lds_x[koff[j] >= 0 ? base + koff[j] : zero_slot];
        const u32x4 bv = (u32x4){(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                                 (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
        f32x4 acc[CS];
#pragma unroll
        for (int cs = 0; cs < CS; ++cs)
            acc[cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[cs]), __builtin_bit_cast(f16x8, bv),
                                                             (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const int oy = y0 + ry;
        if (oy < p.Ho) {  // wave-uniform
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x2 lo = f16_pack4(f16_epi4(acc[2 * j], sc[2 * j], sh[2 * j], false, (u32x2){0u, 0u}, false, (u32x2){0u, 0u}, p.relu));
                const u32x2 hi = f16_pack4(f16_epi4(acc[2 * j + 1], sc[2 * j + 1], sh[2 * j + 1], false, (u32x2){0u, 0u}, false, (u32x2){0u, 0u}, p.relu));
                const size_t e = ((size_t)n * 8 + 4 * j + lq) * plane_o + (size_t)oy * p.Wo + ox;
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){lo.x, lo.y, hi.x, hi.y}, rs_o, (unsigned)(e * 16), 0, 0);
            }
        }
    }
}

}  // namespace

int stemf16_build(const float* x, const float* w, const float* scale, const float* shift, int relu, void* out, int n, int h, int wd,
                  StemF16Launch& L) {
    if (!x || !w || !scale || !shift || !out) return MP_ERR_NULL;
    if (n <= 0 || h <= 0 || wd <= 0) return MP_ERR_SHAPE;
    if ((h & 1) || (wd & 3) || ((wd >> 1) & 15)) return MP_ERR_UNSUPPORTED;  // even rows, float4 units, whole pixel tiles per output row
    const int ho = h / 2, wo = wd / 2;
    if ((long long)n * 8 * ho * wo * 16 >= 0x7FFFFFF0LL) return MP_ERR_UNSUPPORTED;
    StemF16Params& p = L.p;
    p.x = x; p.w = w; p.scale = scale; p.shift = shift; p.out = out; p.relu = relu ? 1 : 0;
    p.N = n; p.H = h; p.W = wd; p.Ho = ho; p.Wo = wo;
    p.magic_upr = magic_of((unsigned)(wd >> 2));
    p.pitch = wd + 4;  // even: the 4-byte stores of the staging loop are aligned; index W + 1 .. W + 3 never read
    p.tiles_y = (ho + kStemR - 1) / kStemR;
    p.total_blocks = n * p.tiles_y;
    L.lds_bytes = ((size_t)3 * (2 * kStemR + 1) * p.pitch + 8) * 2;
    if (L.lds_bytes > 64 * 1024) return MP_ERR_UNSUPPORTED;
    return MP_OK;
}

int stemf16_launch(const StemF16Launch& L, hipStream_t s) {
    hipLaunchKernelGGL(stem_conv_f16_kernel, dim3(L.p.total_blocks), dim3(256), L.lds_bytes, s, L.p);
    return check_launch();
}

}  // namespace mp

using namespace mp;

extern "C" int mp_f16_stem_conv_fwd(const float* x, const float* weight, const float* scale, const float* shift, int relu, void* out, int n,
                                    int h, int w, mp_stream_t stream) {
    StemF16Launch L{};
    const int rc = stemf16_build(x, weight, scale, shift, relu, out, n, h, w, L);
    if (rc != MP_OK) return rc;
    return stemf16_launch(L, as_stream(stream));
}
