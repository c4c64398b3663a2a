// First convolution of the network in fp32 (hrnet.py:377-385: 3x3 stride 2 padding 1, 3 -> 64 channels, BatchNorm, ReLU) as a
// streaming kernel - round 4.
//
// Why: the general direct kernel runs this layer at 168 - 175 us for N = 128 (75 MB in, 403 MB out: 2.8 TB/s, 32 TFLOP/s; this
// kernel: 120 - 143 us over boxes) - a tap-major
// loop over a 4-deep k-step that holds 3 channels, and a pixel tile sized for the deep layers.  The layer is 5.4 GFLOP for 478 MB:
// what matters is that the output streams.  Here, as in the fp16 stem kernel (stem_f16.hip), the k axis is (tap, channel) - 27 real
// positions in seven k-steps of v_mfma_f32_16x16x4_f32 - the weights of all 64 output channels sit in 28 registers per lane for the
// workgroup's life, and a workgroup streams 8 output rows of one image: 17 input rows of the three planes global -> LDS
// ([channel][row][column + 1], zero padding in place; an aligned [column + 4] image with 16-byte LDS stores measured the same), per 16-pixel tile seven 4-byte LDS reads (offsets fixed per lane) and 28 MFMAs,
// 16-byte stores (a lane holds 4 consecutive pixels of one output channel).  fp32 throughout; the sums run in another order than the
// direct kernel's (tests: 1e-5 of an fp64 convolution, like every fp32 kernel here).
#include "conv_mfma.h"
#include "conv_stem.h"

namespace mp {

namespace {

constexpr int kR = 8;  // output rows per workgroup

__global__ __launch_bounds__(256) void stem_conv_f32_kernel(const StemF32Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds_x[];  // [3][2 R + 1][pitch]; + 4 zeros behind
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    constexpr int RIN = 2 * kR + 1, KQ = 7, CB = 4;
    const int pitch = p.pitch;
    int b = blockIdx.x;
    {
        const int nb = p.total_blocks, q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, j = b >> 3;
        b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const int n = b / p.tiles_y, y0 = (b - n * p.tiles_y) * kR;

    // ---- this lane's k positions (A operand: row = pixel lr, k = 4 q + lq): (tap, channel) = (k / 3, k % 3) -> offset in the tile
    int koff[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const int k = 4 * q + lq, tap = k / 3, c = k - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
        koff[q] = k < 27 ? (c * RIN + ky) * pitch + kx : -1;
    }
    // ---- B operand (k = 4 q + lq, cout = 16 cb + lr) from the fp32 weights [64][3][3][3], all k-steps: 28 registers
    float wB[KQ][CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int k = 4 * q + lq, tap = k / 3, c = k - 3 * tap;
            wB[q][cb] = k < 27 ? p.w[((16 * cb + lr) * 3 + c) * 9 + tap] : 0.f;
        }
    float sc[CB], sh[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        sc[cb] = p.scale[16 * cb + lr];
        sh[cb] = p.shift[16 * cb + lr];
    }
    // ---- input rows 2 y0 - 1 ... 2 y0 + 2 R - 1 of the three planes: float4 units; column ix at index ix + 1 (index 0 = left padding)
    {
        const int upr = p.W >> 2, units = 3 * RIN * upr;
        const float* img = p.x + (size_t)n * 3 * p.H * p.W;
        // (one request per loop pass: twelve units per thread requested up front measured SLOWER - 138 -> 166 us - the registers cost a
        //  resident workgroup, and it is the other workgroups of the CU that hide this loop's round trips)
        for (int u = tid; u < units; u += 256) {
            const int cr = (int)__umulhi((unsigned)u, p.magic_upr), xu = u - cr * upr;
            const int c = cr / RIN, r = cr - c * RIN;
            const int yin = 2 * y0 - 1 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yin >= 0 && yin < p.H) v = *reinterpret_cast<const float4*>(img + ((size_t)c * p.H + yin) * p.W + 4 * xu);
            float* dst = lds_x + (c * RIN + r) * pitch + 4 * xu + 1;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        for (int i = tid; i < 3 * RIN; i += 256) lds_x[i * pitch] = 0.f;
        if (tid < 4) lds_x[3 * RIN * pitch + tid] = 0.f;
    }
    __syncthreads();

    const int tiles_row = p.Wo >> 4, n_tiles = kR * tiles_row;  // Wo % 16 == 0
    const size_t plane_o = (size_t)p.Ho * p.Wo;
    float* outn = p.out + (size_t)n * 64 * plane_o;
    const int zero_slot = 3 * RIN * pitch;
    // a wave takes PAIRS of neighbouring tiles: a lane group's 64-byte store of tile 2 i and the one of tile 2 i + 1 are the two halves
    // of one 128-byte line of an output row - written back to back by the same wave they leave L2 as one line
    for (int tt = 2 * wave; tt < n_tiles; tt += (tt & 1) ? 7 : 1) {
        const int t = tt;
        const int ry = t / tiles_row, ox0 = (t - ry * tiles_row) * 16;
        const int base = 2 * ry * pitch + 2 * (ox0 + lr);  // window origin of pixel lr of the tile
        float a[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) a[q] = lds_x[koff[q] >= 0 ? base + koff[q] : zero_slot];
        f32x4 acc[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KQ; ++q)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], wB[q][cb], acc[cb], 0, 0, 0);
        const int oy = y0 + ry;
        if (oy < p.Ho) {  // wave-uniform
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                // D: rows 4 lq .. 4 lq + 3 = four consecutive pixels, column lr = output channel 16 cb + lr
                f32x4 v = acc[cb] * sc[cb] + sh[cb];
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<f32x4*>(outn + (size_t)(16 * cb + lr) * plane_o + (size_t)oy * p.Wo + ox0 + 4 * lq) = v;
            }
        }
    }
}

}  // namespace

int stemf32_build(const float* x, const float* w, const float* scale, const float* shift, int relu, float* out, int n, int h, int wd,
                  StemF32Launch& L) {
    if (!x || !w || !scale || !shift || !out) return MP_ERR_NULL;
    if (n <= 0 || h <= 0 || wd <= 0) return MP_ERR_SHAPE;
    if ((h & 1) || (wd & 3) || ((wd >> 1) & 15)) return MP_ERR_UNSUPPORTED;  // even rows, float4 units, whole pixel tiles per output row
    StemF32Params& p = L.p;
    p.x = x; p.w = w; p.scale = scale; p.shift = shift; p.out = out; p.relu = relu ? 1 : 0;
    p.N = n; p.H = h; p.W = wd; p.Ho = h / 2; p.Wo = wd / 2;
    p.pitch = wd + 4;
    p.magic_upr = (unsigned)(0x100000000ULL / (unsigned)(wd >> 2)) + 1u;
    p.tiles_y = (p.Ho + kR - 1) / kR;
    p.total_blocks = n * p.tiles_y;
    L.lds_bytes = ((size_t)3 * (2 * kR + 1) * p.pitch + 4) * 4;
    if (L.lds_bytes > 64 * 1024) return MP_ERR_UNSUPPORTED;
    return MP_OK;
}

int stemf32_launch(const StemF32Launch& L, hipStream_t s) {
    hipLaunchKernelGGL(stem_conv_f32_kernel, dim3(L.p.total_blocks), dim3(256), L.lds_bytes, s, L.p);
    return check_launch();
}

}  // namespace mp

using namespace mp;

extern "C" int mp_stem_conv_fwd(const float* x, const float* weight, const float* scale, const float* shift, int relu, float* out, int n,
                                int h, int w, mp_stream_t stream) {
    StemF32Launch L{};
    const int rc = stemf32_build(x, weight, scale, shift, relu, out, n, h, w, L);
    if (rc != MP_OK) return rc;
    return stemf32_launch(L, as_stream(stream));
}
