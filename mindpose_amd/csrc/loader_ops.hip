// Loader-side kernel of the top-down path: the crop that feeds the network.
//   cv2.warpAffine(image, trans, (w, h), flags=INTER_LINEAR)   topdown_transform.py:211-216 / :249-254
//   vision.Normalize(mean*255, std*255) + vision.HWC2CHW()      data_factory.py:129-133
// fused into one pass: one thread per destination pixel reads <= 4 source pixels (HWC uint8) and writes three fp32
// planes (or the warped uint8 HWC pixel when only the warp is wanted).
//
// The interpolation restates OpenCV's fixed-point INTER_LINEAR path [cv2-knowledge, PARITY UNPINNED: cv2 is not
// installed here]: the 2x3 matrix is inverted in double, destination coordinates are mapped with AB_BITS = 10
// fixed-point (per-column / per-row terms rounded separately with cvRound, + round_delta 16), quantised to 1/32 pixel
// (INTER_BITS = 5), the four bilinear weights are the exact products (32-fx)(32-fy)*32 ... of the 15-bit table, the
// result is (sum + 2^14) >> 15, BORDER_CONSTANT with value 0.
#include "common.h"

#pragma clang fp contract(off)  // the rounding points of the coordinate arithmetic are part of the result

namespace mp {

namespace {

__device__ __forceinline__ int cv_round(double v) { return (int)rint(v); }  // cvRound: nearest, ties to even

template <bool NORMALIZE>
__global__ __launch_bounds__(256) void warp_affine_kernel(const uint8_t* __restrict__ src, const long long* __restrict__ src_off,
                                                          const int* __restrict__ src_hw, const int* __restrict__ flip,
                                                          const double* __restrict__ trans, void* __restrict__ out, int out_h, int out_w, float m0, float m1,
                                                          float m2, float s0, float s1, float s2) {
    const int n = blockIdx.y;
    __shared__ double inv[6];
    if (threadIdx.x == 0) {
        // cv::warpAffine without WARP_INVERSE_MAP: invert M in double
        const double* M = trans + (size_t)n * 6;
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0.0 ? 1.0 / D : 0.0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        const double i0 = A11, i1 = M[1] * (-D), i3 = M[3] * (-D), i4 = A22;
        inv[0] = i0; inv[1] = i1; inv[3] = i3; inv[4] = i4;
        inv[2] = -i0 * M[2] - i1 * M[5];
        inv[5] = -i3 * M[2] - i4 * M[5];
    }
    __syncthreads();
    const int H = src_hw[2 * n], W = src_hw[2 * n + 1];
    // cv2.flip(image, 1) before the warp (TopDownHorizontalRandomFlip) = sampling the original at the mirrored column:
    // the fixed-point coordinates and the bilinear weights are symmetric, so the result is bit-identical
    const bool mirror = flip && flip[n] != 0;
    const uint8_t* __restrict__ img = src + src_off[n];
    const int total = out_h * out_w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / out_w, x = i - y * out_w;
        const int adelta = cv_round(inv[0] * x * 1024.0), bdelta = cv_round(inv[3] * x * 1024.0);
        const int X0 = cv_round((inv[1] * y + inv[2]) * 1024.0) + 16, Y0 = cv_round((inv[4] * y + inv[5]) * 1024.0) + 16;
        const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
        int sx = X >> 5, sy = Y >> 5;
        sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);  // saturate_cast<short>
        sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
        const int fx = X & 31, fy = Y & 31;
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        int acc[3] = {0, 0, 0};
        auto tap = [&](int yy, int xx, int w) {
            if (w != 0 && yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const uint8_t* px = img + ((size_t)yy * W + (mirror ? W - 1 - xx : xx)) * 3;
                acc[0] += w * px[0]; acc[1] += w * px[1]; acc[2] += w * px[2];
            }
        };
        tap(sy, sx, w00); tap(sy, sx + 1, w01); tap(sy + 1, sx, w10); tap(sy + 1, sx + 1, w11);
        int v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = (acc[c] + (1 << 14)) >> 15;
            v[c] = v[c] > 255 ? 255 : v[c];
        }
        if constexpr (NORMALIZE) {
            float* o = reinterpret_cast<float*>(out) + (size_t)n * 3 * total;
            o[i] = ((float)v[0] - m0) / s0;
            o[total + i] = ((float)v[1] - m1) / s1;
            o[2 * total + i] = ((float)v[2] - m2) / s2;
        } else {
            uint8_t* o = reinterpret_cast<uint8_t*>(out) + ((size_t)n * total + i) * 3;
            o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[2];
        }
    }
}

}  // namespace
}  // namespace mp

using namespace mp;

extern "C" int mp_warp_affine(const uint8_t* src, const long long* src_offsets, const int* src_hw, const int* flip,
                              const double* trans, void* out, int n, int out_h, int out_w, int normalize, const float mean[3], const float stddev[3],
                              mp_stream_t stream) {
    if (!src || !src_offsets || !src_hw || !trans || !out) return MP_ERR_NULL;
    if (n <= 0 || out_h <= 0 || out_w <= 0 || n > 65535) return MP_ERR_SHAPE;
    if (normalize && (!mean || !stddev)) return MP_ERR_NULL;
    if (normalize && (stddev[0] == 0.f || stddev[1] == 0.f || stddev[2] == 0.f)) return MP_ERR_SHAPE;
    const int total = out_h * out_w;
    int bx = (total + 255) / 256;
    if (bx > 1024) bx = 1024;
    if (normalize)
        hipLaunchKernelGGL(warp_affine_kernel<true>, dim3(bx, n), dim3(256), 0, as_stream(stream), src, src_offsets, src_hw, flip, trans, out,
                           out_h, out_w, mean[0], mean[1], mean[2], stddev[0], stddev[1], stddev[2]);
    else
        hipLaunchKernelGGL(warp_affine_kernel<false>, dim3(bx, n), dim3(256), 0, as_stream(stream), src, src_offsets, src_hw, flip, trans, out,
                           out_h, out_w, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f);
    return check_launch();
}

// ---- horizontal flip of an NCHW fp32 batch: the second run of the flip test (topdown_inferencer.py:168-170, ops.ReverseV2 on
// the width axis).  out[n, c, y, x] = in[n, c, y, W - 1 - x]; one pass, written straight into the network's input buffer (the
// host mirror used torch.flip + a copy: two passes through PyTorch kernels).  in and out must not overlap.
namespace mp {
namespace {
__global__ __launch_bounds__(256) void flip_width_kernel(const float* __restrict__ in, float* __restrict__ out, size_t rows, int w) {
    // one thread per output element quad where W % 4 == 0 (16-byte stores, reversed 16-byte loads), else per element
    const size_t total = rows * (size_t)w;
    if ((w & 3) == 0) {
        const size_t quads = total >> 2;
        const int wq = w >> 2;
        for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
            const size_t row = q / wq;
            const int xq = (int)(q - row * wq);
            const float4 v = *reinterpret_cast<const float4*>(in + row * w + (size_t)(wq - 1 - xq) * 4);
            *reinterpret_cast<float4*>(out + q * 4) = make_float4(v.w, v.z, v.y, v.x);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const size_t row = i / w;
            const int x = (int)(i - row * w);
            out[i] = in[row * w + (w - 1 - x)];
        }
    }
}
}  // namespace
}  // namespace mp

extern "C" int mp_flip_width(const float* in, float* out, int n, int c, int h, int w, mp_stream_t stream) {
    if (!in || !out) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    if (in == out) return MP_ERR_UNSUPPORTED;
    const size_t rows = (size_t)n * c * h;
    const size_t work = (w & 3) == 0 ? rows * (size_t)w / 4 : rows * (size_t)w;
    size_t blocks = (work + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mp::flip_width_kernel, dim3((unsigned)blocks), dim3(256), 0, mp::as_stream(stream), in, out, rows, w);
    return mp::check_launch();
}
