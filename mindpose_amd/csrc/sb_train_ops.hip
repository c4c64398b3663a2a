// Training-side kernels that only the SimpleBaseline-ResNet family needs (the HRNet path has no max-pool, 7x7 stem or
// transposed convolution): max-pool backward, the weight gradient of the 3-channel 7x7 stride-2 stem, and the strided
// gather that feeds the sub-pixel phases of the transposed convolution's gradients.
#include "common.h"

#include <math.h>

namespace mp {

namespace {

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

// nn.MaxPool2d(3, 2, pad_mode="same") backward: every output window sends its gradient to its FIRST maximum in window scan
// order (row-major), as the arg-max form of the operator does.  One thread per INPUT element gathers from the <= 4 windows
// that cover it (no atomics, deterministic).
__global__ __launch_bounds__(256) void maxpool3x3s2_same_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    float* __restrict__ dx, int planes, int h, int w, int oh,
                                                                    int ow, int pt, int pl) {
    const size_t total = (size_t)planes * h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w);
        const size_t t = i / w;
        const int yy = (int)(t % h);
        const size_t pl_i = t / h;
        const float* src = x + pl_i * (size_t)h * w;
        const float* g = dy + pl_i * (size_t)oh * ow;
        float acc = 0.f;
        // windows (oy, ox) with oy*2 - pt <= yy <= oy*2 - pt + 2
        for (int oy = (yy + pt - 2 + 1) / 2 > 0 ? (yy + pt - 2 + 1) / 2 : 0; oy < oh && oy * 2 - pt <= yy; ++oy) {
            for (int ox = (xx + pl - 2 + 1) / 2 > 0 ? (xx + pl - 2 + 1) / 2 : 0; ox < ow && ox * 2 - pl <= xx; ++ox) {
                // first maximum of this window in scan order
                float m = -INFINITY;
                int my = -1, mx = -1;
                for (int dy_ = 0; dy_ < 3; ++dy_) {
                    const int y2 = oy * 2 - pt + dy_;
                    if (y2 < 0 || y2 >= h) continue;
                    for (int dx_ = 0; dx_ < 3; ++dx_) {
                        const int x2 = ox * 2 - pl + dx_;
                        if (x2 < 0 || x2 >= w) continue;
                        const float v = src[y2 * w + x2];
                        if (v > m) { m = v; my = y2; mx = x2; }
                    }
                }
                if (my == yy && mx == xx) acc += g[oy * ow + ox];
            }
        }
        dx[i] = acc;
    }
}

// Weight gradient of the stem conv (k x k, stride 2, padding k/2, <= 4 input channels) by direct reduction:
//   dW[co][ci][ky][kx] = sum_{n,y,x} dz[n,co,y,x] * x[n,ci,2y+ky-p,2x+kx-p]
// block = (co, ci, ky): 256 threads stride over (n, y, x), each keeps the k partial sums of its kx row in registers
// (the dz value and the k + 1 neighbouring input pixels are loaded once); fixed-order block reduction (deterministic).
template <int KS>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                         float* __restrict__ dw, int n, int cin, int h, int w, int cout, int ho,
                                                         int wo) {
    constexpr int P = KS / 2;
    const int ky = blockIdx.x % KS, ci = (blockIdx.x / KS) % cin, co = blockIdx.x / (KS * cin);
    float part[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) part[k] = 0.f;
    const int total = n * ho * wo;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int ox = i % wo, t = i / wo, oy = t % ho, img = t / ho;
        const int yin = oy * 2 + ky - P;
        if (yin < 0 || yin >= h) continue;
        const float g = dz[((size_t)(img * cout + co) * ho + oy) * wo + ox];
        const float* row = x + ((size_t)(img * cin + ci) * h + yin) * w;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int xin = ox * 2 + kx - P;
            if (xin >= 0 && xin < w) part[kx] += g * row[xin];
        }
    }
    __shared__ double sm[256];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
        sm[threadIdx.x] = (double)part[kx];
        __syncthreads();
        for (int s = 128; s >= 1; s >>= 1) {
            if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) dw[(((size_t)co * cin + ci) * KS + ky) * KS + kx] = (float)sm[0];
        __syncthreads();
    }
}

// out[n, blk, m, k] = x[n, blk, 2m + py, 2k + px]  (channel-blocked fp16, 16-byte elements): one sub-pixel phase of a
// gradient at the up-sampled resolution, made contiguous so that the 2x2 phase kernels can consume it
__global__ __launch_bounds__(256) void gather_phase_c8_kernel(const u32x4s* __restrict__ x, u32x4s* __restrict__ out, int planes,
                                                              int h, int w, int py, int px) {
    const size_t total = (size_t)planes * h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % w);
        const size_t t = i / w;
        const int m = (int)(t % h);
        const size_t pl = t / h;
        out[i] = x[(pl * (2 * h) + 2 * m + py) * (size_t)(2 * w) + 2 * k + px];
    }
}

// fp32 NCHW form of the phase gather: out[pl, m, k] = x[pl, 2m + py, 2k + px]
__global__ __launch_bounds__(256) void gather_phase_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int planes, int h,
                                                               int w, int py, int px) {
    const size_t total = (size_t)planes * h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % w);
        const size_t t = i / w;
        const int m = (int)(t % h);
        const size_t pl = t / h;
        out[i] = x[(pl * (2 * h) + 2 * m + py) * (size_t)(2 * w) + 2 * k + px];
    }
}

}  // namespace
}  // namespace mp

using namespace mp;

extern "C" {

int mp_maxpool3x3s2_same_bwd(const float* x, const float* dy, float* dx, int n, int c, int h, int w, mp_stream_t stream) {
    if (!x || !dy || !dx) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    const int oh = (h + 1) / 2, ow = (w + 1) / 2;
    const int ph = (oh - 1) * 2 + 3 - h > 0 ? (oh - 1) * 2 + 3 - h : 0, pw = (ow - 1) * 2 + 3 - w > 0 ? (ow - 1) * 2 + 3 - w : 0;
    const size_t total = (size_t)n * c * h * w;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(maxpool3x3s2_same_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, dy, dx, n * c, h, w, oh,
                       ow, ph / 2, pw / 2);
    return check_launch();
}

int mp_stem_conv_wgrad(const float* x, const float* dz, float* dw, int n, int cin, int h, int w, int cout, int k, mp_stream_t stream) {
    if (!x || !dz || !dw) return MP_ERR_NULL;
    if (n <= 0 || cin <= 0 || cin > 4 || cout <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    if (k != 7 && k != 3) return MP_ERR_UNSUPPORTED;
    const int ho = (h + 2 * (k / 2) - k) / 2 + 1, wo = (w + 2 * (k / 2) - k) / 2 + 1;
    if ((long long)n * ho * wo >= 0x7FFFFFFFLL) return MP_ERR_UNSUPPORTED;
    const dim3 grid(cout * cin * k);
    if (k == 7) hipLaunchKernelGGL(stem_wgrad_kernel<7>, grid, dim3(256), 0, as_stream(stream), x, dz, dw, n, cin, h, w, cout, ho, wo);
    else hipLaunchKernelGGL(stem_wgrad_kernel<3>, grid, dim3(256), 0, as_stream(stream), x, dz, dw, n, cin, h, w, cout, ho, wo);
    return check_launch();
}

int mp_gather_phase(const float* x, float* out, int n, int c, int h, int w, int phase_y, int phase_x, mp_stream_t stream) {
    if (!x || !out) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || phase_y < 0 || phase_y > 1 || phase_x < 0 || phase_x > 1) return MP_ERR_SHAPE;
    const size_t total = (size_t)n * c * h * w;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(gather_phase_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, out, n * c, h, w, phase_y,
                       phase_x);
    return check_launch();
}

int mp_f16_gather_phase(const void* x, void* out, int n, int c, int h, int w, int phase_y, int phase_x, mp_stream_t stream) {
    if (!x || !out) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || phase_y < 0 || phase_y > 1 || phase_x < 0 || phase_x > 1) return MP_ERR_SHAPE;
    const int planes = n * ((c + 7) / 8);
    const size_t total = (size_t)planes * h * w;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(gather_phase_c8_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<const u32x4s*>(x),
                       reinterpret_cast<u32x4s*>(out), planes, h, w, phase_y, phase_x);
    return check_launch();
}

}  // extern "C"
