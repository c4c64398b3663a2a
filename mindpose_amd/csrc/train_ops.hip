// Training-side HBM-bound kernels (gfx950): BatchNorm2d in training mode (batch statistics, forward and
// backward), the backward of the exchange-unit sum, and the AdamWeightDecay update.  All reductions are
// two-stage with a fixed partition and a fixed combine order, so results are bit-reproducible run to run.
#include "common.h"

#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

namespace mp {

constexpr int kBnSplit = 32;  // per-channel partial reductions (fixed: determinism)
constexpr int kBn16MaxSplit = 256;

__device__ __forceinline__ double block_sum_256(double v, double* sm) {
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

// stage 1: grid (C, kBnSplit); block (c, s) reduces images n = s, s + kBnSplit, ... of channel c.
// out: part[(c * kBnSplit + s) * 2 + {0,1}] = (sum a, sum a*b) in fp64, where
//   forward  a = z,            b = z            -> sum z, sum z^2
//   backward a = g (masked dy), b = xhat         -> sum g, sum g*xhat
template <bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ a_in, const float* __restrict__ z,
                                                        const float* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, double* __restrict__ part,
                                                        int n, int c, int hw, int relu) {
    const int ch = blockIdx.x, sp = blockIdx.y;
    double s0 = 0.0, s1 = 0.0;
    const float mu = BWD ? mean[ch] : 0.f, is = BWD ? invstd[ch] : 0.f;
    for (int img = sp; img < n; img += kBnSplit) {
        const size_t base = ((size_t)img * c + ch) * hw;
        float f0 = 0.f, f1 = 0.f;  // fp32 per-thread partial over <= hw/256 elements, then fp64
        if ((hw & 3) == 0) {  // planes are 16-byte multiples: float4 loads (same element-to-thread order whatever the width:
                              // thread t takes elements 4q .. 4q+3 for q = t, t + 256, ...)
            for (int q = threadIdx.x; q < (hw >> 2); q += 256) {
                const float4 zv = reinterpret_cast<const float4*>(z + base)[q];
                if (BWD) {
                    float4 g = reinterpret_cast<const float4*>(a_in + base)[q];
                    if (relu) {
                        const float4 yv = reinterpret_cast<const float4*>(y + base)[q];
                        if (!(yv.x > 0.f)) g.x = 0.f;
                        if (!(yv.y > 0.f)) g.y = 0.f;
                        if (!(yv.z > 0.f)) g.z = 0.f;
                        if (!(yv.w > 0.f)) g.w = 0.f;
                    }
                    f0 += g.x; f1 += g.x * ((zv.x - mu) * is);
                    f0 += g.y; f1 += g.y * ((zv.y - mu) * is);
                    f0 += g.z; f1 += g.z * ((zv.z - mu) * is);
                    f0 += g.w; f1 += g.w * ((zv.w - mu) * is);
                } else {
                    f0 += zv.x; f1 += zv.x * zv.x;
                    f0 += zv.y; f1 += zv.y * zv.y;
                    f0 += zv.z; f1 += zv.z * zv.z;
                    f0 += zv.w; f1 += zv.w * zv.w;
                }
            }
        } else {
            for (int i = threadIdx.x; i < hw; i += 256) {
                if (BWD) {
                    float g = a_in[base + i];
                    if (relu && !(y[base + i] > 0.f)) g = 0.f;
                    const float xh = (z[base + i] - mu) * is;
                    f0 += g;
                    f1 += g * xh;
                } else {
                    const float v = z[base + i];
                    f0 += v;
                    f1 += v * v;
                }
            }
        }
        s0 += (double)f0;
        s1 += (double)f1;
    }
    __shared__ double sm[256];
    s0 = block_sum_256(s0, sm);
    s1 = block_sum_256(s1, sm);
    if (threadIdx.x == 0) {
        part[((size_t)ch * kBnSplit + sp) * 2 + 0] = s0;
        part[((size_t)ch * kBnSplit + sp) * 2 + 1] = s1;
    }
}

// stage 2 (folded into the apply kernels below): per channel mean / biased variance -> scale/shift, saved stats, moving-average
// update.  mindspore.nn.BatchNorm2d(momentum=0.9): moving = 0.9 * moving + 0.1 * batch; the moving variance takes the UNBIASED batch
// variance (cuDNN / PyTorch convention) [MS-knowledge, unverifiable here - affects only the moving statistics, never the
// training-mode output or any gradient].
// totals of one channel from its kBnSplit partials: the first wave of the block, lane l = split l, fixed-order shuffle tree
__device__ __forceinline__ void bn_channel_sums(const double* __restrict__ part, int ch, double& s0, double& s1) {
    s0 = 0.0;
    s1 = 0.0;
    if (threadIdx.x < kBnSplit) {
        const double2 v = *reinterpret_cast<const double2*>(part + ((size_t)ch * kBnSplit + threadIdx.x) * 2);
        s0 = v.x;
        s1 = v.y;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        s1 += __shfl_down(s1, off, 64);
    }
}

// y = act(z * scale[c] + shift[c] (+ res)) - block per (n, c) plane, 16 B per lane when hw % 4 == 0.  Stage 2 is folded in: every
// block reduces its channel's 32 partials itself (same values, same order in every block); the block of image 0 also writes the
// saved statistics and the moving averages - one launch less per BatchNorm and direction.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const double* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                       float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                       const float* __restrict__ res, float* __restrict__ y, int c, int hw, int relu,
                                                       double count, float eps, float momentum) {
    const size_t plane = blockIdx.x;
    const int ch = (int)(plane % c);
    __shared__ float s_sc, s_sh;
    if (threadIdx.x < 64) {
        double s0, s1;
        bn_channel_sums(part, ch, s0, s1);
        if (threadIdx.x == 0) {
            const double mean = s0 / count;
            double var = s1 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float a = gamma[ch] * invstd;
            s_sc = a;
            s_sh = beta[ch] - (float)mean * a;
            if (plane < (size_t)c) {  // image 0
                save_mean[ch] = (float)mean;
                save_invstd[ch] = invstd;
                if (moving_mean) {
                    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                    moving_mean[ch] = momentum * moving_mean[ch] + (1.f - momentum) * (float)mean;
                    moving_var[ch] = momentum * moving_var[ch] + (1.f - momentum) * (float)unbiased;
                }
            }
        }
    }
    __syncthreads();
    const float sc = s_sc, sh = s_sh;
    const size_t base = plane * hw;
    if ((hw & 3) == 0) {
        for (int q = threadIdx.x; q < (hw >> 2); q += 256) {
            float4 v = reinterpret_cast<const float4*>(z + base)[q];
            v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
            if (res) { const float4 r = reinterpret_cast<const float4*>(res + base)[q]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            reinterpret_cast<float4*>(y + base)[q] = v;
        }
    } else {
        for (int i = threadIdx.x; i < hw; i += 256) {
            float v = z[base + i] * sc + sh;
            if (res) v += res[base + i];
            if (relu) v = fmaxf(v, 0.f);
            y[base + i] = v;
        }
    }
}

// dz = gamma*invstd * (g - dbeta/M - xhat*dgamma/M), dres = g  (g = dy masked by the ReLU of the forward output); stage 2 folded
// in like the forward (the block of image 0 writes dgamma / dbeta and adds them into the caller's gradient buffers)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                           const float* __restrict__ y, const double* __restrict__ part,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ dgamma_acc,
                                                           float* __restrict__ dbeta_acc, float* __restrict__ dz,
                                                           float* __restrict__ dres, int c, int hw, int relu, float inv_count) {
    const size_t plane = blockIdx.x;
    const int ch = (int)(plane % c);
    __shared__ float s_mb, s_mg;
    if (threadIdx.x < 64) {
        double s0, s1;
        bn_channel_sums(part, ch, s0, s1);
        if (threadIdx.x == 0) {
            const float db = (float)s0, dg = (float)s1;
            s_mb = db * inv_count;
            s_mg = dg * inv_count;
            if (plane < (size_t)c) {  // image 0
                dbeta[ch] = db;
                dgamma[ch] = dg;
                if (dgamma_acc && dbeta_acc) {  // + straight into the caller's gradient buffers (no separate add launch)
                    dbeta_acc[ch] += db;
                    dgamma_acc[ch] += dg;
                }
            }
        }
    }
    __syncthreads();
    const float mu = mean[ch], is = invstd[ch];
    const float k = gamma[ch] * is, mb = s_mb, mg = s_mg;
    const size_t base = plane * hw;
    if ((hw & 3) == 0) {
        for (int q = threadIdx.x; q < (hw >> 2); q += 256) {
            float4 g = reinterpret_cast<const float4*>(dy + base)[q];
            const float4 zv = reinterpret_cast<const float4*>(z + base)[q];
            if (relu) {
                const float4 yv = reinterpret_cast<const float4*>(y + base)[q];
                if (!(yv.x > 0.f)) g.x = 0.f;
                if (!(yv.y > 0.f)) g.y = 0.f;
                if (!(yv.z > 0.f)) g.z = 0.f;
                if (!(yv.w > 0.f)) g.w = 0.f;
            }
            float4 d;
            d.x = k * (g.x - mb - ((zv.x - mu) * is) * mg);
            d.y = k * (g.y - mb - ((zv.y - mu) * is) * mg);
            d.z = k * (g.z - mb - ((zv.z - mu) * is) * mg);
            d.w = k * (g.w - mb - ((zv.w - mu) * is) * mg);
            reinterpret_cast<float4*>(dz + base)[q] = d;
            if (dres) reinterpret_cast<float4*>(dres + base)[q] = g;
        }
    } else {
        for (int i = threadIdx.x; i < hw; i += 256) {
            float g = dy[base + i];
            if (relu && !(y[base + i] > 0.f)) g = 0.f;
            const float xh = (z[base + i] - mu) * is;
            dz[base + i] = k * (g - mb - xh * mg);
            if (dres) dres[base + i] = g;
        }
    }
}

// backward of out = act(base + sum_k up_{s_k}(t_k)):  g = dy * (out > 0);  dbase = g;  dt_k = s_k x s_k block sums of g
__global__ __launch_bounds__(256) void fuse_sum_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out,
                                                           float* __restrict__ dbase, float* __restrict__ dt, int planes,
                                                           int h, int w, int sh, int relu) {
    // one thread per element of the LOW-resolution grid of this term (sh = log2 scale); sh == 0 covers dbase-like terms
    const int lh = h >> sh, lw = w >> sh, s = 1 << sh;
    const size_t total = (size_t)planes * lh * lw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lx = (int)(i % lw);
        const size_t r = i / lw;
        const int ly = (int)(r % lh);
        const size_t plane = r / lh;
        const size_t o = (plane * h + (size_t)ly * s) * w + (size_t)lx * s;
        float acc = 0.f;
        for (int a = 0; a < s; ++a)
            for (int b = 0; b < s; ++b) {
                float g = dy[o + (size_t)a * w + b];
                if (relu && !(out[o + (size_t)a * w + b] > 0.f)) g = 0.f;
                acc += g;
                if (dbase && sh == 0) dbase[o] = g;
            }
        if (dt) dt[i] = acc;
    }
}

// mindspore.nn.AdamWeightDecay: Adam WITHOUT bias correction, decoupled weight decay, eps added to sqrt(v)
// (mindpose/optim/optim_factory.py:69-72 selects it for "adamw")
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                    float wd) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        float upd = mi / (sqrtf(vi) + eps);
        upd += wd * p[i];
        p[i] = p[i] - lr * upd;
    }
}

// The other optimizers the reference registers (mindpose/optim/optim_factory.py:9-14), same flat-arena form.  Update rules as
// documented for mindspore.nn.{Adam, SGD, Momentum, Adagrad} [MS-knowledge: no reference test pins them]; the gradient is
// first divided by the static loss scale and given the L2 term (g = g * grad_scale + wd * p), as those optimizers do.
//   kind 1 Adam:     m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
//        (h: b1, b2, eps, b1^t, b2^t)
//   kind 2 SGD:      buf = g on the first step, else mom buf + (1-damp) g; d = nesterov ? g + mom buf : buf (d = g when
//        mom == 0); p -= lr d      (h: mom, damp, nesterov, first_step)
//   kind 3 Momentum: acc = mom acc + g; p -= lr (nesterov ? g + mom acc : acc)      (h: mom, nesterov)
//   kind 4 Adagrad:  acc += g^2; p -= lr g / sqrt(acc)      (accumulator initialised by the caller, 0.1 in MindSpore)
template <int KIND>
__global__ __launch_bounds__(256) void optimizer_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ s1,
                                                        float* __restrict__ s2, size_t n, float lr, float grad_scale, float wd,
                                                        float h0, float h1, float h2, float h3, float h4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float w = p[i];
        const float gi = g[i] * grad_scale + wd * w;
        if (KIND == 1) {
            const float mi = h0 * s1[i] + (1.f - h0) * gi;
            const float vi = h1 * s2[i] + (1.f - h1) * gi * gi;
            s1[i] = mi;
            s2[i] = vi;
            const float lr_t = lr * sqrtf(1.f - h4) / (1.f - h3);
            p[i] = w - lr_t * mi / (sqrtf(vi) + h2);
        } else if (KIND == 2) {
            float d = gi;
            if (h0 != 0.f) {
                const float buf = h3 != 0.f ? gi : h0 * s1[i] + (1.f - h1) * gi;
                s1[i] = buf;
                d = h2 != 0.f ? gi + h0 * buf : buf;
            }
            p[i] = w - lr * d;
        } else if (KIND == 3) {
            const float acc = h0 * s1[i] + gi;
            s1[i] = acc;
            p[i] = w - lr * (h1 != 0.f ? gi + h0 * acc : acc);
        } else {
            const float acc = s1[i] + gi * gi;
            s1[i] = acc;
            p[i] = w - lr * gi / sqrtf(acc);
        }
    }
}

// ---- fp16 (amp O2) training: the same BatchNorm passes over channel-blocked fp16 activations [N][C8][HW][8] ------------
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// The forward output is y = half(max(fma(z, sc, sh), 0)) with sc = gamma * invstd, sh = fma(-mean, sc, beta): these two helpers are
// the ONLY place that arithmetic is written, because the backward pass of a ReLU layer WITHOUT a residual input re-derives the
// ReLU mask (y > 0) from z with them instead of reading y (relu mode 2: two tensor reads less per backward BatchNorm).
// 16-bit lanes of channel block blk that hold real channels (c8 layout: channel 8 blk + j in half j)
__device__ __forceinline__ u32x4_t bn16_channel_mask(int blk, int c) {
    u32x4_t m;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        m[w] = (blk * 8 + 2 * w < c ? 0x0000FFFFu : 0u) | (blk * 8 + 2 * w + 1 < c ? 0xFFFF0000u : 0u);
    return m;
}
__device__ __forceinline__ float bn16_shift(float beta, float mean, float sc) { return __builtin_fmaf(-mean, sc, beta); }
__device__ __forceinline__ float bn16_affine(float z, float sc, float sh) { return __builtin_fmaf(z, sc, sh); }
__device__ __forceinline__ bool bn16_relu_open(float z, float sc, float sh) { return (float)(_Float16)bn16_affine(z, sc, sh) > 0.f; }

// fixed-order block reduction of 16 doubles per thread (wave shuffles, then the four waves through LDS)
__device__ __forceinline__ void block_sum16_256(double (&v)[16], double (*sm)[16]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double x = v[j];
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off, 64);
        v[j] = x;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) sm[wave][j] = v[j];
    }
    __syncthreads();
    if (threadIdx.x < 16) v[0] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    __syncthreads();
}

// grid (C8, kBnSplit): block (blk, sp) reduces images sp, sp + kBnSplit, ... of the 8 channels of block blk; partials in the
// fp32 kernels' layout part[(ch * kBnSplit + sp) * 2 + {0,1}] so that the finalize kernels are shared
template <bool BWD>
__global__ __launch_bounds__(256) void bn16_reduce_kernel(const u32x4_t* __restrict__ a_in, const u32x4_t* __restrict__ z,
                                                          const u32x4_t* __restrict__ y, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, double* __restrict__ part, int n,
                                                          int c, int c8, int hw, int relu, int gi, int gp) {
    // split sp = (image group sp % gi, pixel chunk sp / gi): enough blocks to fill the chip even for 4 channel blocks
    const int blk = blockIdx.x, sp = blockIdx.y, nsplit = gi * gp;
    const int ig = sp % gi, pc = sp / gi;
    const int chunk = (hw + gp - 1) / gp, p0 = pc * chunk, p1 = min(p0 + chunk, hw);
    float mu[8], is[8], msc[8], msh[8];  // msc / msh: forward scale / shift, for the recomputed ReLU mask (relu == 2)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = blk * 8 + j;
        mu[j] = (BWD && ch < c) ? mean[ch] : 0.f;
        is[j] = (BWD && ch < c) ? invstd[ch] : 0.f;
        msc[j] = (BWD && relu == 2 && ch < c) ? gamma[ch] * is[j] : 0.f;
        msh[j] = (BWD && relu == 2 && ch < c) ? bn16_shift(beta[ch], mu[j], msc[j]) : 0.f;
    }
    // The block's elements are {images ig, ig + gi, ...} x {pixels p0 .. p1}: walked as ONE flat index space (the small maps give a
    // block only 1-2 elements per thread and image; an image loop around a pixel loop kept a single load in flight per thread and
    // made these passes latency-bound), 32-bit index arithmetic with a multiply-high division by the chunk length.
    double acc[16];
    float f[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = 0.f;
    const unsigned len = (unsigned)(p1 > p0 ? p1 - p0 : 0);
    const unsigned nimg = ig < n ? (unsigned)((n - ig + gi - 1) / gi) : 0u;
    const unsigned cnt = len * nimg;
    const unsigned magic = len > 1 ? (unsigned)(0x100000000ULL / len) + 1u : 0u;
    const bool exact = (unsigned long long)cnt * len < 0x100000000ULL;  // multiply-high division exact on [0, cnt)
    const size_t img_stride = (size_t)c8 * hw;
#pragma unroll 4
    for (unsigned e = threadIdx.x; e < cnt; e += 256) {
        const unsigned a = len <= 1 ? e : (exact ? __umulhi(e, magic) : e / len);
        const size_t i = ((size_t)(ig + a * gi)) * img_stride + (size_t)blk * hw + p0 + (e - a * len);
        const h16x8 zv = __builtin_bit_cast(h16x8, z[i]);
        if (BWD) {
            const h16x8 gv = __builtin_bit_cast(h16x8, a_in[i]);
            h16x8 yv = zv;
            if (relu == 1) yv = __builtin_bit_cast(h16x8, y[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float g = (float)gv[j];
                if (relu == 1 && !((float)yv[j] > 0.f)) g = 0.f;
                if (relu == 2 && !bn16_relu_open((float)zv[j], msc[j], msh[j])) g = 0.f;
                const float xh = ((float)zv[j] - mu[j]) * is[j];
                f[2 * j] += g;
                f[2 * j + 1] += g * xh;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = (float)zv[j];
                f[2 * j] += v;
                f[2 * j + 1] += v * v;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = (double)f[j];
    __shared__ double sm[4][16];
    block_sum16_256(acc, sm);
    if (threadIdx.x < 16) {
        const int ch = blk * 8 + (threadIdx.x >> 1);
        if (ch < c) part[((size_t)ch * nsplit + sp) * 2 + (threadIdx.x & 1)] = acc[0];
    }
}

// Stage 2 folded into the consumers: a block of the apply kernels serves ONE 8-channel block (grid = (C8, chunks of the
// image x pixel range)) and first reduces that block's partial sums itself - thread t: channel t >> 5, splits t & 31, + 32, ...
// then a 32-lane shuffle tree; every block computes the same values in the same order (deterministic) - which removes the
// ~5 us stage-2 launch between the two passes of every BatchNorm in both directions (6 % of the amp-O2 step).  Chunk 0 of a
// channel block also writes the per-channel results (saved statistics / moving averages, or dgamma / dbeta).
__device__ __forceinline__ void bn16_channel_sums(const double* __restrict__ part, int ch, int c, int nsplit, double& s0, double& s1) {
    s0 = 0.0;
    s1 = 0.0;
    if (ch < c) {
        // fixed trip count (kBn16MaxSplit / 32): all loads issue before the first add - the same sums in the same order as a
        // `sp < nsplit` loop, without its chain of dependent L2 round trips at the head of every consumer block
        double v0[kBn16MaxSplit / 32], v1[kBn16MaxSplit / 32];
#pragma unroll
        for (int k = 0; k < kBn16MaxSplit / 32; ++k) {
            const int sp = (threadIdx.x & 31) + 32 * k;
            const bool ok = sp < nsplit;
            const double2 v = ok ? *reinterpret_cast<const double2*>(part + ((size_t)ch * nsplit + sp) * 2) : make_double2(0.0, 0.0);
            v0[k] = v.x;
            v1[k] = v.y;
        }
#pragma unroll
        for (int k = 0; k < kBn16MaxSplit / 32; ++k) {
            if ((threadIdx.x & 31) + 32 * k < nsplit) {
                s0 += v0[k];
                s1 += v1[k];
            }
        }
    }
    for (int off = 16; off >= 1; off >>= 1) {  // xor tree inside each 32-lane half: every lane ends with the total
        s0 += __shfl_xor(s0, off, 64);
        s1 += __shfl_xor(s1, off, 64);
    }
}

// y = act(z * scale[c] + shift[c] (+ res)) over c8 elements; padding channels stay zero
__global__ __launch_bounds__(256) void bn16_apply_kernel(const u32x4_t* __restrict__ z, const double* __restrict__ part,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                         float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                         const u32x4_t* __restrict__ res, u32x4_t* __restrict__ y, int n, int c,
                                                         int c8, int hw, int nsplit, double count, float eps, float momentum,
                                                         int relu) {
    const int blk = blockIdx.x;
    __shared__ float s_scale[8], s_shift[8];
    {
        const int j = threadIdx.x >> 5, ch = blk * 8 + j;
        double s0, s1;
        bn16_channel_sums(part, ch, c, nsplit, s0, s1);
        if ((threadIdx.x & 31) == 0) {
            float sc = 0.f, sh = 0.f;
            if (ch < c) {
                const double mean = s0 / count;
                double var = s1 / count - mean * mean;
                if (var < 0.0) var = 0.0;
                const float invstd = (float)(1.0 / sqrt(var + (double)eps));
                sc = gamma[ch] * invstd;
                sh = bn16_shift(beta[ch], (float)mean, sc);
                if (blockIdx.y == 0) {
                    save_mean[ch] = (float)mean;
                    save_invstd[ch] = invstd;
                    if (moving_mean) {
                        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                        moving_mean[ch] = momentum * moving_mean[ch] + (1.f - momentum) * (float)mean;
                        moving_var[ch] = momentum * moving_var[ch] + (1.f - momentum) * (float)unbiased;
                    }
                }
            }
            s_scale[j] = sc;
            s_shift[j] = sh;
        }
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_scale[j]; sh[j] = s_shift[j]; }
    // elements of this channel block: (image, pixel) flat, 32-bit, image = multiply-high division by hw (64-bit divisions per
    // element cost more than the element's memory traffic)
    const unsigned per_blk = (unsigned)n * (unsigned)hw;
    const unsigned len = (per_blk + gridDim.y - 1) / gridDim.y;
    const unsigned e0 = blockIdx.y * len, e1 = e0 + len < per_blk ? e0 + len : per_blk;
    const unsigned magic_hw = hw > 1 ? (unsigned)(0x100000000ULL / (unsigned)hw) + 1u : 0u;
    const bool exact = (unsigned long long)per_blk * (unsigned)hw < 0x100000000ULL;
    const size_t blk_off = (size_t)blk * hw, img_extra = (size_t)(c8 - 1) * hw;
#pragma unroll 4
    for (unsigned e = e0 + threadIdx.x; e < e1; e += 256) {
        const unsigned img = hw <= 1 ? e : (exact ? __umulhi(e, magic_hw) : e / (unsigned)hw);
        const size_t i = (size_t)e + (size_t)img * img_extra + blk_off;
        const h16x8 zv = __builtin_bit_cast(h16x8, z[i]);
        h16x8 rv = zv;
        if (res) rv = __builtin_bit_cast(h16x8, res[i]);
        h16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (blk * 8 + j < c) {
                v = bn16_affine((float)zv[j], sc[j], sh[j]);
                if (res) v += (float)rv[j];
                if (relu) v = fmaxf(v, 0.f);
            }
            o[j] = (_Float16)v;
        }
        y[i] = __builtin_bit_cast(u32x4_t, o);
    }
}

__global__ __launch_bounds__(256) void bn16_bwd_apply_kernel(const u32x4_t* __restrict__ dy, const u32x4_t* __restrict__ z,
                                                             const u32x4_t* __restrict__ y, const double* __restrict__ part,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dgamma_acc,
                                                             float* __restrict__ dbeta_acc, u32x4_t* __restrict__ dz,
                                                             u32x4_t* __restrict__ dres, int n, int c, int c8, int hw, int nsplit,
                                                             int relu, float inv_count, const float* __restrict__ beta) {
    const int blk = blockIdx.x;
    __shared__ float s_k[8], s_mu[8], s_is[8], s_mb[8], s_mg[8], s_sh[8];
    {
        const int j = threadIdx.x >> 5, ch = blk * 8 + j;
        double s0, s1;
        bn16_channel_sums(part, ch, c, nsplit, s0, s1);
        if ((threadIdx.x & 31) == 0) {
            float k = 0.f, mu = 0.f, is = 0.f, mb = 0.f, mg = 0.f;
            if (ch < c) {
                const float db = (float)s0, dg = (float)s1;
                mu = mean[ch];
                is = invstd[ch];
                k = gamma[ch] * is;
                mb = db * inv_count;
                mg = dg * inv_count;
                if (blockIdx.y == 0) {
                    dbeta[ch] = db;
                    dgamma[ch] = dg;
                    if (dgamma_acc && dbeta_acc) {  // + straight into the caller's gradient buffers (no separate add launch)
                        dbeta_acc[ch] += db;
                        dgamma_acc[ch] += dg;
                    }
                }
            }
            s_k[j] = k; s_mu[j] = mu; s_is[j] = is; s_mb[j] = mb; s_mg[j] = mg;
            s_sh[j] = (relu == 2 && ch < c) ? bn16_shift(beta[ch], mu, k) : 0.f;  // k = gamma * invstd = the forward scale
        }
    }
    __syncthreads();
    float k[8], mu[8], is[8], mb[8], mg[8], msh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { k[j] = s_k[j]; mu[j] = s_mu[j]; is[j] = s_is[j]; mb[j] = s_mb[j]; mg[j] = s_mg[j]; msh[j] = s_sh[j]; }
    const unsigned per_blk = (unsigned)n * (unsigned)hw;
    const unsigned len = (per_blk + gridDim.y - 1) / gridDim.y;
    const unsigned e0 = blockIdx.y * len, e1 = e0 + len < per_blk ? e0 + len : per_blk;
    const unsigned magic_hw = hw > 1 ? (unsigned)(0x100000000ULL / (unsigned)hw) + 1u : 0u;
    const bool exact = (unsigned long long)per_blk * (unsigned)hw < 0x100000000ULL;
    const size_t blk_off = (size_t)blk * hw, img_extra = (size_t)(c8 - 1) * hw;
#pragma unroll 4
    for (unsigned e = e0 + threadIdx.x; e < e1; e += 256) {
        const unsigned img = hw <= 1 ? e : (exact ? __umulhi(e, magic_hw) : e / (unsigned)hw);
        const size_t i = (size_t)e + (size_t)img * img_extra + blk_off;
        const h16x8 gv = __builtin_bit_cast(h16x8, dy[i]);
        const h16x8 zv = __builtin_bit_cast(h16x8, z[i]);
        h16x8 yv = zv;
        if (relu == 1) yv = __builtin_bit_cast(h16x8, y[i]);
        h16x8 oz, og;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float g = 0.f, d = 0.f;
            if (blk * 8 + j < c) {
                g = (float)gv[j];
                if (relu == 1 && !((float)yv[j] > 0.f)) g = 0.f;
                if (relu == 2 && !bn16_relu_open((float)zv[j], k[j], msh[j])) g = 0.f;
                const float xh = ((float)zv[j] - mu[j]) * is[j];
                d = k[j] * (g - mb[j] - xh * mg[j]);
            }
            oz[j] = (_Float16)d;
            og[j] = (_Float16)g;
        }
        dz[i] = __builtin_bit_cast(u32x4_t, oz);
        if (dres) dres[i] = __builtin_bit_cast(u32x4_t, og);
    }
}

// ---- statistics from the conv epilogue (round 3): the apply passes alone -----------------------------------------------------------
// The conv launch that produced z (forward) or the pre-masked gradient g (backward) left per-workgroup fp32 partial sums
// pre[blk][n_parts][8 channels][2] (conv_f16_dev.h); the reduction launches above do not run at all.  A block of the apply
// kernels folds the n_parts x 64 bytes of ITS channel block itself (contiguous: coalesced 16-byte loads, fp64 sums in a fixed
// order, every block the same values), kMaxFoldParts slots at most - above that bn16_fold_kernel reduces them to one slot first.
constexpr int kMaxFoldParts = 512;
constexpr int kFoldSplit = 8;  // slot ranges of the fold launch (more than kMaxFoldParts slots)

// totals of the 8 channels of block blk -> tot[16] = {sum0, sumsq0, sum1, ...} (fp64, LDS); ends with a barrier
__device__ __forceinline__ void bn16_fold_parts(const float* __restrict__ pre, int blk, int n_parts, double* tot, double (*sm)[16]) {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(pre + (size_t)blk * n_parts * 16);
    const int q = threadIdx.x & 3, r = threadIdx.x >> 2;  // float4 q of slot r, r + 64, ...
    float4 v[kMaxFoldParts / 64];
#pragma unroll
    for (int k = 0; k < kMaxFoldParts / 64; ++k) {
        const int slot = r + 64 * k;
        v[k] = slot < n_parts ? src[(size_t)slot * 4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < kMaxFoldParts / 64; ++k) {
        a[0] += (double)v[k].x; a[1] += (double)v[k].y; a[2] += (double)v[k].z; a[3] += (double)v[k].w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        for (int off = 4; off <= 32; off <<= 1) a[i] += __shfl_xor(a[i], off, 64);  // over the 16 slot rows of the wave, q fixed
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sm[wave][lane * 4 + i] = a[i];
    }
    __syncthreads();
    if (threadIdx.x < 16) tot[threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    __syncthreads();
}

// n_parts > kMaxFoldParts: grid (C8, kFoldSplit), block (blk, f) folds slot range f of its channel block (fixed order) into slot f of
// `out` ([C8][kFoldSplit][16]); the consumers then fold kFoldSplit slots
__global__ __launch_bounds__(256) void bn16_fold_kernel(const float* __restrict__ pre, float* __restrict__ out, int n_parts) {
    const int blk = blockIdx.x, f = blockIdx.y;
    const int per = (n_parts + kFoldSplit - 1) / kFoldSplit, s0 = f * per, s1 = min(s0 + per, n_parts);
    const float4* __restrict__ src = reinterpret_cast<const float4*>(pre + (size_t)blk * n_parts * 16);
    const int q = threadIdx.x & 3, r = threadIdx.x >> 2;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int slot = s0 + r; slot < s1; slot += 64) {
        const float4 v = src[(size_t)slot * 4 + q];
        a[0] += (double)v.x; a[1] += (double)v.y; a[2] += (double)v.z; a[3] += (double)v.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        for (int off = 4; off <= 32; off <<= 1) a[i] += __shfl_xor(a[i], off, 64);
    __shared__ double sm[4][16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sm[wave][lane * 4 + i] = a[i];
    }
    __syncthreads();
    // fp32 slots: a range total (|sum| < 2^24 x its mean term) keeps 24 bits - the precision the conv's slots themselves have
    if (threadIdx.x < 16)
        out[((size_t)blk * kFoldSplit + f) * 16 + threadIdx.x] =
            (float)(((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x]);
}

// channel j of block blk: batch mean / variance from the folded sums -> the forward scale / shift (0 / 0 on a padding channel).
// gamma_v / beta_v were requested at the top of the kernel (their latency runs under the fold's); the saved statistics and the
// moving averages are written by `bn16_fwd_publish` AFTER the caller has released scale / shift to the other threads - the
// read-modify-write of the moving averages is a memory round trip nobody should wait for
struct Bn16Coeffs { float sc, sh, mean, invstd, var_unbiased; };
__device__ __forceinline__ Bn16Coeffs bn16_fwd_coeffs(int j, int blk, int c, const double* s_tot, float gamma_v, float beta_v, double inv_count,
                                                      double unbias, float eps) {
    Bn16Coeffs o = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (blk * 8 + j < c) {
        const double mean = s_tot[2 * j] * inv_count;
        double var = s_tot[2 * j + 1] * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        // 1 / sqrt(var + eps): fp32 estimate + one Newton step in fp64 (relative error ~1e-14, no fp64 divide / sqrt sequence)
        const double x = var + (double)eps;
        double r = (double)rsqrtf((float)x);
        r = r * (1.5 - 0.5 * x * r * r);
        o.invstd = (float)r;
        o.mean = (float)mean;
        o.var_unbiased = (float)(var * unbias);
        o.sc = gamma_v * o.invstd;
        o.sh = bn16_shift(beta_v, o.mean, o.sc);
    }
    return o;
}
__device__ __forceinline__ void bn16_fwd_publish(int j, int blk, int c, const Bn16Coeffs& o, float* __restrict__ save_mean,
                                                 float* __restrict__ save_invstd, float* __restrict__ moving_mean,
                                                 float* __restrict__ moving_var, float momentum) {
    const int ch = blk * 8 + j;
    if (ch < c) {
        save_mean[ch] = o.mean;
        save_invstd[ch] = o.invstd;
        if (moving_mean) {
            moving_mean[ch] = momentum * moving_mean[ch] + (1.f - momentum) * o.mean;
            moving_var[ch] = momentum * moving_var[ch] + (1.f - momentum) * o.var_unbiased;
        }
    }
}

// The prologue of the apply pass as a launch of its own, grid (C8): the statistics of a mode-1 conv launch -> the folded scale /
// shift its CONSUMER applies on its own operand (conv_f16_wreg.hip, PRE) - same fold, same arithmetic, same saved statistics
__global__ __launch_bounds__(256) void bn16_finalize_kernel(const float* __restrict__ pre, int n_parts, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ save_mean,
                                                            float* __restrict__ save_invstd, float* __restrict__ moving_mean,
                                                            float* __restrict__ moving_var, float* __restrict__ scale,
                                                            float* __restrict__ shift, int c, double inv_count, double unbias, float eps,
                                                            float momentum) {
    __shared__ double s_tot[16];
    __shared__ double s_sm[4][16];
    const int blk = blockIdx.x;
    const bool coef = threadIdx.x < 8 && blk * 8 + (int)threadIdx.x < c;
    const float gamma_v = coef ? gamma[blk * 8 + threadIdx.x] : 0.f, beta_v = coef ? beta[blk * 8 + threadIdx.x] : 0.f;
    bn16_fold_parts(pre, blk, n_parts, s_tot, s_sm);
    if (threadIdx.x < 8) {
        const Bn16Coeffs o = bn16_fwd_coeffs(threadIdx.x, blk, c, s_tot, gamma_v, beta_v, inv_count, unbias, eps);
        scale[blk * 8 + threadIdx.x] = o.sc;
        shift[blk * 8 + threadIdx.x] = o.sh;
        bn16_fwd_publish(threadIdx.x, blk, c, o, save_mean, save_invstd, moving_mean, moving_var, momentum);
    }
}

// forward apply with the statistics folded from the conv's partials: y = act(z * scale + shift (+ res))
// (a __device__ body: the one-layer kernel passes blockIdx, the grouped kernel below the block's place inside its job)
__device__ __forceinline__ void bn16_apply_pre_body(const u32x4_t* __restrict__ z, const float* __restrict__ pre, int n_parts,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                    float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                    const u32x4_t* __restrict__ res, u32x4_t* __restrict__ y, int n, int c, int c8, int hw,
                                                    double inv_count, double unbias, float eps, float momentum, int relu, int blk,
                                                    unsigned chunk, unsigned chunks) {
    __shared__ double s_tot[16];
    __shared__ double s_sm[4][16];
    __shared__ float s_scale[8], s_shift[8];
    const unsigned per_blk = (unsigned)n * (unsigned)hw;
    const unsigned len = (per_blk + chunks - 1) / chunks;
    const unsigned e0 = chunk * len, e1 = e0 + len < per_blk ? e0 + len : per_blk;
    const unsigned magic_hw = hw > 1 ? (unsigned)(0x100000000ULL / (unsigned)hw) + 1u : 0u;
    const bool exact = (unsigned long long)per_blk * (unsigned)hw < 0x100000000ULL;
    const size_t blk_off = (size_t)blk * hw, img_extra = (size_t)(c8 - 1) * hw;
    auto index_of = [&](unsigned e) {
        const unsigned img = hw <= 1 ? e : (exact ? __umulhi(e, magic_hw) : e / (unsigned)hw);
        return (size_t)e + (size_t)img * img_extra + blk_off;
    };
    // Batches of kBatch elements per thread: all loads of a batch are issued back to back, and the NEXT batch is requested before
    // this one is transformed (a loop of load / wait / transform / store per element keeps one load per tensor in flight and ran
    // the large maps at ~2 TB/s).  The first batch goes out BEFORE the fold: the statistics prologue (partial slots -> totals ->
    // scale / shift: three dependent round trips) then runs under its latency - on the small maps that batch is all a thread has.
    constexpr int kBatch = 4;
    const bool has_res = res != nullptr;
    const u32x4_t zero4 = (u32x4_t){0u, 0u, 0u, 0u};
    u32x4_t cz[kBatch], cr[kBatch];
    size_t ci[kBatch];
    auto request = [&](unsigned base, u32x4_t (&qz)[kBatch], u32x4_t (&qr)[kBatch], size_t (&qi)[kBatch]) {
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const unsigned e = base + threadIdx.x + 256u * k;
            const bool ok = e < e1;
            qi[k] = index_of(ok ? e : e0);
            qz[k] = ok ? z[qi[k]] : zero4;
            qr[k] = (ok && has_res) ? res[qi[k]] : zero4;
        }
    };
    const bool coef = threadIdx.x < 8 && blk * 8 + (int)threadIdx.x < c;
    const float gamma_v = coef ? gamma[blk * 8 + threadIdx.x] : 0.f, beta_v = coef ? beta[blk * 8 + threadIdx.x] : 0.f;
    request(e0, cz, cr, ci);
    bn16_fold_parts(pre, blk, n_parts, s_tot, s_sm);
    Bn16Coeffs co = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x < 8) {
        co = bn16_fwd_coeffs(threadIdx.x, blk, c, s_tot, gamma_v, beta_v, inv_count, unbias, eps);
        s_scale[threadIdx.x] = co.sc;
        s_shift[threadIdx.x] = co.sh;
    }
    __syncthreads();
    if (threadIdx.x < 8 && chunk == 0) bn16_fwd_publish(threadIdx.x, blk, c, co, save_mean, save_invstd, moving_mean, moving_var, momentum);
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_scale[j]; sh[j] = s_shift[j]; }  // 0 / 0 on the padding channels: they stay zero
    const float floor_v = relu ? 0.f : -__builtin_inff();
    const u32x4_t keep = bn16_channel_mask(blk, c);  // padding channels leave as zeros whatever the operands hold there
    auto apply = [&](const u32x4_t zq, const u32x4_t rq, size_t i) {
        const h16x8 zv = __builtin_bit_cast(h16x8, zq), rv = __builtin_bit_cast(h16x8, rq);
        h16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = bn16_affine((float)zv[j], sc[j], sh[j]);
            v += (float)rv[j];  // zeros without a residual: exact
            o[j] = (_Float16)fmaxf(v, floor_v);
        }
        y[i] = __builtin_bit_cast(u32x4_t, o) & keep;
    };
    for (unsigned base = e0; base < e1; base += 256u * kBatch) {
        u32x4_t nz[kBatch], nr[kBatch];
        size_t ni[kBatch];
        const bool more = base + 256u * kBatch < e1;  // uniform
        if (more) request(base + 256u * kBatch, nz, nr, ni);
#pragma unroll
        for (int k = 0; k < kBatch; ++k)
            if (base + threadIdx.x + 256u * k < e1) apply(cz[k], cr[k], ci[k]);
        if (more) {
#pragma unroll
            for (int k = 0; k < kBatch; ++k) { cz[k] = nz[k]; cr[k] = nr[k]; ci[k] = ni[k]; }
        }
    }
}

__global__ __launch_bounds__(256) void bn16_apply_pre_kernel(const u32x4_t* __restrict__ z, const float* __restrict__ pre, int n_parts,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                             float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                             const u32x4_t* __restrict__ res, u32x4_t* __restrict__ y, int n, int c,
                                                             int c8, int hw, double inv_count, double unbias, float eps, float momentum,
                                                             int relu) {
    bn16_apply_pre_body(z, pre, n_parts, gamma, beta, save_mean, save_invstd, moving_mean, moving_var, res, y, n, c, c8, hw, inv_count,
                        unbias, eps, momentum, relu, (int)blockIdx.x, blockIdx.y, gridDim.y);
}

// ---- grouped apply passes: up to four BatchNorms of ONE step of parallel chains (the k-th BatchNorm of every branch of an HRModule,
// hrnet.py:202-241) as one launch.  On the 32x24 ... 8x6 maps an apply pass is ~9 - 14 us of launch, fold and tail around 1 - 3 us of
// streaming; the branches' passes are independent, so a flat grid of [job 0's blocks | job 1's | ...] runs them under the largest
// one.  The job table travels by value; a block picks its job with compares on constant indices (no dynamic index into the
// argument struct - that would put it in scratch).
constexpr int kBnJobs = 4;
struct Bn16FwdJobs {
    const u32x4_t* z[kBnJobs]; const float* pre[kBnJobs]; const float* gamma[kBnJobs]; const float* beta[kBnJobs];
    float* save_mean[kBnJobs]; float* save_invstd[kBnJobs]; float* moving_mean[kBnJobs]; float* moving_var[kBnJobs];
    const u32x4_t* res[kBnJobs]; u32x4_t* y[kBnJobs];
    double inv_count[kBnJobs], unbias[kBnJobs];
    int n_parts[kBnJobs], n[kBnJobs], c[kBnJobs], c8[kBnJobs], hw[kBnJobs], relu[kBnJobs];
    unsigned chunks[kBnJobs], first[kBnJobs];  // first block of the job in the flat grid (0xFFFFFFFF: no such job)
    float eps, momentum;
};
#define MP_BN_PICK(t, a, j) ((j) == 0 ? (t).a[0] : (j) == 1 ? (t).a[1] : (j) == 2 ? (t).a[2] : (t).a[3])

__global__ __launch_bounds__(256) void bn16_apply_pre_grouped_kernel(const Bn16FwdJobs t) {
    const unsigned b = blockIdx.x;
    const int j = (b >= t.first[1] ? 1 : 0) + (b >= t.first[2] ? 1 : 0) + (b >= t.first[3] ? 1 : 0);
    const unsigned local = b - MP_BN_PICK(t, first, j);
    const int c8 = MP_BN_PICK(t, c8, j);
    bn16_apply_pre_body(MP_BN_PICK(t, z, j), MP_BN_PICK(t, pre, j), MP_BN_PICK(t, n_parts, j), MP_BN_PICK(t, gamma, j),
                        MP_BN_PICK(t, beta, j), MP_BN_PICK(t, save_mean, j), MP_BN_PICK(t, save_invstd, j), MP_BN_PICK(t, moving_mean, j),
                        MP_BN_PICK(t, moving_var, j), MP_BN_PICK(t, res, j), MP_BN_PICK(t, y, j), MP_BN_PICK(t, n, j), MP_BN_PICK(t, c, j), c8,
                        MP_BN_PICK(t, hw, j), MP_BN_PICK(t, inv_count, j), MP_BN_PICK(t, unbias, j), t.eps, t.momentum,
                        MP_BN_PICK(t, relu, j), (int)(local % (unsigned)c8), local / (unsigned)c8, MP_BN_PICK(t, chunks, j));
}

// backward apply on a PRE-MASKED gradient g with sum g, sum g * z folded from the data-gradient conv's partials:
// dz = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)); the residual branch's gradient is g itself (nothing to write)
__device__ __forceinline__ void bn16_bwd_apply_pre_body(const u32x4_t* __restrict__ g_in, const u32x4_t* __restrict__ z,
                                                        const float* __restrict__ pre, int n_parts, const float* __restrict__ gamma,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        float* __restrict__ dgamma_acc, float* __restrict__ dbeta_acc,
                                                        u32x4_t* __restrict__ dz, int n, int c, int c8, int hw, float inv_count, int blk,
                                                        unsigned chunk, unsigned chunks) {
    __shared__ double s_tot[16];
    __shared__ double s_sm[4][16];
    __shared__ float s_k[8], s_mu[8], s_is[8], s_mb[8], s_mg[8];
    const unsigned per_blk = (unsigned)n * (unsigned)hw;
    const unsigned len = (per_blk + chunks - 1) / chunks;
    const unsigned e0 = chunk * len, e1 = e0 + len < per_blk ? e0 + len : per_blk;
    const unsigned magic_hw = hw > 1 ? (unsigned)(0x100000000ULL / (unsigned)hw) + 1u : 0u;
    const bool exact = (unsigned long long)per_blk * (unsigned)hw < 0x100000000ULL;
    const size_t blk_off = (size_t)blk * hw, img_extra = (size_t)(c8 - 1) * hw;
    auto index_of = [&](unsigned e) {
        const unsigned img = hw <= 1 ? e : (exact ? __umulhi(e, magic_hw) : e / (unsigned)hw);
        return (size_t)e + (size_t)img * img_extra + blk_off;
    };
    constexpr int kBatch = 4;  // batched requests, the first batch before the fold (see bn16_apply_pre_kernel)
    const u32x4_t zero4 = (u32x4_t){0u, 0u, 0u, 0u};
    u32x4_t cg[kBatch], cz[kBatch];
    size_t ci[kBatch];
    auto request = [&](unsigned base, u32x4_t (&qg)[kBatch], u32x4_t (&qz)[kBatch], size_t (&qi)[kBatch]) {
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const unsigned e = base + threadIdx.x + 256u * k;
            const bool ok = e < e1;
            qi[k] = index_of(ok ? e : e0);
            qg[k] = ok ? g_in[qi[k]] : zero4;
            qz[k] = ok ? z[qi[k]] : zero4;
        }
    };
    // the per-channel parameters are requested first: their latency runs under the fold's (they used to be a second, dependent
    // round trip behind it)
    const bool coef = threadIdx.x < 8 && blk * 8 + (int)threadIdx.x < c;
    const float mean_v = coef ? mean[blk * 8 + threadIdx.x] : 0.f, invstd_v = coef ? invstd[blk * 8 + threadIdx.x] : 0.f;
    const float gamma_v = coef ? gamma[blk * 8 + threadIdx.x] : 0.f;
    request(e0, cg, cz, ci);
    bn16_fold_parts(pre, blk, n_parts, s_tot, s_sm);
    float db = 0.f, dg = 0.f;
    if (threadIdx.x < 8) {
        const int j = threadIdx.x;
        float k = 0.f, mu = 0.f, is = 0.f, mb = 0.f, mg = 0.f;
        if (coef) {
            mu = mean_v;
            is = invstd_v;
            // the conv epilogue summed g and g * z (raw z): sum g * xhat = invstd * (sum g z - mean * sum g), in fp64
            db = (float)s_tot[2 * j];
            dg = (float)((s_tot[2 * j + 1] - (double)mu * s_tot[2 * j]) * (double)is);
            k = gamma_v * is;
            mb = db * inv_count;
            mg = dg * inv_count;
        }
        s_k[j] = k; s_mu[j] = mu; s_is[j] = is; s_mb[j] = mb; s_mg[j] = mg;
    }
    __syncthreads();
    if (coef && chunk == 0) {  // parameter gradients: published behind the barrier (the accumulating form reads the arena first)
        const int ch = blk * 8 + threadIdx.x;
        dbeta[ch] = db;
        dgamma[ch] = dg;
        if (dgamma_acc && dbeta_acc) {
            dbeta_acc[ch] += db;
            dgamma_acc[ch] += dg;
        }
    }
    float k[8], mu[8], is[8], mb[8], mg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { k[j] = s_k[j]; mu[j] = s_mu[j]; is[j] = s_is[j]; mb[j] = s_mb[j]; mg[j] = s_mg[j]; }  // zeros on padding channels
    const u32x4_t keep = bn16_channel_mask(blk, c);  // padding channels leave as zeros whatever the operands hold there
    auto apply = [&](const u32x4_t gq, const u32x4_t zq, size_t i) {
        const h16x8 gv = __builtin_bit_cast(h16x8, gq), zv = __builtin_bit_cast(h16x8, zq);
        h16x8 oz;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = ((float)zv[j] - mu[j]) * is[j];
            oz[j] = (_Float16)(k[j] * ((float)gv[j] - mb[j] - xh * mg[j]));
        }
        dz[i] = __builtin_bit_cast(u32x4_t, oz) & keep;
    };
    for (unsigned base = e0; base < e1; base += 256u * kBatch) {
        u32x4_t ng[kBatch], nz[kBatch];
        size_t ni[kBatch];
        const bool more = base + 256u * kBatch < e1;  // uniform
        if (more) request(base + 256u * kBatch, ng, nz, ni);
#pragma unroll
        for (int q = 0; q < kBatch; ++q)
            if (base + threadIdx.x + 256u * q < e1) apply(cg[q], cz[q], ci[q]);
        if (more) {
#pragma unroll
            for (int q = 0; q < kBatch; ++q) { cg[q] = ng[q]; cz[q] = nz[q]; ci[q] = ni[q]; }
        }
    }
}

__global__ __launch_bounds__(256) void bn16_bwd_apply_pre_kernel(const u32x4_t* __restrict__ g_in, const u32x4_t* __restrict__ z,
                                                                 const float* __restrict__ pre, int n_parts,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ dgamma_acc,
                                                                 float* __restrict__ dbeta_acc, u32x4_t* __restrict__ dz, int n, int c,
                                                                 int c8, int hw, float inv_count) {
    bn16_bwd_apply_pre_body(g_in, z, pre, n_parts, gamma, mean, invstd, dgamma, dbeta, dgamma_acc, dbeta_acc, dz, n, c, c8, hw, inv_count,
                            (int)blockIdx.x, blockIdx.y, gridDim.y);
}

struct Bn16BwdJobs {
    const u32x4_t* g[kBnJobs]; const u32x4_t* z[kBnJobs]; const float* pre[kBnJobs]; const float* gamma[kBnJobs];
    const float* mean[kBnJobs]; const float* invstd[kBnJobs]; float* dgamma[kBnJobs]; float* dbeta[kBnJobs];
    float* dgamma_acc[kBnJobs]; float* dbeta_acc[kBnJobs]; u32x4_t* dz[kBnJobs];
    float inv_count[kBnJobs];
    int n_parts[kBnJobs], n[kBnJobs], c[kBnJobs], c8[kBnJobs], hw[kBnJobs];
    unsigned chunks[kBnJobs], first[kBnJobs];
};

__global__ __launch_bounds__(256) void bn16_bwd_apply_pre_grouped_kernel(const Bn16BwdJobs t) {
    const unsigned b = blockIdx.x;
    const int j = (b >= t.first[1] ? 1 : 0) + (b >= t.first[2] ? 1 : 0) + (b >= t.first[3] ? 1 : 0);
    const unsigned local = b - MP_BN_PICK(t, first, j);
    const int c8 = MP_BN_PICK(t, c8, j);
    bn16_bwd_apply_pre_body(MP_BN_PICK(t, g, j), MP_BN_PICK(t, z, j), MP_BN_PICK(t, pre, j), MP_BN_PICK(t, n_parts, j),
                            MP_BN_PICK(t, gamma, j), MP_BN_PICK(t, mean, j), MP_BN_PICK(t, invstd, j), MP_BN_PICK(t, dgamma, j),
                            MP_BN_PICK(t, dbeta, j), MP_BN_PICK(t, dgamma_acc, j), MP_BN_PICK(t, dbeta_acc, j), MP_BN_PICK(t, dz, j),
                            MP_BN_PICK(t, n, j), MP_BN_PICK(t, c, j), c8, MP_BN_PICK(t, hw, j), MP_BN_PICK(t, inv_count, j),
                            (int)(local % (unsigned)c8), local / (unsigned)c8, MP_BN_PICK(t, chunks, j));
}

// ---- small maps: both passes of a BatchNorm direction in ONE launch --------------------------------------------------------------
// On the 32x24 / 16x12 / 8x6 maps (<= 12 MB per tensor at N = 128) the two dependent launches above cost 13 - 20 us whatever they
// move: the second pass waits for a kernel boundary.  Here a grid of <= 128 workgroups does pass 1 over its slice, publishes its
// partial sums, meets the other workgroups at a grid barrier (a counter slot owned by the library, one per stream, release /
// acquire at agent scope) and does pass 2 over the SAME slice, which its XCD's L2 still holds.
//
// Forward progress: a grid barrier needs every workgroup resident.  128 workgroups x 256 threads x <= 96 VGPRs x 1.2 KB LDS is a
// small fraction of the chip (6 such workgroups fit on ONE CU), so even with a cooperative kernel on each of the step's five streams
// all of them are resident together; other kernels in the way finish on their own.  The wait is bounded anyway (~0.5 s): on expiry
// the workgroup poisons its outputs with NaN - the step's overflow check then skips the update - instead of hanging the GPU.
// Same fixed partition and combine order every launch: bit-reproducible.
constexpr int kCoopMaxGrid = 128;

// slot = {arrival count, generation}, 64 bytes apart; launches on one slot are serialised by their stream.  The last arriver
// resets the count and bumps the generation: correct for any grid size, no host-side reset between launches.
// Everything that crosses workgroups here (the partial sums, the two counters) is written and read with agent-scope ATOMIC
// accesses, which go to the coherence point; the barrier itself therefore needs only workgroup-scope fences.  Agent-scope
// release / acquire FENCES were measured first: they write back and invalidate the whole L2 (the conv output of the previous
// kernel is still dirty in it), which cost 10 us per barrier and evicted the slice pass 2 wants to re-read.
__device__ __forceinline__ bool grid_barrier(unsigned long long* slot, unsigned total) {
    __shared__ int s_ok;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's (atomic, write-through) partial stores have completed
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* count = slot;
        unsigned long long* generation = slot + 8;
        const unsigned long long gen = __hip_atomic_load(generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        const unsigned long long old = __hip_atomic_fetch_add(count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        if (old == total - 1) {
            __hip_atomic_store(count, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_fetch_add(generation, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            ok = 0;
            for (int it = 0; it < (1 << 22); ++it) {
                if (__hip_atomic_load(generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return s_ok != 0;
}

// totals of this block's 8 channels from the per-split partials: thread t = (channel t >> 5, split t & 31); fixed-order xor tree
__device__ __forceinline__ void coop_channel_sums(const double* part, int ch, int c, int nsplit, double& s0, double& s1) {
    s0 = 0.0;
    s1 = 0.0;
    const int sp = threadIdx.x & 31;
    if (ch < c && sp < nsplit) {  // written by other workgroups (other XCDs): read at the coherence point
        s0 = __hip_atomic_load(part + ((size_t)ch * nsplit + sp) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s1 = __hip_atomic_load(part + ((size_t)ch * nsplit + sp) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int off = 16; off >= 1; off >>= 1) {
        s0 += __shfl_xor(s0, off, 64);
        s1 += __shfl_xor(s1, off, 64);
    }
}

// grid (C8, S): block (blk, sp) owns elements [e0, e1) of the flat (image, pixel) space of channel block blk
template <bool BWD>
__global__ __launch_bounds__(256) void bn16_coop_kernel(const u32x4_t* __restrict__ dy, const u32x4_t* __restrict__ z,
                                                        const u32x4_t* __restrict__ yres, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ mean_io,
                                                        float* __restrict__ invstd_io, float* __restrict__ moving_mean,
                                                        float* __restrict__ moving_var, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, float* __restrict__ dgamma_acc,
                                                        float* __restrict__ dbeta_acc, u32x4_t* __restrict__ out,
                                                        u32x4_t* __restrict__ dres, double* __restrict__ part,
                                                        unsigned long long* __restrict__ counter, int n, int c, int c8, int hw, int relu,
                                                        float eps, float momentum) {
    const int blk = blockIdx.x, sp = blockIdx.y, nsplit = gridDim.y;
    const unsigned per_blk = (unsigned)n * (unsigned)hw;
    const unsigned len = (per_blk + nsplit - 1) / nsplit;
    const unsigned e0 = sp * len, e1 = e0 + len < per_blk ? e0 + len : per_blk;
    const unsigned magic_hw = hw > 1 ? (unsigned)(0x100000000ULL / (unsigned)hw) + 1u : 0u;
    const bool exact = (unsigned long long)per_blk * (unsigned)hw < 0x100000000ULL;
    const size_t blk_off = (size_t)blk * hw, img_extra = (size_t)(c8 - 1) * hw;
    auto index_of = [&](unsigned e) {
        const unsigned img = hw <= 1 ? e : (exact ? __umulhi(e, magic_hw) : e / (unsigned)hw);
        return (size_t)e + (size_t)img * img_extra + blk_off;
    };
    float mu[8], is[8], msc[8], msh[8];  // msc / msh: forward scale / shift for the recomputed ReLU mask (relu == 2)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = blk * 8 + j;
        mu[j] = (BWD && ch < c) ? mean_io[ch] : 0.f;
        is[j] = (BWD && ch < c) ? invstd_io[ch] : 0.f;
        msc[j] = (BWD && relu == 2 && ch < c) ? gamma[ch] * is[j] : 0.f;
        msh[j] = (BWD && relu == 2 && ch < c) ? bn16_shift(beta[ch], mu[j], msc[j]) : 0.f;
    }
    // ---- pass 1: partial sums of this slice
    float f[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = 0.f;
#pragma unroll 4
    for (unsigned e = e0 + threadIdx.x; e < e1; e += 256) {
        const size_t i = index_of(e);
        const h16x8 zv = __builtin_bit_cast(h16x8, z[i]);
        if (BWD) {
            const h16x8 gv = __builtin_bit_cast(h16x8, dy[i]);
            h16x8 yv = zv;
            if (relu == 1) yv = __builtin_bit_cast(h16x8, yres[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float g = (float)gv[j];
                if (relu == 1 && !((float)yv[j] > 0.f)) g = 0.f;
                if (relu == 2 && !bn16_relu_open((float)zv[j], msc[j], msh[j])) g = 0.f;
                f[2 * j] += g;
                f[2 * j + 1] += g * (((float)zv[j] - mu[j]) * is[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = (float)zv[j];
                f[2 * j] += v;
                f[2 * j + 1] += v * v;
            }
        }
    }
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = (double)f[j];
    __shared__ double sm[4][16];
    block_sum16_256(acc, sm);
    if (threadIdx.x < 16) {
        const int ch = blk * 8 + (threadIdx.x >> 1);
        if (ch < c) __hip_atomic_store(part + ((size_t)ch * nsplit + sp) * 2 + (threadIdx.x & 1), acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // only the workgroups of ONE channel block depend on each other: one barrier slot per channel block, gridDim.y arrivals each
    // (an arrival is a cross-XCD atomic, ~70 ns serialised per address: 128 arrivals on one slot cost ~9 us)
    const bool ok = grid_barrier(counter + (size_t)blk * 16, gridDim.y);
    // ---- totals of the 8 channels, per-channel constants
    __shared__ float s_a[8], s_b[8], s_c[8];
    {
        const int j = threadIdx.x >> 5, ch = blk * 8 + j;
        double s0, s1;
        coop_channel_sums(part, ch, c, nsplit, s0, s1);
        if ((threadIdx.x & 31) == 0) {
            float a = 0.f, b = 0.f, cc = 0.f;
            if (ch < c) {
                if (BWD) {
                    const float db = (float)s0, dg = (float)s1, inv_count = (float)(1.0 / ((double)n * hw));
                    a = gamma[ch] * is[j];   // k
                    b = db * inv_count;      // mean of g
                    cc = dg * inv_count;     // mean of g * xhat
                    if (sp == 0) {
                        dbeta[ch] = db;
                        dgamma[ch] = dg;
                        if (dgamma_acc && dbeta_acc) {
                            dbeta_acc[ch] += db;
                            dgamma_acc[ch] += dg;
                        }
                    }
                } else {
                    const double count = (double)n * hw, mean = s0 / count;
                    double var = s1 / count - mean * mean;
                    if (var < 0.0) var = 0.0;
                    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
                    a = gamma[ch] * invstd;
                    b = bn16_shift(beta[ch], (float)mean, a);
                    if (sp == 0) {
                        mean_io[ch] = (float)mean;
                        invstd_io[ch] = invstd;
                        if (moving_mean) {
                            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                            moving_mean[ch] = momentum * moving_mean[ch] + (1.f - momentum) * (float)mean;
                            moving_var[ch] = momentum * moving_var[ch] + (1.f - momentum) * (float)unbiased;
                        }
                    }
                }
                if (!ok) a = b = cc = __builtin_nanf("");  // barrier timed out: poison, never hang
            }
            s_a[j] = a; s_b[j] = b; s_c[j] = cc;
        }
    }
    __syncthreads();
    float ka[8], kb[8], kc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ka[j] = s_a[j]; kb[j] = s_b[j]; kc[j] = s_c[j]; }
    // ---- pass 2 over the same slice (L2-resident)
#pragma unroll 4
    for (unsigned e = e0 + threadIdx.x; e < e1; e += 256) {
        const size_t i = index_of(e);
        const h16x8 zv = __builtin_bit_cast(h16x8, z[i]);
        h16x8 o, og;
        if (BWD) {
            const h16x8 gv = __builtin_bit_cast(h16x8, dy[i]);
            h16x8 yv = zv;
            if (relu == 1) yv = __builtin_bit_cast(h16x8, yres[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float g = 0.f, d = 0.f;
                if (blk * 8 + j < c) {
                    g = (float)gv[j];
                    if (relu == 1 && !((float)yv[j] > 0.f)) g = 0.f;
                if (relu == 2 && !bn16_relu_open((float)zv[j], msc[j], msh[j])) g = 0.f;
                    const float xh = ((float)zv[j] - mu[j]) * is[j];
                    d = ka[j] * (g - kb[j] - xh * kc[j]);
                }
                o[j] = (_Float16)d;
                og[j] = (_Float16)g;
            }
            out[i] = __builtin_bit_cast(u32x4_t, o);
            if (dres) dres[i] = __builtin_bit_cast(u32x4_t, og);
        } else {
            h16x8 rv = zv;
            if (yres) rv = __builtin_bit_cast(h16x8, yres[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = 0.f;
                if (blk * 8 + j < c) {
                    v = bn16_affine((float)zv[j], ka[j], kb[j]);
                    if (yres) v += (float)rv[j];
                    if (relu) v = fmaxf(v, 0.f);
                }
                o[j] = (_Float16)v;
            }
            out[i] = __builtin_bit_cast(u32x4_t, o);
        }
    }
}

// Barrier slots: a zeroed device pool created on the first use outside a stream capture, 128 slots of 128 bytes per stream that ever
// launched a one-launch BatchNorm (the captured graph keeps the slot of its capture stream; kernels of one stream are ordered).
constexpr int kCoopSlots = 64;
static unsigned long long* coop_slot_for(hipStream_t s) {
    static std::mutex mu;
    static unsigned long long* pool = nullptr;
    static bool failed = false;
    static std::unordered_map<hipStream_t, int> slots;
    std::lock_guard<std::mutex> lock(mu);
    if (!pool) {
        if (failed) return nullptr;
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return nullptr;  // not now: an allocation inside a capture is not allowed; the two-launch form runs instead
        }
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)kCoopSlots * kCoopMaxGrid * 128) != hipSuccess ||
            hipMemset(p, 0, (size_t)kCoopSlots * kCoopMaxGrid * 128) != hipSuccess) {
            (void)hipGetLastError();
            failed = true;
            return nullptr;
        }
        pool = reinterpret_cast<unsigned long long*>(p);
    }
    auto it = slots.find(s);
    if (it == slots.end()) {
        if ((int)slots.size() >= kCoopSlots) return nullptr;
        it = slots.emplace(s, (int)slots.size()).first;
    }
    return pool + (size_t)it->second * kCoopMaxGrid * 16;  // kCoopMaxGrid slots of 128 bytes: one per channel block
}

// the one-launch form applies to: tensors of <= MP_BN16_COOP_MAX (default 512 K) 16-byte elements with C8 <= 128 - measured on the
// HRNet-W32 amp-O2 step at N = 128: threshold 0 / 256 K / 512 K / 1 M -> 36.06 / 35.31 / 35.04 / 36.64 ms
static unsigned long long* bn16_coop_plan(int n, int c8, int hw, hipStream_t s, int& nsplit) {
    static const long long max_elems = [] {
        const char* e = knob("MP_BN16_COOP_MAX");  // 0 switches the one-launch form off
        return e ? atoll(e) : (1LL << 19);
    }();
    if ((long long)n * c8 * hw > max_elems || c8 > kCoopMaxGrid) return nullptr;
    nsplit = kCoopMaxGrid / c8;
    if (nsplit > 32) nsplit = 32;
    while (nsplit > 1 && ((long long)n * hw + nsplit - 1) / nsplit < 256) --nsplit;
    return coop_slot_for(s);
}

// backward of the exchange-unit sum in the c8 layout: g = dy * (out > 0); term k gets the s_k x s_k block sums of g
__global__ __launch_bounds__(256) void fuse_sum16_bwd_kernel(const u32x4_t* __restrict__ dy, const u32x4_t* __restrict__ out,
                                                             u32x4_t* __restrict__ dt, int planes, int h, int w, int sh, int relu) {
    const int lh = h >> sh, lw = w >> sh, s = 1 << sh;
    const size_t total = (size_t)planes * lh * lw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lx = (int)(i % lw);
        const size_t r = i / lw;
        const int ly = (int)(r % lh);
        const size_t plane = r / lh;
        const size_t o = (plane * h + (size_t)ly * s) * w + (size_t)lx * s;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < s; ++a)
            for (int b = 0; b < s; ++b) {
                const h16x8 gv = __builtin_bit_cast(h16x8, dy[o + (size_t)a * w + b]);
                h16x8 ov = gv;
                if (relu) ov = __builtin_bit_cast(h16x8, out[o + (size_t)a * w + b]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float g = (float)gv[j];
                    if (relu && !((float)ov[j] > 0.f)) g = 0.f;
                    acc[j] += g;
                }
            }
        h16x8 res;
#pragma unroll
        for (int j = 0; j < 8; ++j) res[j] = (_Float16)acc[j];
        dt[i] = __builtin_bit_cast(u32x4_t, res);
    }
}

// out = a + b (+ c) (+ d): the gradients that reach one tensor from its consumers (the exchange unit feeds every branch output to
// every row) summed in ONE pass with fp32 arithmetic - autograd would add them pairwise, k - 1 launches of 3 tensors each.
// fp16: 16-byte units of 8 halves, one rounding at the end.
template <bool HALF>
__global__ __launch_bounds__(256) void sum_tensors_kernel(const u32x4_t* __restrict__ a, const u32x4_t* __restrict__ b,
                                                          const u32x4_t* __restrict__ c, const u32x4_t* __restrict__ d,
                                                          u32x4_t* __restrict__ out, size_t units) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < units; i += (size_t)gridDim.x * 256) {
        const u32x4_t va = a[i], vb = b[i];
        u32x4_t vc = va, vd = va;
        if (c) vc = c[i];
        if (d) vd = d[i];
        if (HALF) {
            const h16x8 ha = __builtin_bit_cast(h16x8, va), hb = __builtin_bit_cast(h16x8, vb), hc = __builtin_bit_cast(h16x8, vc),
                        hd = __builtin_bit_cast(h16x8, vd);
            h16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = (float)ha[j] + (float)hb[j];
                if (c) v += (float)hc[j];
                if (d) v += (float)hd[j];
                o[j] = (_Float16)v;
            }
            out[i] = __builtin_bit_cast(u32x4_t, o);
        } else {
            const float4 fa = __builtin_bit_cast(float4, va), fb = __builtin_bit_cast(float4, vb), fc = __builtin_bit_cast(float4, vc),
                         fd = __builtin_bit_cast(float4, vd);
            float4 o = make_float4(fa.x + fb.x, fa.y + fb.y, fa.z + fb.z, fa.w + fb.w);
            if (c) { o.x += fc.x; o.y += fc.y; o.z += fc.z; o.w += fc.w; }
            if (d) { o.x += fd.x; o.y += fd.y; o.z += fd.z; o.w += fd.w; }
            out[i] = __builtin_bit_cast(u32x4_t, o);
        }
    }
}

// ---- the element-wise producers of a BatchNorm's output gradient, with that BatchNorm's backward sums (round 3) --------------------
// Where the gradient that reaches a BatchNorm does not come out of a data-gradient conv - the sum over the consumers of a branch
// output (FanOutFn), a term of the exchange unit's backward (FuseSum16Fn) - the kernel that writes it takes the conv epilogue's
// role: grid (C8, chunks) like the BatchNorm passes (a block stays inside ONE channel block, so per-channel sums live in registers),
// g = value * [y > 0] stored, partial sums of g and g * z per (channel block, chunk) in the conv epilogue's layout
// [C8][chunks][8][2] fp32.  Fixed partition, fixed order: bit-reproducible.
__device__ __forceinline__ void ew_stats_tail(float (&f)[16], float* __restrict__ part, int blk, int chunk, int n_parts) {
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = (double)f[j];
    __shared__ double sm[4][16];
    block_sum16_256(acc, sm);
    if (threadIdx.x < 16) part[((size_t)blk * n_parts + chunk) * 16 + threadIdx.x] = (float)acc[0];
}

// out = fp16(((a + b) + c) + d) * [y > 0]  (mp_sum_tensors' arithmetic, then the mask)
__global__ __launch_bounds__(256) void sum_tensors16_stats_kernel(const u32x4_t* __restrict__ a, const u32x4_t* __restrict__ b,
                                                                  const u32x4_t* __restrict__ c, const u32x4_t* __restrict__ d,
                                                                  const u32x4_t* __restrict__ z, const u32x4_t* __restrict__ y,
                                                                  u32x4_t* __restrict__ out, float* __restrict__ part, int n, int c8,
                                                                  int hw, int relu) {
    const int blk = blockIdx.x;
    const unsigned per_blk = (unsigned)n * (unsigned)hw;
    const unsigned len = (per_blk + gridDim.y - 1) / gridDim.y;
    const unsigned e0 = blockIdx.y * len, e1 = e0 + len < per_blk ? e0 + len : per_blk;
    const unsigned magic_hw = hw > 1 ? (unsigned)(0x100000000ULL / (unsigned)hw) + 1u : 0u;
    const bool exact = (unsigned long long)per_blk * (unsigned)hw < 0x100000000ULL;
    const size_t blk_off = (size_t)blk * hw, img_extra = (size_t)(c8 - 1) * hw;
    float f[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = 0.f;
#pragma unroll 2
    for (unsigned e = e0 + threadIdx.x; e < e1; e += 256) {
        const unsigned img = hw <= 1 ? e : (exact ? __umulhi(e, magic_hw) : e / (unsigned)hw);
        const size_t i = (size_t)e + (size_t)img * img_extra + blk_off;
        const h16x8 ha = __builtin_bit_cast(h16x8, a[i]), hb = __builtin_bit_cast(h16x8, b[i]);
        h16x8 hc = ha, hd = ha;
        if (c) hc = __builtin_bit_cast(h16x8, c[i]);
        if (d) hd = __builtin_bit_cast(h16x8, d[i]);
        const h16x8 zv = __builtin_bit_cast(h16x8, z[i]);
        h16x8 yv = zv;
        if (relu) yv = __builtin_bit_cast(h16x8, y[i]);
        h16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = (float)ha[j] + (float)hb[j];
            if (c) v += (float)hc[j];
            if (d) v += (float)hd[j];
            _Float16 h = (_Float16)v;
            if (relu && !((float)yv[j] > 0.f)) h = (_Float16)0.f;
            o[j] = h;
            f[2 * j] += (float)h;
            f[2 * j + 1] = __builtin_fmaf((float)h, (float)zv[j], f[2 * j + 1]);
        }
        out[i] = __builtin_bit_cast(u32x4_t, o);
    }
    ew_stats_tail(f, part, blk, blockIdx.y, gridDim.y);
}

// one term of the exchange unit's backward: dt = fp16(sum over the s x s block of dy * [out > 0]) * [y_t > 0]
// (SH = log2 of the scale as a template parameter: the s x s block is walked by fully unrolled loops whose 2 s loads per row
// are all issued before the first use - with a run-time s the thread had one load in flight at a time)
template <int SH>
__global__ __launch_bounds__(256) void fuse_sum16_bwd_stats_kernel(const u32x4_t* __restrict__ dy, const u32x4_t* __restrict__ outp,
                                                                   const u32x4_t* __restrict__ z, const u32x4_t* __restrict__ y,
                                                                   u32x4_t* __restrict__ dt, float* __restrict__ part, int n, int c8, int h,
                                                                   int w, int relu, int relu_t) {
    constexpr int sh = SH;
    const int blk = blockIdx.x;
    const int lh = h >> sh, lw = w >> sh, lhw = lh * lw;
    constexpr int s = 1 << SH;
    const unsigned per_blk = (unsigned)n * (unsigned)lhw;
    const unsigned len = (per_blk + gridDim.y - 1) / gridDim.y;
    const unsigned e0 = blockIdx.y * len, e1 = e0 + len < per_blk ? e0 + len : per_blk;
    float f[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = 0.f;
    for (unsigned e = e0 + threadIdx.x; e < e1; e += 256) {
        const unsigned img = e / (unsigned)lhw, pix = e - img * (unsigned)lhw;
        const unsigned ly = pix / (unsigned)lw, lx = pix - ly * (unsigned)lw;
        const size_t plane = (size_t)img * c8 + blk;
        const size_t i = plane * lhw + pix;
        const size_t o = (plane * h + (size_t)ly * s) * w + (size_t)lx * s;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // same summation order as before (row by row, left to right): same bits
#pragma unroll
        for (int a = 0; a < s; ++a) {
            u32x4_t gq[s], oq[s];
#pragma unroll
            for (int b = 0; b < s; ++b) {
                gq[b] = dy[o + (size_t)a * w + b];
                oq[b] = relu ? outp[o + (size_t)a * w + b] : gq[b];
            }
#pragma unroll
            for (int b = 0; b < s; ++b) {
                const h16x8 gv = __builtin_bit_cast(h16x8, gq[b]), ov = __builtin_bit_cast(h16x8, oq[b]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float g = (float)gv[j];
                    if (relu && !((float)ov[j] > 0.f)) g = 0.f;
                    acc[j] += g;
                }
            }
        }
        const h16x8 zv = __builtin_bit_cast(h16x8, z[i]);
        h16x8 yv = zv;
        if (relu_t) yv = __builtin_bit_cast(h16x8, y[i]);
        h16x8 res;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 hv = (_Float16)acc[j];
            if (relu_t && !((float)yv[j] > 0.f)) hv = (_Float16)0.f;
            res[j] = hv;
            f[2 * j] += (float)hv;
            f[2 * j + 1] = __builtin_fmaf((float)hv, (float)zv[j], f[2 * j + 1]);
        }
        dt[i] = __builtin_bit_cast(u32x4_t, res);
    }
    ew_stats_tail(f, part, blk, blockIdx.y, gridDim.y);
}

// image groups x pixel chunks of the fp16 reductions: about 512+ blocks, fixed by the shape (deterministic)
static void bn16_split(int n, int c8, int hw, int& gi, int& gp) {
    int want = 512 / c8;  // a block should stream >= ~50 KB to amortise its reduction tail
    if (want < 32) want = 32;
    if (want > kBn16MaxSplit) want = kBn16MaxSplit;
    gi = n < want ? n : want;
    gp = want / gi;
    if (gp < 1) gp = 1;
    while (gp > 1 && (hw + gp - 1) / gp < 256) --gp;  // at least one element per thread
}

// pixel-range chunks per channel block of the apply kernels: ~768 blocks in all, each streaming at least 1024 elements
static unsigned bn16_apply_chunks(int n, int c8, int hw) {
    size_t chunks = (768 + c8 - 1) / c8;
    const size_t per_blk = (size_t)n * hw;
    while (chunks > 1 && per_blk / chunks < 1024) --chunks;
    return (unsigned)chunks;
}

static int log2_exact_t(int v) {
    for (int i = 0; i < 16; ++i)
        if ((1 << i) == v) return i;
    return -1;
}

}  // namespace mp

using namespace mp;

extern "C" {

size_t mp_bn_workspace_bytes(int c) {
    if (c <= 0) return 0;
    // fp64 partials [C][splits][2] (32 splits in the fp32 passes, up to kBn16MaxSplit in the fp16 ones) + scale/shift
    return (size_t)c * kBn16MaxSplit * 2 * sizeof(double) + (size_t)c * 2 * sizeof(float) + 256;
}

int mp_bn_train_fwd(const float* z, const float* gamma, const float* beta, const float* res, float* y, float* save_mean,
                    float* save_invstd, float* moving_mean, float* moving_var, int n, int c, int hw, float eps,
                    float momentum, int relu, void* workspace, size_t workspace_bytes, mp_stream_t stream) {
    if (!z || !gamma || !beta || !y || !save_mean || !save_invstd) return MP_ERR_NULL;
    if ((moving_mean == nullptr) != (moving_var == nullptr)) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || hw <= 0) return MP_ERR_SHAPE;
    if (!workspace || workspace_bytes < mp_bn_workspace_bytes(c)) return MP_ERR_WORKSPACE;
    double* part = reinterpret_cast<double*>(workspace);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(bn_reduce_kernel<false>, dim3(c, kBnSplit), dim3(256), 0, s, nullptr, z, nullptr, nullptr, nullptr, part,
                       n, c, hw, 0);
    int rc = check_launch();
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(n * c), dim3(256), 0, s, z, part, gamma, beta, save_mean, save_invstd, moving_mean,
                       moving_var, res, y, c, hw, relu ? 1 : 0, (double)n * hw, eps, momentum);
    return check_launch();
}

int mp_bn_train_bwd(const float* dy, const float* z, const float* y, const float* gamma, const float* save_mean,
                    const float* save_invstd, float* dz, float* dres, float* dgamma, float* dbeta, int n, int c, int hw,
                    int relu, void* workspace, size_t workspace_bytes, mp_stream_t stream) {
    return mp_bn_train_bwd_acc(dy, z, y, gamma, save_mean, save_invstd, dz, dres, dgamma, dbeta, nullptr, nullptr, n, c, hw, relu,
                               workspace, workspace_bytes, stream);
}

int mp_bn_train_bwd_acc(const float* dy, const float* z, const float* y, const float* gamma, const float* save_mean,
                        const float* save_invstd, float* dz, float* dres, float* dgamma, float* dbeta, float* dgamma_acc,
                        float* dbeta_acc, int n, int c, int hw, int relu, void* workspace, size_t workspace_bytes,
                        mp_stream_t stream) {
    if (!dy || !z || !gamma || !save_mean || !save_invstd || !dz || !dgamma || !dbeta) return MP_ERR_NULL;
    if (relu && !y) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || hw <= 0) return MP_ERR_SHAPE;
    if (!workspace || workspace_bytes < mp_bn_workspace_bytes(c)) return MP_ERR_WORKSPACE;
    double* part = reinterpret_cast<double*>(workspace);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(bn_reduce_kernel<true>, dim3(c, kBnSplit), dim3(256), 0, s, dy, z, y, save_mean, save_invstd, part, n, c,
                       hw, relu ? 1 : 0);
    int rc = check_launch();
    if (rc != MP_OK) return rc;
    const bool acc = dgamma_acc && dbeta_acc;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(n * c), dim3(256), 0, s, dy, z, y, part, gamma, save_mean, save_invstd, dgamma, dbeta,
                       acc ? dgamma_acc : nullptr, acc ? dbeta_acc : nullptr, dz, dres, c, hw, relu ? 1 : 0,
                       (float)(1.0 / ((double)n * hw)));
    return check_launch();
}

int mp_fuse_upsample_sum_bwd(const float* dy, const float* out, float* dbase, float* dt1, int s1, float* dt2, int s2,
                             float* dt3, int s3, int n, int c, int h, int w, int relu, mp_stream_t stream) {
    if (!dy || (relu && !out)) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    hipStream_t s = as_stream(stream);
    float* dts[4] = {dbase, dt1, dt2, dt3};
    const int ss[4] = {1, s1, s2, s3};
    for (int k = 0; k < 4; ++k) {
        if (!dts[k]) continue;
        const int sh = log2_exact_t(ss[k]);
        if (sh < 0 || (h % ss[k]) || (w % ss[k])) return MP_ERR_UNSUPPORTED;
        const size_t total = (size_t)n * c * (h >> sh) * (w >> sh);
        size_t blocks = (total + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL(fuse_sum_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dy, out, nullptr, dts[k], n * c, h, w, sh,
                           relu ? 1 : 0);
        int rc = check_launch();
        if (rc != MP_OK) return rc;
    }
    return MP_OK;
}

int mp_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t count, float lr, float beta1,
                  float beta2, float eps, float weight_decay, mp_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) return MP_ERR_NULL;
    if (count == 0) return MP_OK;
    size_t blocks = (count + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq,
                       count, lr, beta1, beta2, eps, weight_decay);
    return check_launch();
}

int mp_optimizer_step(int kind, float* param, const float* grad, float* state1, float* state2, size_t count, float lr,
                      float grad_scale, float weight_decay, const float hyper[5], mp_stream_t stream) {
    if (!param || !grad || !hyper) return MP_ERR_NULL;
    if (kind < 1 || kind > 4) return MP_ERR_UNSUPPORTED;
    if (!state1 && !(kind == 2 && hyper[0] == 0.f)) return MP_ERR_NULL;
    if (kind == 1 && !state2) return MP_ERR_NULL;
    if (count == 0) return MP_OK;
    size_t blocks = (count + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = as_stream(stream);
    const float h0 = hyper[0], h1 = hyper[1], h2 = hyper[2], h3 = hyper[3], h4 = hyper[4];
    switch (kind) {
        case 1: hipLaunchKernelGGL(optimizer_kernel<1>, grid, block, 0, s, param, grad, state1, state2, count, lr, grad_scale, weight_decay, h0, h1, h2, h3, h4); break;
        case 2: hipLaunchKernelGGL(optimizer_kernel<2>, grid, block, 0, s, param, grad, state1, state2, count, lr, grad_scale, weight_decay, h0, h1, h2, h3, h4); break;
        case 3: hipLaunchKernelGGL(optimizer_kernel<3>, grid, block, 0, s, param, grad, state1, state2, count, lr, grad_scale, weight_decay, h0, h1, h2, h3, h4); break;
        default: hipLaunchKernelGGL(optimizer_kernel<4>, grid, block, 0, s, param, grad, state1, state2, count, lr, grad_scale, weight_decay, h0, h1, h2, h3, h4); break;
    }
    return check_launch();
}

int mp_f16_bn_train_fwd(const void* z, const float* gamma, const float* beta, const void* res, void* y, float* save_mean,
                        float* save_invstd, float* moving_mean, float* moving_var, int n, int c, int hw, float eps, float momentum,
                        int relu, void* workspace, size_t workspace_bytes, mp_stream_t stream) {
    if (!z || !gamma || !beta || !y || !save_mean || !save_invstd) return MP_ERR_NULL;
    if ((moving_mean == nullptr) != (moving_var == nullptr)) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || hw <= 0) return MP_ERR_SHAPE;
    if (!workspace || workspace_bytes < mp_bn_workspace_bytes(c)) return MP_ERR_WORKSPACE;
    double* part = reinterpret_cast<double*>(workspace);
    const int c8 = (c + 7) / 8;
    hipStream_t s = as_stream(stream);
    int coop_split;
    if (unsigned long long* counter = bn16_coop_plan(n, c8, hw, s, coop_split)) {  // small map: both passes in one launch
        hipLaunchKernelGGL(bn16_coop_kernel<false>, dim3(c8, coop_split), dim3(256), 0, s, nullptr, reinterpret_cast<const u32x4_t*>(z),
                           reinterpret_cast<const u32x4_t*>(res), gamma, beta, save_mean, save_invstd, moving_mean, moving_var, nullptr,
                           nullptr, nullptr, nullptr, reinterpret_cast<u32x4_t*>(y), nullptr, part, counter, n, c, c8, hw, relu ? 1 : 0,
                           eps, momentum);
        return check_launch();
    }
    int gi, gp;
    bn16_split(n, c8, hw, gi, gp);
    float* scale = reinterpret_cast<float*>(part + (size_t)c * kBn16MaxSplit * 2);
    float* shift = scale + c;
    hipLaunchKernelGGL(bn16_reduce_kernel<false>, dim3(c8, gi * gp), dim3(256), 0, s, nullptr, reinterpret_cast<const u32x4_t*>(z),
                       nullptr, nullptr, nullptr, nullptr, nullptr, part, n, c, c8, hw, 0, gi, gp);
    int rc = check_launch();
    if (rc != MP_OK) return rc;
    (void)scale; (void)shift;
    hipLaunchKernelGGL(bn16_apply_kernel, dim3(c8, bn16_apply_chunks(n, c8, hw)), dim3(256), 0, s, reinterpret_cast<const u32x4_t*>(z),
                       part, gamma, beta, save_mean, save_invstd, moving_mean, moving_var, reinterpret_cast<const u32x4_t*>(res),
                       reinterpret_cast<u32x4_t*>(y), n, c, c8, hw, gi * gp, (double)n * hw, eps, momentum, relu ? 1 : 0);
    return check_launch();
}

int mp_f16_bn_train_bwd(const void* dy, const void* z, const void* y, const float* gamma, const float* beta, const float* save_mean,
                        const float* save_invstd, void* dz, void* dres, float* dgamma, float* dbeta, float* dgamma_acc, float* dbeta_acc,
                        int n, int c, int hw, int relu, void* workspace, size_t workspace_bytes, mp_stream_t stream) {
    if (!dy || !z || !gamma || !save_mean || !save_invstd || !dz || !dgamma || !dbeta) return MP_ERR_NULL;
    // ReLU mask: from the stored output y - or, for a layer without residual input, re-derived from z with the forward arithmetic
    // (y == NULL, beta given): y is then not read at all
    if (relu && !y && (!beta || dres)) return MP_ERR_NULL;
    relu = relu ? (y ? 1 : 2) : 0;
    if (n <= 0 || c <= 0 || hw <= 0) return MP_ERR_SHAPE;
    if (!workspace || workspace_bytes < mp_bn_workspace_bytes(c)) return MP_ERR_WORKSPACE;
    double* part = reinterpret_cast<double*>(workspace);
    const int c8 = (c + 7) / 8;
    hipStream_t s = as_stream(stream);
    int coop_split;
    if (unsigned long long* counter = bn16_coop_plan(n, c8, hw, s, coop_split)) {  // small map: both passes in one launch
        const bool acc2 = dgamma_acc && dbeta_acc;
        hipLaunchKernelGGL(bn16_coop_kernel<true>, dim3(c8, coop_split), dim3(256), 0, s, reinterpret_cast<const u32x4_t*>(dy),
                           reinterpret_cast<const u32x4_t*>(z), reinterpret_cast<const u32x4_t*>(y), gamma, beta,
                           const_cast<float*>(save_mean), const_cast<float*>(save_invstd), nullptr, nullptr, dgamma, dbeta,
                           acc2 ? dgamma_acc : nullptr, acc2 ? dbeta_acc : nullptr, reinterpret_cast<u32x4_t*>(dz),
                           reinterpret_cast<u32x4_t*>(dres), part, counter, n, c, c8, hw, relu, 0.f, 0.f);
        return check_launch();
    }
    int gi, gp;
    bn16_split(n, c8, hw, gi, gp);
    hipLaunchKernelGGL(bn16_reduce_kernel<true>, dim3(c8, gi * gp), dim3(256), 0, s, reinterpret_cast<const u32x4_t*>(dy),
                       reinterpret_cast<const u32x4_t*>(z), reinterpret_cast<const u32x4_t*>(y), save_mean, save_invstd, gamma, beta,
                       part, n, c, c8, hw, relu, gi, gp);
    int rc = check_launch();
    if (rc != MP_OK) return rc;
    const bool acc = dgamma_acc && dbeta_acc;
    hipLaunchKernelGGL(bn16_bwd_apply_kernel, dim3(c8, bn16_apply_chunks(n, c8, hw)), dim3(256), 0, s,
                       reinterpret_cast<const u32x4_t*>(dy), reinterpret_cast<const u32x4_t*>(z), reinterpret_cast<const u32x4_t*>(y),
                       part, gamma, save_mean, save_invstd, dgamma, dbeta, acc ? dgamma_acc : nullptr, acc ? dbeta_acc : nullptr,
                       reinterpret_cast<u32x4_t*>(dz), reinterpret_cast<u32x4_t*>(dres), n, c, c8, hw, gi * gp, relu,
                       (float)(1.0 / ((double)n * hw)), beta);
    return check_launch();
}

// pixel-range chunks per channel block of the statistics-free apply kernels.  These passes are bound by the bytes they keep in
// flight (a thread has 4 x 16 B per operand tensor outstanding), not by the n_parts x 64 bytes of partials every block folds first:
// ~1024 blocks (four per CU) of >= 1024 elements measured best on every map size of the HRNet step - 512 fatter blocks ran the
// 64x48 maps at 2.2 TB/s, 1024 at 3.3 TB/s; 2048 and more lose again to the redundant folds (round 3 sweep with MP_BN_PRE_BLOCKS / MP_BN_PRE_MIN, tools/ab_train.sh).
static unsigned bn16_pre_chunks(int n, int c8, int hw, int n_parts) {
    (void)n_parts;
    size_t total = 1024, floor_elems = 1024;
    if (const char* e = knob("MP_BN_PRE_BLOCKS")) total = (size_t)atoi(e);
    if (const char* e = knob("MP_BN_PRE_MIN")) floor_elems = (size_t)atoi(e);
    size_t chunks = (total + c8 - 1) / c8;
    const size_t per_blk = (size_t)n * hw;
    while (chunks > 1 && per_blk / chunks < floor_elems) --chunks;
    return (unsigned)chunks;
}

// more slots than a consumer block folds: reduce them to one per channel block (into the workspace)
static int bn16_prefold(const float*& pre, int& n_parts, int c8, void* workspace, hipStream_t s) {
    int above = kMaxFoldParts;
    if (const char* e = knob("MP_BN_PREFOLD_ABOVE")) above = atoi(e);
    if (n_parts <= above) return MP_OK;
    float* folded = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(bn16_fold_kernel, dim3(c8, kFoldSplit), dim3(256), 0, s, pre, folded, n_parts);
    pre = folded;
    n_parts = kFoldSplit;
    return check_launch();
}

int mp_f16_bn_train_fwd_stats(const void* z, const float* gamma, const float* beta, const void* res, void* y, float* save_mean,
                              float* save_invstd, float* moving_mean, float* moving_var, int n, int c, int hw, float eps,
                              float momentum, int relu, const float* partials, int n_parts, void* workspace, size_t workspace_bytes,
                              mp_stream_t stream) {
    if (!z || !gamma || !beta || !y || !save_mean || !save_invstd || !partials) return MP_ERR_NULL;
    if ((moving_mean == nullptr) != (moving_var == nullptr)) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || hw <= 0 || n_parts <= 0) return MP_ERR_SHAPE;
    if (!workspace || workspace_bytes < mp_bn_workspace_bytes(c)) return MP_ERR_WORKSPACE;
    const int c8 = (c + 7) / 8;
    hipStream_t s = as_stream(stream);
    int rc = bn16_prefold(partials, n_parts, c8, workspace, s);
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(bn16_apply_pre_kernel, dim3(c8, bn16_pre_chunks(n, c8, hw, n_parts)), dim3(256), 0, s, reinterpret_cast<const u32x4_t*>(z),
                       partials, n_parts, gamma, beta, save_mean, save_invstd, moving_mean, moving_var,
                       reinterpret_cast<const u32x4_t*>(res), reinterpret_cast<u32x4_t*>(y), n, c, c8, hw, 1.0 / ((double)n * hw),
                       (double)n * hw > 1.0 ? ((double)n * hw) / ((double)n * hw - 1.0) : 1.0, eps, momentum, relu ? 1 : 0);
    return check_launch();
}

int mp_f16_bn_train_finalize(const float* gamma, const float* beta, float* save_mean, float* save_invstd, float* moving_mean,
                             float* moving_var, int n, int c, int hw, float eps, float momentum, const float* partials, int n_parts,
                             float* scale, float* shift, void* workspace, size_t workspace_bytes, mp_stream_t stream) {
    if (!gamma || !beta || !save_mean || !save_invstd || !partials || !scale || !shift) return MP_ERR_NULL;
    if ((moving_mean == nullptr) != (moving_var == nullptr)) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || hw <= 0 || n_parts <= 0) return MP_ERR_SHAPE;
    if (!workspace || workspace_bytes < mp_bn_workspace_bytes(c)) return MP_ERR_WORKSPACE;
    const int c8 = (c + 7) / 8;
    hipStream_t s = as_stream(stream);
    int rc = bn16_prefold(partials, n_parts, c8, workspace, s);
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(bn16_finalize_kernel, dim3(c8), dim3(256), 0, s, partials, n_parts, gamma, beta, save_mean, save_invstd, moving_mean,
                       moving_var, scale, shift, c, 1.0 / ((double)n * hw),
                       (double)n * hw > 1.0 ? ((double)n * hw) / ((double)n * hw - 1.0) : 1.0, eps, momentum);
    return check_launch();
}

int mp_f16_bn_train_bwd_stats(const void* g, const void* z, const float* gamma, const float* save_mean, const float* save_invstd,
                              void* dz, float* dgamma, float* dbeta, float* dgamma_acc, float* dbeta_acc, int n, int c, int hw,
                              const float* partials, int n_parts, void* workspace, size_t workspace_bytes, mp_stream_t stream) {
    if (!g || !z || !gamma || !save_mean || !save_invstd || !dz || !dgamma || !dbeta || !partials) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || hw <= 0 || n_parts <= 0) return MP_ERR_SHAPE;
    if (!workspace || workspace_bytes < mp_bn_workspace_bytes(c)) return MP_ERR_WORKSPACE;
    const int c8 = (c + 7) / 8;
    hipStream_t s = as_stream(stream);
    int rc = bn16_prefold(partials, n_parts, c8, workspace, s);
    if (rc != MP_OK) return rc;
    const bool acc = dgamma_acc && dbeta_acc;
    hipLaunchKernelGGL(bn16_bwd_apply_pre_kernel, dim3(c8, bn16_pre_chunks(n, c8, hw, n_parts)), dim3(256), 0, s,
                       reinterpret_cast<const u32x4_t*>(g), reinterpret_cast<const u32x4_t*>(z), partials, n_parts, gamma, save_mean,
                       save_invstd, dgamma, dbeta, acc ? dgamma_acc : nullptr, acc ? dbeta_acc : nullptr, reinterpret_cast<u32x4_t*>(dz),
                       n, c, c8, hw, (float)(1.0 / ((double)n * hw)));
    return check_launch();
}

// Grouped apply passes (kernels above): job j must satisfy what the one-layer entries check; a job with more than kMaxFoldParts
// partial slots is folded by its own bn16_fold_kernel launch first (its workspace).
int mp_f16_bn_train_fwd_stats_grouped(const mp_f16_bn_fwd_job* jobs, int n_jobs, float eps, float momentum, mp_stream_t stream) {
    if (!jobs) return MP_ERR_NULL;
    if (n_jobs < 1 || n_jobs > kBnJobs) return MP_ERR_SHAPE;
    hipStream_t s = as_stream(stream);
    Bn16FwdJobs t{};
    unsigned total = 0;
    size_t fold_off = 0;
    for (int j = 0; j < kBnJobs; ++j) t.first[j] = 0xFFFFFFFFu;
    for (int j = 0; j < n_jobs; ++j) {
        const mp_f16_bn_fwd_job& q = jobs[j];
        if (!q.z_dev || !q.gamma_dev || !q.beta_dev || !q.y_dev || !q.save_mean_dev || !q.save_invstd_dev || !q.partials_dev) return MP_ERR_NULL;
        if ((q.moving_mean_dev == nullptr) != (q.moving_var_dev == nullptr)) return MP_ERR_NULL;
        if (q.n <= 0 || q.c <= 0 || q.hw <= 0 || q.n_parts <= 0) return MP_ERR_SHAPE;
        if (!q.workspace_dev || q.workspace_bytes < mp_bn_workspace_bytes(q.c)) return MP_ERR_WORKSPACE;
        const int c8 = (q.c + 7) / 8;
        const float* pre = q.partials_dev;
        int n_parts = q.n_parts;
        // jobs may share one workspace (chains on one stream): job j folds into its own region behind those of the jobs before it
        if (q.workspace_bytes < fold_off + (size_t)c8 * kFoldSplit * 16 * sizeof(float)) return MP_ERR_WORKSPACE;
        const int rc = bn16_prefold(pre, n_parts, c8, reinterpret_cast<char*>(q.workspace_dev) + fold_off, s);
        if (rc != MP_OK) return rc;
        fold_off += ((size_t)c8 * kFoldSplit * 16 * sizeof(float) + 255) / 256 * 256;
        const double cnt = (double)q.n * q.hw;
        t.z[j] = reinterpret_cast<const u32x4_t*>(q.z_dev); t.pre[j] = pre; t.gamma[j] = q.gamma_dev; t.beta[j] = q.beta_dev;
        t.save_mean[j] = q.save_mean_dev; t.save_invstd[j] = q.save_invstd_dev; t.moving_mean[j] = q.moving_mean_dev;
        t.moving_var[j] = q.moving_var_dev; t.res[j] = reinterpret_cast<const u32x4_t*>(q.res_dev); t.y[j] = reinterpret_cast<u32x4_t*>(q.y_dev);
        t.inv_count[j] = 1.0 / cnt; t.unbias[j] = cnt > 1.0 ? cnt / (cnt - 1.0) : 1.0;
        t.n_parts[j] = n_parts; t.n[j] = q.n; t.c[j] = q.c; t.c8[j] = c8; t.hw[j] = q.hw; t.relu[j] = q.relu ? 1 : 0;
        t.chunks[j] = bn16_pre_chunks(q.n, c8, q.hw, n_parts);
        t.first[j] = total;
        total += (unsigned)c8 * t.chunks[j];
    }
    for (int j = n_jobs; j < kBnJobs; ++j) { t.c8[j] = 1; t.chunks[j] = 1; }
    t.eps = eps; t.momentum = momentum;
    hipLaunchKernelGGL(bn16_apply_pre_grouped_kernel, dim3(total), dim3(256), 0, s, t);
    return check_launch();
}

int mp_f16_bn_train_bwd_stats_grouped(const mp_f16_bn_bwd_job* jobs, int n_jobs, mp_stream_t stream) {
    if (!jobs) return MP_ERR_NULL;
    if (n_jobs < 1 || n_jobs > kBnJobs) return MP_ERR_SHAPE;
    hipStream_t s = as_stream(stream);
    Bn16BwdJobs t{};
    unsigned total = 0;
    size_t fold_off = 0;
    for (int j = 0; j < kBnJobs; ++j) t.first[j] = 0xFFFFFFFFu;
    for (int j = 0; j < n_jobs; ++j) {
        const mp_f16_bn_bwd_job& q = jobs[j];
        if (!q.g_dev || !q.z_dev || !q.gamma_dev || !q.save_mean_dev || !q.save_invstd_dev || !q.dz_dev || !q.dgamma_dev || !q.dbeta_dev ||
            !q.partials_dev)
            return MP_ERR_NULL;
        if (q.n <= 0 || q.c <= 0 || q.hw <= 0 || q.n_parts <= 0) return MP_ERR_SHAPE;
        if (!q.workspace_dev || q.workspace_bytes < mp_bn_workspace_bytes(q.c)) return MP_ERR_WORKSPACE;
        const int c8 = (q.c + 7) / 8;
        const float* pre = q.partials_dev;
        int n_parts = q.n_parts;
        if (q.workspace_bytes < fold_off + (size_t)c8 * kFoldSplit * 16 * sizeof(float)) return MP_ERR_WORKSPACE;
        const int rc = bn16_prefold(pre, n_parts, c8, reinterpret_cast<char*>(q.workspace_dev) + fold_off, s);
        if (rc != MP_OK) return rc;
        fold_off += ((size_t)c8 * kFoldSplit * 16 * sizeof(float) + 255) / 256 * 256;
        const bool acc = q.dgamma_acc_dev && q.dbeta_acc_dev;
        t.g[j] = reinterpret_cast<const u32x4_t*>(q.g_dev); t.z[j] = reinterpret_cast<const u32x4_t*>(q.z_dev); t.pre[j] = pre;
        t.gamma[j] = q.gamma_dev; t.mean[j] = q.save_mean_dev; t.invstd[j] = q.save_invstd_dev; t.dgamma[j] = q.dgamma_dev;
        t.dbeta[j] = q.dbeta_dev; t.dgamma_acc[j] = acc ? q.dgamma_acc_dev : nullptr; t.dbeta_acc[j] = acc ? q.dbeta_acc_dev : nullptr;
        t.dz[j] = reinterpret_cast<u32x4_t*>(q.dz_dev);
        t.inv_count[j] = (float)(1.0 / ((double)q.n * q.hw));
        t.n_parts[j] = n_parts; t.n[j] = q.n; t.c[j] = q.c; t.c8[j] = c8; t.hw[j] = q.hw;
        t.chunks[j] = bn16_pre_chunks(q.n, c8, q.hw, n_parts);
        t.first[j] = total;
        total += (unsigned)c8 * t.chunks[j];
    }
    for (int j = n_jobs; j < kBnJobs; ++j) { t.c8[j] = 1; t.chunks[j] = 1; }
    hipLaunchKernelGGL(bn16_bwd_apply_pre_grouped_kernel, dim3(total), dim3(256), 0, s, t);
    return check_launch();
}

int mp_f16_fuse_upsample_sum_bwd(const void* dy, const void* out, void* dbase, void* dt1, int s1, void* dt2, int s2, void* dt3,
                                 int s3, int n, int c, int h, int w, int relu, mp_stream_t stream) {
    if (!dy || (relu && !out)) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    hipStream_t s = as_stream(stream);
    void* dts[4] = {dbase, dt1, dt2, dt3};
    const int ss[4] = {1, s1, s2, s3};
    const int planes = n * ((c + 7) / 8);
    for (int k = 0; k < 4; ++k) {
        if (!dts[k]) continue;
        const int sh = log2_exact_t(ss[k]);
        if (sh < 0 || (h % ss[k]) || (w % ss[k])) return MP_ERR_UNSUPPORTED;
        const size_t total = (size_t)planes * (h >> sh) * (w >> sh);
        size_t blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(fuse_sum16_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const u32x4_t*>(dy),
                           reinterpret_cast<const u32x4_t*>(out), reinterpret_cast<u32x4_t*>(dts[k]), planes, h, w, sh, relu ? 1 : 0);
        int rc = check_launch();
        if (rc != MP_OK) return rc;
    }
    return MP_OK;
}

int mp_f16_ew_stats_parts(int n, int c, int hw) {
    if (n <= 0 || c <= 0 || hw <= 0) return 0;
    return (int)bn16_apply_chunks(n, (c + 7) / 8, hw);
}

// a term at scale s sums s x s gradient elements per output element: blocks of >= 1024 / s^2 outputs (>= 64) keep the chip busy on
// the low-resolution terms
int mp_f16_fuse_term_stats_parts(int n, int c, int h, int w, int s) {
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || s <= 0 || (h % s) || (w % s)) return 0;
    const int c8 = (c + 7) / 8;
    const size_t per_blk = (size_t)n * (h / s) * (w / s);
    size_t min_elems = 1024 / ((size_t)s * s);
    if (min_elems < 64) min_elems = 64;
    size_t chunks = (768 + c8 - 1) / c8;
    while (chunks > 1 && per_blk / chunks < min_elems) --chunks;
    return (int)chunks;
}

int mp_f16_sum_tensors_stats(const void* a, const void* b, const void* c, const void* d, void* out, const void* z, const void* y,
                             int relu, int n, int ch, int hw, float* partials, size_t partials_bytes, mp_stream_t stream) {
    if (!a || !b || !out || !z || !partials || (relu && !y)) return MP_ERR_NULL;
    if (d && !c) return MP_ERR_NULL;
    if (n <= 0 || ch <= 0 || hw <= 0) return MP_ERR_SHAPE;
    const int c8 = (ch + 7) / 8, parts = mp_f16_ew_stats_parts(n, ch, hw);
    if (partials_bytes < (size_t)c8 * parts * 16 * sizeof(float)) return MP_ERR_WORKSPACE;
    hipLaunchKernelGGL(sum_tensors16_stats_kernel, dim3(c8, parts), dim3(256), 0, as_stream(stream), reinterpret_cast<const u32x4_t*>(a),
                       reinterpret_cast<const u32x4_t*>(b), reinterpret_cast<const u32x4_t*>(c), reinterpret_cast<const u32x4_t*>(d),
                       reinterpret_cast<const u32x4_t*>(z), reinterpret_cast<const u32x4_t*>(y), reinterpret_cast<u32x4_t*>(out), partials, n,
                       c8, hw, relu ? 1 : 0);
    return check_launch();
}

int mp_f16_fuse_sum_bwd_term_stats(const void* dy, const void* out, void* dt, int s, int n, int c, int h, int w, int relu, const void* z_t,
                                   const void* y_t, int relu_t, float* partials, size_t partials_bytes, mp_stream_t stream) {
    if (!dy || !dt || !z_t || !partials || (relu && !out) || (relu_t && !y_t)) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    const int sh = log2_exact_t(s);
    if (sh < 0 || (h % s) || (w % s)) return MP_ERR_UNSUPPORTED;
    const int c8 = (c + 7) / 8, parts = mp_f16_fuse_term_stats_parts(n, c, h, w, s);
    if (partials_bytes < (size_t)c8 * parts * 16 * sizeof(float)) return MP_ERR_WORKSPACE;
    if (sh > 3) return MP_ERR_UNSUPPORTED;  // HRNet's exchange units up-sample by 2, 4 and 8 (1: the identity term)
    auto kern = sh == 0 ? fuse_sum16_bwd_stats_kernel<0> : sh == 1 ? fuse_sum16_bwd_stats_kernel<1>
              : sh == 2 ? fuse_sum16_bwd_stats_kernel<2> : fuse_sum16_bwd_stats_kernel<3>;
    hipLaunchKernelGGL(kern, dim3(c8, parts), dim3(256), 0, as_stream(stream), reinterpret_cast<const u32x4_t*>(dy),
                       reinterpret_cast<const u32x4_t*>(out), reinterpret_cast<const u32x4_t*>(z_t), reinterpret_cast<const u32x4_t*>(y_t),
                       reinterpret_cast<u32x4_t*>(dt), partials, n, c8, h, w, relu ? 1 : 0, relu_t ? 1 : 0);
    return check_launch();
}

int mp_sum_tensors(const void* a, const void* b, const void* c, const void* d, void* out, size_t bytes, int half, mp_stream_t stream) {
    if (!a || !b || !out) return MP_ERR_NULL;
    if (d && !c) return MP_ERR_NULL;
    if (bytes == 0 || (bytes & 15)) return MP_ERR_SHAPE;
    const size_t units = bytes / 16;
    size_t blocks = (units + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const u32x4_t *pa = reinterpret_cast<const u32x4_t*>(a), *pb = reinterpret_cast<const u32x4_t*>(b),
                  *pc = reinterpret_cast<const u32x4_t*>(c), *pd = reinterpret_cast<const u32x4_t*>(d);
    if (half)
        hipLaunchKernelGGL(sum_tensors_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), pa, pb, pc, pd,
                           reinterpret_cast<u32x4_t*>(out), units);
    else
        hipLaunchKernelGGL(sum_tensors_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), pa, pb, pc, pd,
                           reinterpret_cast<u32x4_t*>(out), units);
    return check_launch();
}

}  // extern "C"
