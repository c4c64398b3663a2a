// HBM-bound kernels of the top-down pose hot path (gfx950): decoder, flip-test aggregation,
// Gaussian target generation, JointsMSELoss.  One 64-lane wavefront per (sample, joint) map for the
// reductions; float4 (16 B / lane) coalesced streaming for the element-wise passes.
//
// fp contraction is OFF in this file: the decoder / target arithmetic restates reference expressions
// evaluated without FMA (numpy / MindSpore CPU), and index / coordinate results are compared
// bit-for-bit with the CPU oracle.
#include "common.h"

#include <math.h>

#pragma clang fp contract(off)

namespace mp {

thread_local int g_last_hip_error = 0;
thread_local bool g_dry_launch = false;

struct DecodeParams {
    const float* hm;          // [N,K,H,W]
    const float* hf;          // flipped run output [N,K,H,W] or nullptr
    const int32_t* flip_index;
    float* avg_out;           // optional
    const float* center;
    const float* scale;
    const float* score;
    float* preds;
    float* boxes;
    int32_t* argmax;
    const float* blur;
    float* dark_terms;        // optional debug output [N*K][16] (mp_decode_topdown_debug), else nullptr
    int n, k, h, w;
    int refine, use_udp, to_original, ks, shift_heatmap;
    float pixel_std;
};

// value of the (possibly flip-aggregated) heat-map at (y, x) of one (n, k) plane
// topdown_inferencer.py:171-187: flipped_back[k][y][x] = flipped[flip_index[k]][y][W-1-x'],
// x' = x-1 for x >= 1 when shift_heatmap (column 0 keeps its own value), then (h + fb) * 0.5
template <bool FLIP>
__device__ __forceinline__ float plane_val(const float* __restrict__ a, const float* __restrict__ b, int w,
                                           int shift, int y, int x) {
    float v = a[y * w + x];
    if (FLIP) {
        int xs = (shift && x >= 1) ? x - 1 : x;
        float f = b[y * w + (w - 1 - xs)];
        v = (v + f) * 0.5f;
    }
    return v;
}

__device__ __forceinline__ void argmax_combine(float& best, int& bidx, float ov, int oi) {
    if (ov > best || (ov == best && oi < bidx)) {
        best = ov;
        bidx = oi;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One wave per (n, k) map.  top_down_decoder.py:72-205.
constexpr int kDarkFastKs = 17;                                  // largest blur kernel of the register-patch path: (17 + 2)^2 = 361 <= 6 x 64
constexpr int kDarkPatchPerLane = 6;
constexpr int kDarkTable = (kDarkFastKs + 4) * (kDarkFastKs + 4);  // the blur table with a two-wide zero border

template <bool FLIP>
__global__ __launch_bounds__(256) void decode_kernel(DecodeParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rows = p.n * p.k;
    // DARK: the k x k blur table goes to LDS ONCE per block, zero-padded by two on every side - the nine windows around the arg-max
    // then read their taps at fixed offsets from one address per patch pixel, no bounds test and no dependent global load per tap
    // (27 scattered table loads per lane were most of the refinement's 5 us)
    __shared__ float s_blur[kDarkTable];
    const bool dark_fast = p.refine == MP_REFINE_DARK && p.ks <= kDarkFastKs;
    if (dark_fast) {
        const int tw = p.ks + 4;
        for (int i = threadIdx.x; i < tw * tw; i += 256) {
            const int ty = i / tw - 2, tx = i - (i / tw) * tw - 2;
            s_blur[i] = (ty >= 0 && ty < p.ks && tx >= 0 && tx < p.ks) ? p.blur[ty * p.ks + tx] : 0.f;
        }
        __syncthreads();
    }
    if (row >= rows) return;  // whole wave exits; no block-level barrier below
    const int n = row / p.k, k = row - n * p.k;
    const int h = p.h, w = p.w, hw = h * w;
    const float* a = p.hm + (size_t)row * hw;
    const float* b = FLIP ? p.hf + ((size_t)n * p.k + p.flip_index[k]) * hw : nullptr;
    float* avg = (FLIP && p.avg_out) ? p.avg_out + (size_t)row * hw : nullptr;

    float best = -INFINITY;
    int bidx = 0x7fffffff;
    if ((hw & 3) == 0 && (!FLIP || (w & 3) == 0)) {
        // float4 streaming: lane reads 16 B at float4 index lane + 64 t (indices ascend per lane), FOUR quads requested before the
        // first is looked at (a 64x48 map is 12 quads per lane: three round trips instead of twelve).  Flip test: the four values of
        // the mirrored run that meet pixels x .. x + 3 of row y are b[y][W-1-xs], xs = x - 1 (x >= 1, shift) - four consecutive
        // floats read backwards, one element off 16-byte alignment when shifted: scalar loads of one cache line
        const float4* a4 = reinterpret_cast<const float4*>(a);
        float4* avg4 = reinterpret_cast<float4*>(avg);
        const int nq = hw >> 2, wq = w >> 2;
        // (twelve quads in flight per lane - the whole 64x48 map in one round trip - measured no faster than four: 8.9 vs 8.1 us
        // HBM-resident, 5.4 vs 5.1 us cache-resident)
        constexpr int B = 4;
        for (int q0 = lane; q0 < nq; q0 += 64 * B) {
            float4 v[B], f[B];
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const int q = q0 + 64 * j;
                if (q < nq) {
                    v[j] = a4[q];
                    if (FLIP) {
                        const int y = q / wq, x = (q - y * wq) << 2;
                        const float* br = b + y * w;
                        const int s = p.shift_heatmap ? 1 : 0;
                        f[j].x = br[w - 1 - (x >= 1 ? x - s : x)];
                        f[j].y = br[w - 1 - (x + 1 - s)];
                        f[j].z = br[w - 1 - (x + 2 - s)];
                        f[j].w = br[w - 1 - (x + 3 - s)];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const int q = q0 + 64 * j;
                if (q < nq) {
                    float4 u = v[j];
                    if (FLIP) {  // the expression of plane_val: (v + f) * 0.5f
                        u.x = (u.x + f[j].x) * 0.5f; u.y = (u.y + f[j].y) * 0.5f; u.z = (u.z + f[j].z) * 0.5f; u.w = (u.w + f[j].w) * 0.5f;
                        if (avg) avg4[q] = u;
                    }
                    const int i = q << 2;
                    if (u.x > best) { best = u.x; bidx = i; }
                    if (u.y > best) { best = u.y; bidx = i + 1; }
                    if (u.z > best) { best = u.z; bidx = i + 2; }
                    if (u.w > best) { best = u.w; bidx = i + 3; }
                }
            }
        }
    } else {
        for (int y = 0; y < h; ++y) {
            for (int x = lane; x < w; x += 64) {
                float v = plane_val<FLIP>(a, b, w, p.shift_heatmap, y, x);
                if (avg) avg[y * w + x] = v;
                if (v > best) { best = v; bidx = y * w + x; }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ov = __shfl_xor(best, off, 64);
        int oi = __shfl_xor(bidx, off, 64);
        argmax_combine(best, bidx, ov, oi);
    }
    if (bidx == 0x7fffffff) {  // all -inf / NaN map: numpy argmax -> 0
        bidx = 0;
        best = plane_val<FLIP>(a, b, w, p.shift_heatmap, 0, 0);
    }
    const int yi = bidx / w, xi = bidx - yi * w;
    float cx = (float)xi, cy = (float)yi;  // :111-112 (exact: idx < 2^24)

    if (p.refine == MP_REFINE_SHIFT) {
        // :118-141  dx defined for 1 <= x <= W-2 (any y), dy for 1 <= y <= H-2 (any x)
        float dx = 0.f, dy = 0.f;
        if (xi >= 1 && xi <= w - 2)
            dx = plane_val<FLIP>(a, b, w, p.shift_heatmap, yi, xi + 1) - plane_val<FLIP>(a, b, w, p.shift_heatmap, yi, xi - 1);
        if (yi >= 1 && yi <= h - 2)
            dy = plane_val<FLIP>(a, b, w, p.shift_heatmap, yi + 1, xi) - plane_val<FLIP>(a, b, w, p.shift_heatmap, yi - 1, xi);
        float sx = (dx > 0.f) ? 1.f : ((dx < 0.f) ? -1.f : 0.f);
        float sy = (dy > 0.f) ? 1.f : ((dy < 0.f) ? -1.f : 0.f);
        cx = cx + sx * 0.25f;
        cy = cy + sy * 0.25f;
    } else if (p.refine == MP_REFINE_DARK) {
        // :171-205  only the 3x3 neighbourhood of the arg-max needs the k x k blur
        const int ks = p.ks, r = ks >> 1, taps = ks * ks;
        float L[3][3];
        const int side = ks + 2;  // the nine blurred values read a (ks + 2)^2 patch around the arg-max
        if (dark_fast) {
            // every pixel of the patch is fetched ONCE (<= 6 per lane, all requested before the first is used; zero outside the map =
            // the blur's padding) and feeds the windows of all nine positions from registers: window (oy, ox) meets patch pixel
            // (iy, ix) with tap (iy - 1 - oy, ix - 1 - ox) = padded-table entry (iy + 1 - oy, ix + 1 - ox).  Nine wave sums.
            const int tw = ks + 4, np = side * side;
            float pv[kDarkPatchPerLane];
            int tb[kDarkPatchPerLane];
#pragma unroll
            for (int e = 0; e < kDarkPatchPerLane; ++e) {
                const int idx = min(lane + 64 * e, np - 1);
                const int iy = idx / side, ix = idx - iy * side;
                const int yy = yi + iy - (r + 1), xx = xi + ix - (r + 1);
                const bool ok = lane + 64 * e < np && yy >= 0 && yy < h && xx >= 0 && xx < w;
                pv[e] = ok ? plane_val<FLIP>(a, b, w, p.shift_heatmap, yy, xx) : 0.f;
                tb[e] = (iy + 1) * tw + ix + 1;
            }
            float part[3][3];
#pragma unroll
            for (int oy = 0; oy < 3; ++oy)
#pragma unroll
                for (int ox = 0; ox < 3; ++ox) part[oy][ox] = 0.f;
#pragma unroll
            for (int e = 0; e < kDarkPatchPerLane; ++e) {
                if (64 * e >= np) break;  // wave-uniform
#pragma unroll
                for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
                    for (int ox = -1; ox <= 1; ++ox) part[oy + 1][ox + 1] += s_blur[tb[e] - oy * tw - ox] * pv[e];
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1)  // the nine butterflies advance together (independent chains)
#pragma unroll
                for (int i = 0; i < 9; ++i) part[i / 3][i % 3] += __shfl_xor(part[i / 3][i % 3], off, 64);
#pragma unroll
            for (int oy = -1; oy <= 1; ++oy) {
#pragma unroll
                for (int ox = -1; ox <= 1; ++ox) {
                    const int py = yi + oy, px = xi + ox;
                    const bool inside = (py >= 0 && py < h && px >= 0 && px < w);
                    // clip [1e-3, 50] -> log ; positions outside the map are the zero pad applied AFTER log
                    L[oy + 1][ox + 1] = inside ? logf(fminf(fmaxf(part[oy + 1][ox + 1], 0.001f), 50.f)) : 0.f;
                }
            }
        } else {
#pragma unroll
        for (int oy = -1; oy <= 1; ++oy) {
#pragma unroll
            for (int ox = -1; ox <= 1; ++ox) {
                const int py = yi + oy, px = xi + ox;
                float part = 0.f;
                const bool inside = (py >= 0 && py < h && px >= 0 && px < w);
                if (inside) {
                    for (int t = lane; t < taps; t += 64) {
                        int ti = t / ks, tj = t - ti * ks;
                        int yy = py + ti - r, xx = px + tj - r;
                        if (yy >= 0 && yy < h && xx >= 0 && xx < w)
                            part += p.blur[t] * plane_val<FLIP>(a, b, w, p.shift_heatmap, yy, xx);
                    }
                }
                float s = wave_sum(part);
                // clip [1e-3, 50] -> log ; positions outside the map are the zero pad applied AFTER log
                L[oy + 1][ox + 1] = inside ? logf(fminf(fmaxf(s, 0.001f), 50.f)) : 0.f;
            }
        }
        }
        const float i_ = L[1][1], ix1 = L[1][2], ix1_ = L[1][0], iy1 = L[2][1], iy1_ = L[0][1];
        const float ix1y1 = L[2][2], ix1_y1_ = L[0][0];
        const float dx = 0.5f * (ix1 - ix1_);
        const float dy = 0.5f * (iy1 - iy1_);
        const float dxx = ix1 - 2.f * i_ + ix1_;
        const float dyy = iy1 - 2.f * i_ + iy1_;
        const float dxy = 0.5f * (ix1y1 - ix1 - iy1 + i_ + i_ - ix1_ - iy1_ + ix1_y1_);
        const float ha = dxx + 1e-7f, hd = dyy + 1e-7f;  // Hessian + 1e-7 I
        const float det = ha * hd - dxy * dxy;
        if (p.dark_terms && lane == 0) {  // intermediates of the refinement for the parity tests (uniform branch)
            float* t = p.dark_terms + (size_t)row * 16;
#pragma unroll
            for (int i = 0; i < 9; ++i) t[i] = L[i / 3][i % 3];
            t[9] = dx; t[10] = dy; t[11] = dxx; t[12] = dyy; t[13] = dxy; t[14] = det; t[15] = 0.f;
        }
        cx = cx - (hd * dx - dxy * dy) / det;
        cy = cy - (ha * dy - dxy * dx) / det;
    }

    if (lane == 0) {
        const float sxs = p.scale[n * 2 + 0], sys = p.scale[n * 2 + 1];
        const float ctx = p.center[n * 2 + 0], cty = p.center[n * 2 + 1];
        if (p.to_original) {
            // :143-169
            const float s_x = sxs * p.pixel_std, s_y = sys * p.pixel_std;
            const float den_x = p.use_udp ? (float)(w - 1) : (float)w;
            const float den_y = p.use_udp ? (float)(h - 1) : (float)h;
            const float kx = s_x / den_x, ky = s_y / den_y;
            cx = cx * kx + ctx - s_x * 0.5f;
            cy = cy * ky + cty - s_y * 0.5f;
        }
        float* pr = p.preds + (size_t)row * 3;
        pr[0] = cx;
        pr[1] = cy;
        pr[2] = best;
        if (p.argmax) p.argmax[row] = bidx;
        if (k == 0) {
            float* bx = p.boxes + (size_t)n * 6;  // :88-92
            bx[0] = ctx;
            bx[1] = cty;
            bx[2] = sxs;
            bx[3] = sys;
            bx[4] = (sxs * p.pixel_std) * (sys * p.pixel_std);
            bx[5] = p.score[n];
        }
    }
}

// stand-alone aggregation (EvalNet output_raw wants the averaged heat-map): block per (n, k) plane
__global__ __launch_bounds__(256) void flip_aggregate_kernel(const float* __restrict__ hm, const float* __restrict__ hf,
                                                             const int32_t* __restrict__ flip_index,
                                                             float* __restrict__ out, int k, int h, int w, int shift) {
    const int row = blockIdx.x;  // (n, k)
    const int nn = row / k, kk = row - nn * k;
    const int hw = h * w;
    const float* a = hm + (size_t)row * hw;
    const float* b = hf + ((size_t)nn * k + flip_index[kk]) * hw;
    float* o = out + (size_t)row * hw;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) {
        int y = i / w, x = i - y * w;
        o[i] = plane_val<true>(a, b, w, shift, y, x);
    }
}

// ------------------------------------------------------------------------------------------
// Gaussian target: one 256-thread block per (n, k) plane; zero stores of the whole plane (16 B / lane
// when W % 4 == 0) first, the stamp's pixels behind them.  topdown_transform.py:324-430.
// ------------------------------------------------------------------------------------------
struct TargetParams {
    const float* kp;
    const float* patch;
    const double* jw;
    float* target;
    float* tw;
    int n, k, h, w, side, use_udp;
    double fsx, fsy, sigma;
};

__device__ __forceinline__ float target_value(const TargetParams& p, bool stamp, int y, int x, int ix0, int ix1,
                                              int iy0, int iy1, int gx0, int gy0, double x0p, double y0p,
                                              double two_sigma2) {
    if (!stamp || x < ix0 || x >= ix1 || y < iy0 || y >= iy1) return 0.f;
    int gx = gx0 + (x - ix0), gy = gy0 + (y - iy0);
    if (p.use_udp) {
        double dx = (double)gx - x0p, dy = (double)gy - y0p;
        double e = -(dx * dx + dy * dy) / two_sigma2;
        return (float)exp(e);
    }
    gx = min(gx, p.side - 1);
    gy = min(gy, p.side - 1);
    return p.patch[gy * p.side + gx];
}

__global__ __launch_bounds__(256) void gaussian_target_kernel(TargetParams p) {
    const int row = blockIdx.x;
    const int k = row % p.k;
    const int h = p.h, w = p.w;
    float* out = p.target + (size_t)row * h * w;
    // The plane is zeros but for one (6 sigma + 1)^2 stamp, and WHERE the stamp lies takes a chain of fp64 divisions and roundings:
    // the zero stores go out first - nothing of them depends on the key point - and the set-up runs under them (round 4 did the
    // set-up first: 8.2 us for 26.7 MB of stores, 0.40 of the HBM peak).  The stamp's pixels are stored a second time behind a
    // workgroup barrier (same block, so the barrier's release orders them behind the zeros): <= 5 % more bytes.
    if ((w & 3) == 0) {
        const int nq = (h * w) >> 2;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = threadIdx.x; q < nq; q += blockDim.x) reinterpret_cast<float4*>(out)[q] = z4;
    } else {
        for (int i = threadIdx.x; i < h * w; i += blockDim.x) out[i] = 0.f;
    }
    const float kx = p.kp[(size_t)row * 3 + 0], ky = p.kp[(size_t)row * 3 + 1], vis = p.kp[(size_t)row * 3 + 2];
    double fx = (double)kx / p.fsx, fy = (double)ky / p.fsy;  // fp32 scalar / fp64 scalar -> fp64
    fx = fmin(fmax(fx, -1.0e9), 1.0e9);
    fy = fmin(fmax(fy, -1.0e9), 1.0e9);
    int mu_x, mu_y;
    if (p.use_udp) {
        mu_x = (int)(fx + 0.5);  // :399-400 int() truncates toward zero
        mu_y = (int)(fy + 0.5);
    } else {
        mu_x = (int)rint(fx);  // :350-351 Python round(): half-to-even
        mu_y = (int)rint(fy);
    }
    const double tmp = p.sigma * 3.0;
    const int ulx = (int)((double)mu_x - tmp), uly = (int)((double)mu_y - tmp);
    const int brx = (int)((double)mu_x + tmp + 1.0), bry = (int)((double)mu_y + tmp + 1.0);
    const bool oob = (ulx >= w) || (uly >= h) || (brx < 0) || (bry < 0);
    float weight = oob ? 0.f : vis;
    const bool stamp = (!oob) && (weight > 0.5f);
    const int gx0 = max(0, -ulx), gy0 = max(0, -uly);
    const int ix0 = max(0, ulx), ix1 = min(brx, w);
    const int iy0 = max(0, uly), iy1 = min(bry, h);
    const double size = 2.0 * tmp + 1.0;
    const double c0 = floor(size / 2.0);  // size // 2
    const double x0p = c0 + fx - (double)mu_x, y0p = c0 + fy - (double)mu_y;
    const double two_sigma2 = 2.0 * (p.sigma * p.sigma);

    __syncthreads();  // (uniform: every thread of the block gets here)
    if (stamp && ix1 > ix0 && iy1 > iy0) {
        const int sw = ix1 - ix0, cnt = sw * (iy1 - iy0);
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int dy = i / sw, y = iy0 + dy, x = ix0 + (i - dy * sw);
            out[y * w + x] = target_value(p, true, y, x, ix0, ix1, iy0, iy1, gx0, gy0, x0p, y0p, two_sigma2);
        }
    }
    if (threadIdx.x == 0) {
        if (p.jw) weight = (float)((double)weight * p.jw[k]);  // :371-372 np.multiply(fp32, fp64)
        p.tw[row] = weight;
    }
}

// ------------------------------------------------------------------------------------------
// JointsMSELoss: block per (n, k) row -> fp32 partial (w * d^2 accumulated per element), then one
// block sums the N*K partials in fp64 in a fixed order (deterministic).  mse.py:36-44.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_row_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                      const float* __restrict__ wgt, float* __restrict__ partial,
                                                      int hw) {
    const int row = blockIdx.x;
    const float wv = wgt ? wgt[row] : 1.f;
    const float* a = pred + (size_t)row * hw;
    const float* b = tgt + (size_t)row * hw;
    float acc = 0.f;
    if ((hw & 3) == 0) {
        const float4* a4 = reinterpret_cast<const float4*>(a);
        const float4* b4 = reinterpret_cast<const float4*>(b);
        for (int q = threadIdx.x; q < (hw >> 2); q += blockDim.x) {
            float4 x = a4[q], y = b4[q];
            float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
            acc += (d0 * d0) * wv;
            acc += (d1 * d1) * wv;
            acc += (d2 * d2) * wv;
            acc += (d3 * d3) * wv;
        }
    } else {
        for (int i = threadIdx.x; i < hw; i += blockDim.x) {
            float d = a[i] - b[i];
            acc += (d * d) * wv;
        }
    }
    acc = wave_sum(acc);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[row] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ __launch_bounds__(256) void mse_final_kernel(const float* __restrict__ partial, float* __restrict__ loss,
                                                        int rows, double inv_count) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < rows; i += blockDim.x) acc += (double)partial[i];
    __shared__ double sm[256];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sm[0] * inv_count);
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                      const float* __restrict__ wgt, const float* __restrict__ gout,
                                                      float* __restrict__ gpred, int hw, float two_over_count) {
    const int row = blockIdx.x;
    const float g = (gout ? gout[0] : 1.f) * two_over_count;
    const float c = wgt ? wgt[row] * g : g;
    const float* a = pred + (size_t)row * hw;
    const float* b = tgt + (size_t)row * hw;
    float* o = gpred + (size_t)row * hw;
    if ((hw & 3) == 0) {
        for (int q = threadIdx.x; q < (hw >> 2); q += blockDim.x) {
            float4 x = reinterpret_cast<const float4*>(a)[q], y = reinterpret_cast<const float4*>(b)[q];
            float4 r;
            r.x = (x.x - y.x) * c;
            r.y = (x.y - y.y) * c;
            r.z = (x.z - y.z) * c;
            r.w = (x.w - y.w) * c;
            reinterpret_cast<float4*>(o)[q] = r;
        }
    } else {
        for (int i = threadIdx.x; i < hw; i += blockDim.x) o[i] = (a[i] - b[i]) * c;
    }
}

// nn.MaxPool2d(3, 2, pad_mode="same") resnet.py:190: out = ceil(in/2); pad only bottom/right (even H, W).
__global__ __launch_bounds__(256) void maxpool3x3s2_same_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                int planes, int h, int w, int oh, int ow, int pt,
                                                                int pl) {
    const size_t total = (size_t)planes * oh * ow;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int ox = (int)(i % ow);
        size_t t = i / ow;
        int oy = (int)(t % oh);
        size_t pl_i = t / oh;
        const float* src = x + pl_i * (size_t)h * w;
        float m = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            int yy = oy * 2 - pt + dy;
            if (yy < 0 || yy >= h) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                int xx = ox * 2 - pl + dx;
                if (xx < 0 || xx >= w) continue;
                m = fmaxf(m, src[yy * w + xx]);
            }
        }
        out[i] = m;
    }
}

// HRModule exchange unit, up-sampling side: out = act(((base + up(t1)) + up(t2)) + up(t3)), nearest up-sampling by
// s_k = 1 << sh_k.  One thread per 4 output columns (16 B loads/stores of base/out); a low-resolution term
// contributes one value per s_k columns, served from L2 (each low-res row is re-read by s_k output rows).
struct FuseSumParams {
    const float* base;
    const float* t[3];
    float* out;
    int sh[3];  // log2(scale), -1 = absent
    int n_planes, h, w, relu;
};

__global__ __launch_bounds__(256) void fuse_sum_scalar_kernel(FuseSumParams p) {
    const size_t total = (size_t)p.n_planes * p.h * p.w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.w);
        const size_t r = i / p.w;
        const int y = (int)(r % p.h);
        const size_t plane = r / p.h;
        float v = p.base[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (p.sh[k] < 0) continue;
            const int sh = p.sh[k];
            v += p.t[k][(plane * (p.h >> sh) + (y >> sh)) * (p.w >> sh) + (x >> sh)];
        }
        if (p.relu) v = fmaxf(v, 0.f);
        p.out[i] = v;
    }
}

__global__ __launch_bounds__(256) void fuse_sum_kernel(FuseSumParams p) {
    const int wq = p.w >> 2;
    const size_t total = (size_t)p.n_planes * p.h * wq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xq = (int)(i % wq);
        const size_t r = i / wq;
        const int y = (int)(r % p.h);
        const size_t plane = r / p.h;
        const size_t o = (plane * p.h + y) * p.w + (size_t)xq * 4;
        float4 v = *reinterpret_cast<const float4*>(p.base + o);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (p.sh[k] < 0) continue;
            const int sh = p.sh[k];
            const int lw = p.w >> sh, lh = p.h >> sh;
            const float* row = p.t[k] + (plane * lh + (y >> sh)) * lw;
            const int x0 = xq * 4;
            v.x += row[(x0 + 0) >> sh];
            v.y += row[(x0 + 1) >> sh];
            v.z += row[(x0 + 2) >> sh];
            v.w += row[(x0 + 3) >> sh];
        }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(p.out + o) = v;
    }
}

static int launch_decode(const DecodeParams& p, hipStream_t s) {
    const int rows = p.n * p.k;
    dim3 grid((rows + 3) / 4), block(256);
    if (p.hf)
        hipLaunchKernelGGL(decode_kernel<true>, grid, block, 0, s, p);
    else
        hipLaunchKernelGGL(decode_kernel<false>, grid, block, 0, s, p);
    return check_launch();
}

static int validate_decode(const float* heatmap, const float* center, const float* scale, const float* score,
                           float* preds, float* boxes, int n, int k, int h, int w, int refine_mode,
                           const float* blur, int ks) {
    if (!heatmap || !center || !scale || !score || !preds || !boxes) return MP_ERR_NULL;
    if (n <= 0 || k <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    if ((long long)h * w >= (1 << 24)) return MP_ERR_UNSUPPORTED;
    if (refine_mode < MP_REFINE_NONE || refine_mode > MP_REFINE_DARK) return MP_ERR_UNSUPPORTED;
    if (refine_mode == MP_REFINE_DARK) {
        if (!blur) return MP_ERR_NULL;
        if (ks < 1 || (ks & 1) == 0 || ks > 63) return MP_ERR_UNSUPPORTED;
    }
    return MP_OK;
}

}  // namespace mp

using namespace mp;

extern "C" {

const char* mp_version(void) { return "mindpose_hip 0.4.0 (gfx950)"; }

const char* mp_error_string(int code) {
    switch (code) {
        case MP_OK: return "ok";
        case MP_ERR_NULL: return "required pointer is NULL";
        case MP_ERR_SHAPE: return "bad shape";
        case MP_ERR_UNSUPPORTED: return "unsupported configuration";
        case MP_ERR_HIP: return "HIP runtime error";
        case MP_ERR_WORKSPACE: return "workspace missing or too small";
        default: return "unknown error";
    }
}

int mp_last_hip_error(void) { return g_last_hip_error; }

int mp_decode_topdown(const float* heatmap, const float* center, const float* scale, const float* score, float* preds,
                      float* boxes, int32_t* argmax_idx, int n, int k, int h, int w, int refine_mode, int use_udp,
                      int to_original, float pixel_std, const float* blur_kernel, int kernel_size,
                      mp_stream_t stream) {
    int rc = validate_decode(heatmap, center, scale, score, preds, boxes, n, k, h, w, refine_mode, blur_kernel,
                             kernel_size);
    if (rc != MP_OK) return rc;
    DecodeParams p{};
    p.hm = heatmap; p.hf = nullptr; p.flip_index = nullptr; p.avg_out = nullptr;
    p.center = center; p.scale = scale; p.score = score; p.preds = preds; p.boxes = boxes; p.argmax = argmax_idx;
    p.blur = blur_kernel; p.n = n; p.k = k; p.h = h; p.w = w; p.refine = refine_mode; p.use_udp = use_udp;
    p.to_original = to_original; p.ks = kernel_size; p.shift_heatmap = 0; p.pixel_std = pixel_std;
    return launch_decode(p, as_stream(stream));
}

int mp_decode_topdown_debug(const float* heatmap, const float* center, const float* scale, const float* score, float* preds,
                            float* boxes, int32_t* argmax_idx, int n, int k, int h, int w, int refine_mode, int use_udp,
                            int to_original, float pixel_std, const float* blur_kernel, int kernel_size, float* dark_terms,
                            mp_stream_t stream) {
    int rc = validate_decode(heatmap, center, scale, score, preds, boxes, n, k, h, w, refine_mode, blur_kernel,
                             kernel_size);
    if (rc != MP_OK) return rc;
    if (refine_mode != MP_REFINE_DARK) return MP_ERR_UNSUPPORTED;
    if (!dark_terms) return MP_ERR_NULL;
    DecodeParams p{};
    p.hm = heatmap; p.hf = nullptr; p.flip_index = nullptr; p.avg_out = nullptr;
    p.center = center; p.scale = scale; p.score = score; p.preds = preds; p.boxes = boxes; p.argmax = argmax_idx;
    p.blur = blur_kernel; p.dark_terms = dark_terms; p.n = n; p.k = k; p.h = h; p.w = w; p.refine = refine_mode;
    p.use_udp = use_udp; p.to_original = to_original; p.ks = kernel_size; p.shift_heatmap = 0; p.pixel_std = pixel_std;
    return launch_decode(p, as_stream(stream));
}

int mp_flip_aggregate(const float* heatmap, const float* flipped, const int32_t* flip_index, float* avg_out, int n,
                      int k, int h, int w, int shift_heatmap, mp_stream_t stream) {
    if (!heatmap || !flipped || !flip_index || !avg_out) return MP_ERR_NULL;
    if (n <= 0 || k <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    if ((long long)n * k > 0x7fffffffLL) return MP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(flip_aggregate_kernel, dim3(n * k), dim3(256), 0, as_stream(stream), heatmap, flipped,
                       flip_index, avg_out, k, h, w, shift_heatmap ? 1 : 0);
    return check_launch();
}

int mp_flip_aggregate_decode(const float* heatmap, const float* flipped, const int32_t* flip_index, int shift_heatmap,
                             float* avg_out, const float* center, const float* scale, const float* score, float* preds,
                             float* boxes, int32_t* argmax_idx, int n, int k, int h, int w, int refine_mode,
                             int use_udp, int to_original, float pixel_std, const float* blur_kernel, int kernel_size,
                             mp_stream_t stream) {
    if (!flipped || !flip_index) return MP_ERR_NULL;
    int rc = validate_decode(heatmap, center, scale, score, preds, boxes, n, k, h, w, refine_mode, blur_kernel,
                             kernel_size);
    if (rc != MP_OK) return rc;
    DecodeParams p{};
    p.hm = heatmap; p.hf = flipped; p.flip_index = flip_index; p.avg_out = avg_out;
    p.center = center; p.scale = scale; p.score = score; p.preds = preds; p.boxes = boxes; p.argmax = argmax_idx;
    p.blur = blur_kernel; p.n = n; p.k = k; p.h = h; p.w = w; p.refine = refine_mode; p.use_udp = use_udp;
    p.to_original = to_original; p.ks = kernel_size; p.shift_heatmap = shift_heatmap ? 1 : 0; p.pixel_std = pixel_std;
    return launch_decode(p, as_stream(stream));
}

int mp_gaussian_target(const float* keypoints, const float* patch, int patch_side, const double* joint_weights,
                       float* target, float* target_weight, int n, int k, int h, int w, double feat_stride_x,
                       double feat_stride_y, double sigma, int use_udp, mp_stream_t stream) {
    if (!keypoints || !target || !target_weight) return MP_ERR_NULL;
    if (!use_udp && !patch) return MP_ERR_NULL;
    if (n <= 0 || k <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    if (!(sigma > 0.0) || !(feat_stride_x > 0.0) || !(feat_stride_y > 0.0)) return MP_ERR_SHAPE;
    if (!use_udp && patch_side <= 0) return MP_ERR_SHAPE;
    TargetParams p{};
    p.kp = keypoints; p.patch = patch; p.jw = joint_weights; p.target = target; p.tw = target_weight;
    p.n = n; p.k = k; p.h = h; p.w = w; p.side = patch_side; p.use_udp = use_udp ? 1 : 0;
    p.fsx = feat_stride_x; p.fsy = feat_stride_y; p.sigma = sigma;
    hipLaunchKernelGGL(gaussian_target_kernel, dim3(n * k), dim3(256), 0, as_stream(stream), p);
    return check_launch();
}

size_t mp_joints_mse_workspace_bytes(int n, int k) {
    if (n <= 0 || k <= 0) return 0;
    return ((size_t)n * k * sizeof(float) + 255) & ~(size_t)255;
}

int mp_joints_mse_fwd(const float* pred, const float* target, const float* weight, float* loss, void* workspace,
                      size_t workspace_bytes, int n, int k, int hw, mp_stream_t stream) {
    if (!pred || !target || !loss) return MP_ERR_NULL;
    if (n <= 0 || k <= 0 || hw <= 0) return MP_ERR_SHAPE;
    if (!workspace || workspace_bytes < mp_joints_mse_workspace_bytes(n, k)) return MP_ERR_WORKSPACE;
    float* partial = reinterpret_cast<float*>(workspace);
    const int rows = n * k;
    hipLaunchKernelGGL(mse_row_kernel, dim3(rows), dim3(256), 0, as_stream(stream), pred, target, weight, partial, hw);
    int rc = check_launch();
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, loss, rows,
                       1.0 / ((double)rows * (double)hw));
    return check_launch();
}

int mp_joints_mse_bwd(const float* pred, const float* target, const float* weight, const float* grad_out,
                      float* grad_pred, int n, int k, int hw, mp_stream_t stream) {
    if (!pred || !target || !grad_pred) return MP_ERR_NULL;
    if (n <= 0 || k <= 0 || hw <= 0) return MP_ERR_SHAPE;
    const float c = (float)(2.0 / ((double)n * k * (double)hw));
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(n * k), dim3(256), 0, as_stream(stream), pred, target, weight, grad_out,
                       grad_pred, hw, c);
    return check_launch();
}

static int log2_exact(int v) {
    for (int i = 0; i < 16; ++i)
        if ((1 << i) == v) return i;
    return -1;
}

int mp_fuse_upsample_sum(const float* base, const float* t1, int s1, const float* t2, int s2, const float* t3, int s3,
                         float* out, int n, int c, int h, int w, int relu, mp_stream_t stream) {
    if (!base || !t1 || !out) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    FuseSumParams p{};
    p.base = base; p.out = out; p.n_planes = n * c; p.h = h; p.w = w; p.relu = relu ? 1 : 0;
    const float* ts[3] = {t1, t2, t3};
    const int ss[3] = {s1, s2, s3};
    for (int k = 0; k < 3; ++k) {
        p.t[k] = ts[k];
        p.sh[k] = -1;
        if (!ts[k]) continue;
        const int sh = log2_exact(ss[k]);
        if (sh < 0 || (h % ss[k]) || (w % ss[k])) return MP_ERR_UNSUPPORTED;
        p.sh[k] = sh;
    }
    const bool vec = (w & 3) == 0;
    const size_t total = vec ? (size_t)n * c * h * (w >> 2) : (size_t)n * c * h * w;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (vec) hipLaunchKernelGGL(fuse_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(fuse_sum_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p);
    return check_launch();
}

int mp_maxpool3x3s2_same(const float* x, float* out, int n, int c, int h, int w, mp_stream_t stream) {
    if (!x || !out) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    const int oh = (h + 1) / 2, ow = (w + 1) / 2;
    const int ph = max((oh - 1) * 2 + 3 - h, 0), pw = max((ow - 1) * 2 + 3 - w, 0);
    const size_t total = (size_t)n * c * oh * ow;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(maxpool3x3s2_same_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, out, n * c, h, w, oh,
                       ow, ph / 2, pw / 2);
    return check_launch();
}

}  // extern "C"
