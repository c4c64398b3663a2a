// Convolution weight gradient on the fp32 matrix cores (gfx950):
//   dW[co][ci][ky][kx] = sum_{n,y,x} dz[n,co,y,x] * x[n,ci,y*S+ky-pad, x*S+kx-pad]
// One GEMM per tap t: dW_t[Cout x Cin] = dZ[Cout x P] * X_t^T[P x Cin], P = N*Ho*Wo, on v_mfma_f32_16x16x4_f32
// (lane l feeds A[cout l&15][pixel l>>4] and B[pixel l>>4][cin l&15]; K = 4 pixels per MFMA).
// Workgroup = 32 couts x 32 cins x all taps; wave w owns the 16x16 sub-tile (w&1, w>>1) for every tap
// (T accumulators).  The pixel axis is split over blockIdx.y; every split writes its partial dW into its own
// slab and a second kernel sums the slabs in a fixed order (deterministic, no atomics).
#include <stdlib.h>

#include "common.h"

namespace mp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradParams {
    const float* x;    // [N,Cin,H,W]
    const float* dz;   // [N,Cout,Ho,Wo]
    float* slabs;      // [splits][Cout][Cin][T]
    int N, Cin, H, W, Cout, Ho, Wo, pad;
    int R;             // output rows per pixel tile
    int Rin, Wp;       // staged input rows / pitch
    int xplane;        // per-cin LDS plane (odd)
    int zpitch;        // per-cout LDS pitch of the dz tile (odd)
    int tiles_y, n_tiles, splits;
    int ci_tiles;
    int vec;           // 1: pipelined kernel with 16-B range-checked staging (needs W % 4 == 0 and Wo % 4 == 0)
    int zq, xw4;       // 16-B units per cout row of the dz tile (R*Wo/4) / per input row (W/4)
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOobW = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes < 0x7FFFFFF0u ? bytes : 0x7FFFFFF0u), 0x00020000);
}
__device__ __forceinline__ f32x4 wg_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// Pipelined variant: the dz tile [32 couts][R*Wo] and the input tile [32 cins][Rin][Wp] are double-buffered in LDS;
// while the MFMA loop runs on tile t the NZ + NX 16-B range-checked buffer loads of tile t+1 are in flight into
// registers (rows outside the image / channels beyond Cout, Cin read 0 with no branch), one barrier per tile.
// LDS pitches are == 2 (mod 32) floats: the 16 channels x 2 adjacent pixels a half-wave reads hit 32 distinct banks.
template <int KS, int S, int NZ, int NX>
__global__ __launch_bounds__(256, 2) void conv_wgrad_pipe_kernel(const WgradParams p) {
    constexpr int T = KS * KS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int zbuf = 32 * p.zpitch, xbuf = 32 * p.xplane;
    float* __restrict__ lz = smem;              // [2][32][zpitch]
    float* __restrict__ lx = smem + 2 * zbuf;   // [2][32][xplane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_sub = wave & 1, ci_sub = wave >> 1;
    const int lq = lane >> 4, lr = lane & 15;
    const int cot = blockIdx.x / p.ci_tiles, cit = blockIdx.x % p.ci_tiles;
    const int co0 = cot * 32, ci0 = cit * 32;
    const int HW = p.H * p.W, HoWo = p.Ho * p.Wo;

    // zero both x buffers once (halo columns are never written) and both z buffers (pad pixels beyond a short tile)
    for (int i = tid; i < 2 * (zbuf + xbuf); i += 256) smem[i] = 0.f;

    // staging tables (tile independent)
    unsigned zsrc[NZ], zdst[NZ];  // float offsets; zdst packs the pixel offset (for the rows < R check) in the top bits
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        const int u = tid + 256 * i;
        zsrc[i] = kOobW;
        zdst[i] = 0;
        if (u < 32 * p.zq) {
            const int c = u / p.zq, q4 = (u - c * p.zq) * 4;
            if (co0 + c < p.Cout) zsrc[i] = (unsigned)((co0 + c) * HoWo + q4);
            zdst[i] = ((unsigned)q4 << 16) | (unsigned)(c * p.zpitch + q4);
        }
    }
    unsigned xsrc[NX], xdst[NX];  // xdst packs the input row index in the top bits
    const int xper = p.Rin * p.xw4;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int u = tid + 256 * i;
        xsrc[i] = kOobW;
        xdst[i] = 0;
        if (u < 32 * xper) {
            const int c = u / xper, rem = u - c * xper;
            const int r = rem / p.xw4, x4 = (rem - r * p.xw4) * 4;
            if (ci0 + c < p.Cin) xsrc[i] = (unsigned)((ci0 + c) * HW + r * p.W + x4);
            xdst[i] = ((unsigned)r << 16) | (unsigned)(c * p.xplane + r * p.Wp + p.pad + x4);
        }
    }

    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 vz[NZ], vx[NX];
    auto tile_geom = [&](int tile, int& n, int& y0, int& npix, int& yin0) {
        n = tile / p.tiles_y;
        y0 = (tile - n * p.tiles_y) * p.R;
        npix = min(p.R, p.Ho - y0) * p.Wo;
        yin0 = y0 * S - p.pad;
    };
    auto stage_load = [&](int tile) {
        int n, y0, npix, yin0;
        tile_geom(tile, n, y0, npix, yin0);
        const __amdgpu_buffer_rsrc_t rz = wg_rsrc(p.dz + (size_t)n * p.Cout * HoWo, (size_t)p.Cout * HoWo * 4);
        const __amdgpu_buffer_rsrc_t rx = wg_rsrc(p.x + (size_t)n * p.Cin * HW, (size_t)p.Cin * HW * 4);
        unsigned oz = 0;
        asm volatile("" : "+v"(oz));  // keep the per-unit field arithmetic out of loop-invariant hoisting
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const bool ok = zsrc[i] != kOobW && (int)((zdst[i] + oz) >> 16) < npix;
            vz[i] = wg_load4(rz, ok ? (zsrc[i] + (unsigned)(y0 * p.Wo)) * 4u : kOobW);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int yin = yin0 + (int)((xdst[i] + oz) >> 16);
            const bool ok = xsrc[i] != kOobW && (unsigned)yin < (unsigned)p.H;
            vx[i] = wg_load4(rx, ok ? (unsigned)((int)xsrc[i] + yin0 * p.W) * 4u : kOobW);
        }
    };
    auto stage_store = [&](int buf) {
        float* __restrict__ dz_l = lz + buf * zbuf;
        float* __restrict__ dx_l = lx + buf * xbuf;
        unsigned oz = 0;
        asm volatile("" : "+v"(oz));
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            if (tid + 256 * i < 32 * p.zq) {
                float* d = dz_l + ((zdst[i] + oz) & 0xFFFFu);
                *reinterpret_cast<float2*>(d) = make_float2(vz[i].x, vz[i].y);  // pitch is even: 8-B aligned
                *reinterpret_cast<float2*>(d + 2) = make_float2(vz[i].z, vz[i].w);
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            if (tid + 256 * i < 32 * xper) {
                float* d = dx_l + ((xdst[i] + oz) & 0xFFFFu);
                d[0] = vx[i].x; d[1] = vx[i].y; d[2] = vx[i].z; d[3] = vx[i].w;
            }
        }
    };

    int tile = blockIdx.y;
    if (tile < p.n_tiles) stage_load(tile);
    __syncthreads();  // zero fill done
    if (tile < p.n_tiles) stage_store(0);
    __syncthreads();
    int buf = 0;
    for (; tile < p.n_tiles; tile += p.splits, buf ^= 1) {
        const int next = tile + p.splits;
        if (next < p.n_tiles) stage_load(next);
        int n, y0, npix, yin0;
        tile_geom(tile, n, y0, npix, yin0);
        const int npix4 = (npix + 3) & ~3;
        const float* __restrict__ az = lz + buf * zbuf + (co_sub * 16 + lr) * p.zpitch + lq;
        const float* __restrict__ bx = lx + buf * xbuf + (ci_sub * 16 + lr) * p.xplane + lq * S;
        // Wo % 4 == 0: the 4 pixels of a k-step share one output row, so the k-steps are walked row by row (no division in the
        // loop) and the T + 1 operand fragments of k-step i + 1 are fetched while the T MFMAs of k-step i run
        const int ksteps = npix4 >> 2, kpr = p.Wo >> 2;  // k-steps of the tile / per output row
        float a_cur = az[0], b_cur[T];
#pragma unroll
        for (int t = 0; t < T; ++t) b_cur[t] = bx[(t / KS) * p.Wp + (t % KS)];
        int col = 0, zoff = 0, xoff = 0;  // k-step inside its row; offsets of the CURRENT k-step in the dz / x tiles
        for (int i = 0; i < ksteps; ++i) {
            // offsets of the next k-step (the last one re-reads the current: discarded)
            int zn = zoff, xn = xoff, cn = col;
            if (i + 1 < ksteps) {
                zn = zoff + 4;
                if (++cn == kpr) { cn = 0; xn = xoff - (kpr - 1) * 4 * S + S * p.Wp; }
                else xn = xoff + 4 * S;
            }
            const float a_nxt = az[zn];
            float b_nxt[T];
#pragma unroll
            for (int t = 0; t < T; ++t) b_nxt[t] = bx[xn + (t / KS) * p.Wp + (t % KS)];
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
            a_cur = a_nxt;
#pragma unroll
            for (int t = 0; t < T; ++t) b_cur[t] = b_nxt[t];
            zoff = zn; xoff = xn; col = cn;
        }
        if (next < p.n_tiles) stage_store(buf ^ 1);
        __syncthreads();
    }
    float* __restrict__ slab = p.slabs + (size_t)blockIdx.y * p.Cout * p.Cin * T;
    const int ci = ci0 + ci_sub * 16 + lr;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + co_sub * 16 + lq * 4 + r;
            if (co < p.Cout && ci < p.Cin) slab[((size_t)co * p.Cin + ci) * T + t] = acc[t][r];
        }
}


template <int KS, int S>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int T = KS * KS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ lz = smem;                       // [32][zpitch]
    float* __restrict__ lx = smem + 32 * p.zpitch;       // [32][xplane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_sub = wave & 1, ci_sub = wave >> 1;
    const int lq = lane >> 4, lr = lane & 15;
    const int cot = blockIdx.x / p.ci_tiles, cit = blockIdx.x % p.ci_tiles;
    const int co0 = cot * 32, ci0 = cit * 32;
    const int HW = p.H * p.W, HoWo = p.Ho * p.Wo;

    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.y; tile < p.n_tiles; tile += p.splits) {
        const int n = tile / p.tiles_y, y0 = (tile % p.tiles_y) * p.R;
        const int rows = min(p.R, p.Ho - y0);
        const int npix = rows * p.Wo;
        const int yin0 = y0 * S - p.pad;
        __syncthreads();  // previous tile consumed
        // dz tile: 32 couts x npix (rows are contiguous in NCHW), zero beyond Cout / npix (rounded up to 4)
        const int npix4 = (npix + 3) & ~3;
        for (int e = tid; e < 32 * npix4; e += 256) {
            const int c = e / npix4, px = e - c * npix4;
            float v = 0.f;
            if (co0 + c < p.Cout && px < npix) v = p.dz[((size_t)n * p.Cout + co0 + c) * HoWo + (size_t)y0 * p.Wo + px];
            lz[c * p.zpitch + px] = v;
        }
        // x tile with halo: 32 cins x Rin x Wp, zero outside the image / beyond Cin
        const int per_c = p.Rin * p.Wp;
        for (int e = tid; e < 32 * per_c; e += 256) {
            const int c = e / per_c, rem = e - c * per_c;
            const int r = rem / p.Wp, xx = rem - r * p.Wp;
            const int yin = yin0 + r, xin = xx - p.pad;
            float v = 0.f;
            if (ci0 + c < p.Cin && yin >= 0 && yin < p.H && xin >= 0 && xin < p.W)
                v = p.x[((size_t)n * p.Cin + ci0 + c) * HW + (size_t)yin * p.W + xin];
            lx[c * p.xplane + rem] = v;
        }
        __syncthreads();
        const float* __restrict__ az = lz + (co_sub * 16 + lr) * p.zpitch;
        const float* __restrict__ bx = lx + (ci_sub * 16 + lr) * p.xplane;
        for (int q = 0; q < npix4; q += 4) {
            const int px = q + lq;                 // this lane's pixel of the k-step (zero-padded beyond npix)
            const int pc = px < npix ? px : 0;     // clamp the address; its dz value is 0 anyway
            const int y = pc / p.Wo, xo = pc - y * p.Wo;
            const float a = az[px];
            const int boff = (y * S) * p.Wp + xo * S;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float b = bx[boff + (t / KS) * p.Wp + (t % KS)];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // D layout: col = lane & 15 -> cin, row = (lane >> 4) * 4 + r -> cout
    float* __restrict__ slab = p.slabs + (size_t)blockIdx.y * p.Cout * p.Cin * T;
    const int ci = ci0 + ci_sub * 16 + lr;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + co_sub * 16 + lq * 4 + r;
            if (co < p.Cout && ci < p.Cin) slab[((size_t)co * p.Cin + ci) * T + t] = acc[t][r];
        }
}

// Sum the slabs in a FIXED order (deterministic): eight interleaved partial sums (slab k goes to partial k & 7) keep
// eight loads in flight per thread, then the partials are combined left to right.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, size_t count,
                                                           int splits, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= splits; k += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) part[j] += slabs[(size_t)(k + j) * count + i];
        }
        for (int j = 0; k < splits; ++k, ++j) part[j] += slabs[(size_t)k * count + i];
        const float s = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
        dw[i] = accumulate ? dw[i] + s : s;
    }
}

// Small weights (the 32- / 64-channel layers): G threads per element share the walk over the slabs - see the fp16 twin in
// wgrad_f16.hip; fixed-order combine through LDS, deterministic.
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_grouped_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                   size_t count, int splits, int accumulate) {
    constexpr int E = 256 / G;
    __shared__ float sm[G][E];
    const int e = threadIdx.x % E, g = threadIdx.x / E;
    const size_t i = (size_t)blockIdx.x * E + e;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < count) {
        int k = g, j = 0;
        for (; k + 3 * G < splits; k += 4 * G) {
#pragma unroll
            for (int u = 0; u < 4; ++u) part[u] += slabs[(size_t)(k + u * G) * count + i];
        }
        for (; k < splits; k += G, ++j) part[j] += slabs[(size_t)k * count + i];
    }
    sm[g][e] = (part[0] + part[1]) + (part[2] + part[3]);
    __syncthreads();
    if (g == 0 && i < count) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < G; ++q) v += sm[q][e];
        dw[i] = accumulate ? dw[i] + v : v;
    }
}

static int odd_up(int v) { return v | 1; }

static int wgrad_geometry(const mp_conv_desc* d, WgradParams& p, size_t& lds_bytes) {
    if (!d) return MP_ERR_NULL;
    if (d->n <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0 || d->conv_h <= 0 || d->conv_w <= 0) return MP_ERR_SHAPE;
    // 1x1 / 3x3 with padding k/2, or the 4x4 stride-2 padding-1 form (the transposed convolution's weight gradient, roles of
    // input and output gradient exchanged by the caller; simple kernel only)
    if (d->kh != d->kw || !(d->kh == 1 || d->kh == 3 || d->kh == 4)) return MP_ERR_UNSUPPORTED;
    if (!(d->stride == 1 || d->stride == 2)) return MP_ERR_UNSUPPORTED;
    if (d->pad_top != d->pad_left) return MP_ERR_UNSUPPORTED;
    if (d->kh == 4 ? (d->stride != 2 || d->pad_top != 1) : (d->pad_top != d->kh / 2)) return MP_ERR_UNSUPPORTED;
    p.N = d->n; p.Cin = d->cin; p.H = d->h; p.W = d->w; p.Cout = d->cout; p.Ho = d->conv_h; p.Wo = d->conv_w; p.pad = d->pad_top;
    const int S = d->stride, KS = d->kh;
    int R = 192 / p.Wo;
    if (R < 1) R = 1;
    if (R > p.Ho) R = p.Ho;
    p.vec = ((p.W & 3) == 0 && (p.Wo & 3) == 0 && d->kh != 4 && !knob("MP_WGRAD_SIMPLE")) ? 1 : 0;
    if (p.vec) {
        // pipelined kernel: double-buffered tiles; R limited by the per-thread staging registers (NZ = 6, NX = 9
        // 16-B units) and by 2 workgroups per CU (<= 78 KiB LDS)
        // (wide stride-2 layers - the stem's 64->64 conv on 128x96 maps - need 99 KB even for one output row: they run one
        // workgroup per CU rather than fall back to the scalar-staging kernel, which is 6x slower there)
        const int r_first = R;
        for (int budget_kb = 78;; --R) {
            p.R = R;
            p.Rin = (R - 1) * S + KS;
            p.Wp = (p.Wo - 1) * S + KS;
            if (p.Wp < p.pad + p.W) p.Wp = p.pad + p.W;  // whole input rows are staged
            p.zq = R * p.Wo / 4;
            p.xw4 = p.W / 4;
            p.xplane = ((p.Rin * p.Wp + 31) / 32) * 32 + 2;
            p.zpitch = ((R * p.Wo + 31) / 32) * 32 + 2;
            lds_bytes = (size_t)2 * 32 * (p.xplane + p.zpitch) * 4;
            const bool fits = 32 * p.zq <= 6 * 256 && 32 * p.Rin * p.xw4 <= 9 * 256 && lds_bytes <= (size_t)budget_kb * 1024 &&
                              p.zpitch < 65536 && 32 * p.xplane < 65536;
            if (fits) break;
            if (R == 1) {
                if (budget_kb == 78) { budget_kb = 150; R = r_first + 1; continue; }
                p.vec = 0;
                break;
            }
        }
    }
    if (!p.vec) {
        R = 192 / p.Wo;
        if (R < 1) R = 1;
        if (R > p.Ho) R = p.Ho;
        for (;;) {
            p.R = R;
            p.Rin = (R - 1) * S + KS;
            p.Wp = (p.Wo - 1) * S + KS;
            p.xplane = odd_up(p.Rin * p.Wp);
            p.zpitch = odd_up(((R * p.Wo + 3) & ~3) + 1);
            lds_bytes = (size_t)32 * (p.xplane + p.zpitch) * 4;
            if (lds_bytes <= 150 * 1024 || R == 1) break;
            R = (R + 1) / 2;
        }
        if (lds_bytes > 150 * 1024) return MP_ERR_UNSUPPORTED;
    }
    p.tiles_y = (p.Ho + p.R - 1) / p.R;
    p.n_tiles = p.N * p.tiles_y;
    p.ci_tiles = (p.Cin + 31) / 32;
    const int out_tiles = ((p.Cout + 31) / 32) * p.ci_tiles;
    int splits = 1024 / out_tiles;
    if (splits < 1) splits = 1;
    if (splits > p.n_tiles) splits = p.n_tiles;
    if (splits > 256) splits = 256;  // (512 / 1024 measured: no gain, the slab reduction grows)
    p.splits = splits;
    return MP_OK;
}

}  // namespace mp

using namespace mp;

extern "C" {

size_t mp_conv_wgrad_workspace_bytes(const mp_conv_desc* desc) {
    WgradParams p{};
    size_t lds = 0;
    if (wgrad_geometry(desc, p, lds) != MP_OK) return 0;
    return (size_t)p.splits * p.Cout * p.Cin * desc->kh * desc->kw * sizeof(float);
}

int mp_conv_wgrad(const mp_conv_desc* desc, const float* x, const float* dz, float* dw, int accumulate, void* workspace,
                  size_t workspace_bytes, mp_stream_t stream) {
    if (!x || !dz || !dw) return MP_ERR_NULL;
    WgradParams p{};
    size_t lds = 0;
    int rc = wgrad_geometry(desc, p, lds);
    if (rc != MP_OK) return rc;
    const size_t count = (size_t)p.Cout * p.Cin * desc->kh * desc->kw;
    if (!workspace || workspace_bytes < (size_t)p.splits * count * sizeof(float)) return MP_ERR_WORKSPACE;
    p.x = x; p.dz = dz; p.slabs = reinterpret_cast<float*>(workspace);
    hipStream_t s = as_stream(stream);
    dim3 grid(((p.Cout + 31) / 32) * p.ci_tiles, p.splits), block(256);
#define MP_WGRAD_LAUNCH(KS_, S_)                                                                                        \
    do {                                                                                                                \
        auto kern = conv_wgrad_kernel<KS_, S_>;                                                                         \
        static AttrOnce attr_once;                                                                                       \
        if (attr_once.need()) {                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            (void)hipGetLastError();                                                                                    \
        }                                                                                                               \
        hipLaunchKernelGGL(kern, grid, block, lds, s, p);                                                               \
    } while (0)
#define MP_WGRAD_LAUNCH_PIPE(KS_, S_)                                                                                   \
    do {                                                                                                                \
        auto kern = conv_wgrad_pipe_kernel<KS_, S_, 6, 9>;                                                              \
        static AttrOnce attr_once;                                                                                       \
        if (attr_once.need()) {                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            (void)hipGetLastError();                                                                                    \
        }                                                                                                               \
        hipLaunchKernelGGL(kern, grid, block, lds, s, p);                                                               \
    } while (0)
    if (p.vec) {
        if (desc->kh == 3 && desc->stride == 1) MP_WGRAD_LAUNCH_PIPE(3, 1);
        else if (desc->kh == 3 && desc->stride == 2) MP_WGRAD_LAUNCH_PIPE(3, 2);
        else if (desc->kh == 1 && desc->stride == 1) MP_WGRAD_LAUNCH_PIPE(1, 1);
        else MP_WGRAD_LAUNCH_PIPE(1, 2);
    } else {
        if (desc->kh == 4) MP_WGRAD_LAUNCH(4, 2);
        else if (desc->kh == 3 && desc->stride == 1) MP_WGRAD_LAUNCH(3, 1);
        else if (desc->kh == 3 && desc->stride == 2) MP_WGRAD_LAUNCH(3, 2);
        else if (desc->kh == 1 && desc->stride == 1) MP_WGRAD_LAUNCH(1, 1);
        else MP_WGRAD_LAUNCH(1, 2);
    }
#undef MP_WGRAD_LAUNCH_PIPE
#undef MP_WGRAD_LAUNCH
    rc = check_launch();
    if (rc != MP_OK) return rc;
    if (count <= 18432 && p.splits >= 64) {
        hipLaunchKernelGGL(wgrad_reduce_grouped_kernel<16>, dim3((unsigned)((count + 15) / 16)), dim3(256), 0, s, p.slabs, dw, count,
                           p.splits, accumulate ? 1 : 0);
    } else if (count <= 73728 && p.splits >= 16) {
        hipLaunchKernelGGL(wgrad_reduce_grouped_kernel<4>, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, s, p.slabs, dw, count,
                           p.splits, accumulate ? 1 : 0);
    } else {
        size_t blocks = (count + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p.slabs, dw, count, p.splits, accumulate ? 1 : 0);
    }
    return check_launch();
}

}  // extern "C"
