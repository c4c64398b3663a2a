// Streaming 1x1 (pointwise) convolution on the fp32 matrix cores for the HBM-bound layers of stage 1 / ResNet layer1
// (hrnet.py:86-146 Bottleneck conv1 / conv3, resnet.py Bottleneck): K = Cin <= 256, Cin * Cout <= 16384, every image plane a
// multiple of 64 pixels.  These layers move 0.5 - 0.9 GB per launch for 13 GFLOP: what matters is that every byte crosses the
// CU once and that enough of them are in flight, not the MFMA rate.
//
//   workgroup   256 threads, persistent: a run of consecutive 64-pixel tiles; ALL output channels of a tile (the input is read
//               from HBM once, not once per cout tile)
//   weights     in REGISTERS for the whole run: wave w owns cout blocks CBW*w .. CBW*w + CBW-1 (16 channels each) and keeps
//               their K x 16 weights as MFMA B fragments (K/4 * CBW <= 64 VGPRs) - loaded once from the direct kernel's packing
//   input       [64 cin][64 px] chunks, global -> registers -> LDS (pitch 80 floats: the four cin rows of an operand fetch on
//               disjoint banks), double-buffered: the chunk after the current one is in flight during the MFMAs
//   epilogue    scale / shift (+res1)(+ReLU), 16-byte loads / stores (a lane holds 4 consecutive pixels of one cout); the
//               residual of a tile is requested before its last chunk's MFMAs
#include <stdlib.h>

#include "conv_mfma.h"
#include "conv_pw.h"

namespace mp {

namespace {

constexpr int kPT = 64;      // pixels per tile
constexpr int kKC = 64;      // input channels per LDS chunk
constexpr int kPitch = 80;   // floats per cin row in LDS

__device__ __forceinline__ void pw_barrier() {
    // LDS traffic of this wave done, then the workgroup barrier; global loads / stores stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// KQ = Cin / 4 (k-steps of the whole K), CBW = cout blocks per wave
template <int KQ, int CBW>
__global__ __launch_bounds__(256, 2) void conv1x1_f32_stream_kernel(const PwParams p) {
    constexpr int NCH = (KQ * 4 + kKC - 1) / kKC;   // chunks per tile
    constexpr int QC = KQ < kKC / 4 ? KQ : kKC / 4; // k-steps per chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const int HW = p.HW;

    // ---- this wave's weights: B fragments (k = lq, cout = lr) of CBW cout blocks, all k-steps
    float wreg[KQ][CBW];
    {
        const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.wp, (size_t)KQ * 4 * p.Cout_pad16 * 4);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
            const int co = (wave * CBW + cb) * 16 + lr;
#pragma unroll
            for (int q = 0; q < KQ; ++q)
                wreg[q][cb] = buf_load1(rs_w, co < p.Cout_pad16 ? (unsigned)((q * 4 + lq) * p.Cout_pad16 + co) * 4u : kOob);
        }
    }
    float sc[CBW], sh[CBW];
    unsigned co_off[CBW];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) {
        const int co = (wave * CBW + cb) * 16 + lr;
        const int cc = co < p.Cout ? co : 0;
        sc[cb] = p.scale[cc];
        sh[cb] = p.shift[cc];
        co_off[cb] = co < p.Cout ? (unsigned)co * HW * 4u : kOob;
    }

    // ---- tiles of this workgroup: [t0, t1) of N * HW / 64, consecutive (one image's plane is a run of whole tiles)
    const int t0 = blockIdx.x * p.tiles_per_wg, t1 = min(t0 + p.tiles_per_wg, p.tiles);
    if (t0 >= t1) return;  // workgroup-uniform
    const int tpi = HW / kPT;  // tiles per image
    const size_t x_bytes = (size_t)p.N * p.Cin * HW * 4, o_bytes = (size_t)p.N * p.Cout * HW * 4;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, x_bytes);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, o_bytes);
    const __amdgpu_buffer_rsrc_t rs_r1 = make_rsrc(p.res1 ? p.res1 : p.out, p.res1 ? o_bytes : 0);

    // staging: a chunk = 64 cin rows x 64 px = 1024 float4 units, 4 per thread: unit u -> (cin row u / 16, float4 u % 16)
    unsigned s_src[4];
    int s_dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = tid + 256 * i, row = u >> 4, c4 = (u & 15) * 4;
        s_src[i] = (unsigned)(row * HW + c4) * 4u;
        s_dst[i] = row * kPitch + c4;
    }
    f32x4 vin[4];
    // stream position = (tile, chunk); base offset of a position in x
    auto pos_base = [&](int tile, int ch) {
        const int n = tile / tpi, pt = tile - n * tpi;
        return (unsigned)(((size_t)n * p.Cin + ch * kKC) * HW + pt * kPT) * 4u;
    };
    auto stage_load = [&](int tile, int ch) {
        const unsigned base = tile < t1 ? pos_base(tile, ch) : kOob;  // past the run: out of range, zeros (never used)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool row_ok = NCH > 1 || (tid + 256 * i) < KQ * 4 * 16;  // Cin < 64: fewer rows
            vin[i] = buf_load4(rs_x, (base == kOob || !row_ok) ? kOob : base + s_src[i]);
        }
    };
    auto stage_store = [&](int buf) {
        float* __restrict__ d = smem + buf * (kKC * kPitch);
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(d + s_dst[i]) = vin[i];
    };

    stage_load(t0, 0);
    stage_store(0);
    pw_barrier();

    const int a_off = lq * kPitch + lr;
    int pos = 0;
    for (int tile = t0; tile < t1; ++tile) {
        f32x4 acc[4][CBW];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) acc[mb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int n = tile / tpi, pt = tile - n * tpi;
        const unsigned o_base = (unsigned)((size_t)n * p.Cout * HW + pt * kPT) * 4u;
        f32x4 r1[4][CBW];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch, ++pos) {
            // the next stream position's input chunk flies in while this one runs on the matrix cores
            if (ch + 1 < NCH) stage_load(tile, ch + 1);
            else stage_load(tile + 1, 0);
            if (ch == NCH - 1) {  // residuals of this tile: requested before its last chunk's MFMAs
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int cb = 0; cb < CBW; ++cb) {
                        const unsigned o = co_off[cb] == kOob ? kOob : o_base + co_off[cb] + (unsigned)(mb * 16 + lq * 4) * 4u;
                        r1[mb][cb] = buf_load4(rs_r1, o);
                    }
            }
            const float* __restrict__ xs = smem + (pos & 1) * (kKC * kPitch) + a_off;
#pragma unroll
            for (int q = 0; q < QC; ++q) {
                float a[4];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) a[mb] = xs[q * 4 * kPitch + mb * 16];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int cb = 0; cb < CBW; ++cb)
                        acc[mb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], wreg[ch * (kKC / 4) + q][cb], acc[mb][cb], 0, 0, 0);
            }
            stage_store((pos + 1) & 1);  // the other buffer: its last readers finished before the previous barrier
            pw_barrier();
        }
        // ---- epilogue of the tile
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
                f32x4 v = acc[mb][cb] * sc[cb] + sh[cb] + r1[mb][cb];
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                buf_store4(rs_o, co_off[cb] == kOob ? kOob : o_base + co_off[cb] + (unsigned)(mb * 16 + lq * 4) * 4u, v);
            }
    }
}

}  // namespace

int pw_configure(const mp_conv_desc* d, PwLaunch& L) {
    if (!d) return MP_ERR_NULL;
    if (d->n <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return MP_ERR_SHAPE;
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad_top != 0 || d->pad_left != 0) return MP_ERR_UNSUPPORTED;
    if (d->conv_h != d->h || d->conv_w != d->w || d->out_h != d->h || d->out_w != d->w) return MP_ERR_UNSUPPORTED;
    if (d->out_mul != 1 || d->out_rep != 1 || d->out_off_y != 0 || d->out_off_x != 0) return MP_ERR_UNSUPPORTED;
    const int hw = d->h * d->w;
    if (hw % kPT) return MP_ERR_UNSUPPORTED;
    if ((long long)d->n * d->cin * hw * 4 >= 0x7FFFFFF0LL || (long long)d->n * d->cout * hw * 4 >= 0x7FFFFFF0LL) return MP_ERR_UNSUPPORTED;
    // built shapes: (Cin / 4, cout blocks per wave)
    if (d->cin == 64 && d->cout > 64 && d->cout <= 256) { L.kq = 16; L.cbw = 4; }
    else if (d->cin == 64 && d->cout <= 64) { L.kq = 16; L.cbw = 1; }
    else if (d->cin == 256 && d->cout <= 64) { L.kq = 64; L.cbw = 1; }
    else if (d->cin == 128 && d->cout <= 128) { L.kq = 32; L.cbw = 2; }
    else return MP_ERR_UNSUPPORTED;
    PwParams& p = L.p;
    p.N = d->n; p.Cin = d->cin; p.Cout = d->cout; p.Cout_pad16 = (d->cout + 15) / 16 * 16; p.HW = hw;
    p.tiles = d->n * (hw / kPT);
    int wgs = 512;  // two persistent workgroups per CU
    if (const char* e = knob("MP_PW_WGS")) {
        const int v = atoi(e);
        if (v >= 1) wgs = v;
    }
    p.tiles_per_wg = (p.tiles + wgs - 1) / wgs;
    if (p.tiles_per_wg < 1) p.tiles_per_wg = 1;
    p.relu = d->relu;
    L.grid = (p.tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
    L.lds_bytes = (size_t)2 * kKC * kPitch * 4;
    return MP_OK;
}

int pw_launch(const PwLaunch& L, hipStream_t s) {
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(L.grid), dim3(256), L.lds_bytes, s, L.p);
        return check_launch();
    };
    if (L.kq == 16 && L.cbw == 4) return go(conv1x1_f32_stream_kernel<16, 4>);
    if (L.kq == 16 && L.cbw == 1) return go(conv1x1_f32_stream_kernel<16, 1>);
    if (L.kq == 64 && L.cbw == 1) return go(conv1x1_f32_stream_kernel<64, 1>);
    if (L.kq == 32 && L.cbw == 2) return go(conv1x1_f32_stream_kernel<32, 2>);
    return MP_ERR_UNSUPPORTED;
}

}  // namespace mp
