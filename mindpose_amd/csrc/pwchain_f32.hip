// fp32 expand + reduce 1x1 chain of HRNet's stage 1 (hrnet.py:86-146 Bottleneck, 440-470 layer1): the expand conv of Bottleneck i,
//   y = relu(conv1x1(mid; w3) * scale3 + shift3 + res)        64 -> 256 channels, res = the block's identity,
// and the reduce conv of Bottleneck i + 1 on it,
//   z = relu(conv1x1(y; w1) * scale1 + shift1)                256 -> 64 channels,
// in ONE persistent, weight-stationary launch - round 5.  Two more forms of the same kernel: DS - the first Bottleneck's residual is
// its down-sample conv (hrnet.py:74-81: conv1x1(x0; wd) * scale_d + shift_d, 64 -> 256): computed in the launch from the block's
// input instead of being written (403 MB at N = 128) and read back (403 MB) by a launch of its own; and the expand conv alone (the
// last Bottleneck: no reduce conv follows).
//
// Why: as two launches the pair is 199 + 140 us at N = 128 (64x48 maps): the expand conv moves 830 MB at 4.1 TB/s with the matrix
// pipe 37 % busy - bound by neither, a 128 x 128 tile with four k chunks is mostly skeleton (barriers behind fresh round trips, a
// store tail that holds the workgroup's slot; tools/variant_builds.sh) - and the reduce conv reads y (403 MB) back.  Here:
//   * one workgroup per CU for the whole launch, walking 64-pixel tiles (pixels of one image plane; stride = the grid);
//   * ALL weight matrices live in registers for the kernel's life as MFMA A operands (v_mfma_f32_32x32x2_f32: lane = (row l % 32,
//     k l / 32)): wave w owns the expand couts 64 w .. 64 w + 63 (2 row tiles x 32 k-steps = 64 registers; the same again for the
//     down-sample weights) and, for the reduce conv, the K SLICE of exactly those 64 channels (2 x 2 x 16 = 64 registers): its own
//     expand output - scaled, shifted, residual added, ReLU'd in its accumulators - IS its reduce B operand, straight from
//     registers.  An accumulator r of a 32 x 32 tile holds row 8 (r / 4) + 4 (l / 32) + r % 4: k-step s of the reduce conv takes
//     register s of the tile, i.e. the channel pair (c, c + 4) - the weight fragments are loaded in that k order, the sum over k is
//     the same sum in another order;
//   * y is stored once (full 128-byte lines: 32 lanes = 32 consecutive pixels of a cout) and never read back; the residual tile is
//     requested before the expand MFMAs and lands under them; the next tile's input (64 x 64 floats; DS: 128 x 64) is requested in
//     front of the reduce MFMAs and goes to the other LDS buffer behind them;
//   * the four K-slice partials of z (64 couts x 64 pixels each) are folded through LDS: every wave writes the three 32 x 32
//     quadrants it does not own, wave w adds quadrant w in wave order (bit-reproducible), scale / shift / ReLU, stores z.  LDS: 2 x
//     24 KB input tiles (DS: 2 x 48 KB; row pitch 96 floats = 32 mod 64: the two k rows of a fetch on disjoint bank halves), 48 KB
//     exchange, 4.5 KB of scale / shift tables.
// Per tile and wave 128 + 128 MFMAs = 16 384 matrix-pipe cycles = 6.8 us: 24 tiles per CU at N = 128 = 164 us if nothing else showed
// (the fp32 matrix peak: 25.8 GFLOP), against 1006 MB of HBM traffic = 188 us (measured with the MFMAs taken out:
// tools/variant_builds.sh).  Measured 280 - 296 us: one wave per SIMD (the weights fill the register file) serialises the
// epilogue, the hand-over and the memory waits with the MFMAs; requesting the residual a tile ahead and batching the LDS operand
// reads moved nothing (292 - 302 us).  The two launches it replaces: 339 us.  A second resident wave per SIMD is what helped: four
// waves on 32-pixel tiles with TWO workgroups per CU (272 us; eight waves in one workgroup: 285 - its barriers keep them in step).
// Results: y as the stand-alone expand conv up to fp32 rounding; z differs from the stand-alone reduce conv by the association of the
// k sum (four slices of 64, pairs (c, c + 4)) - tests/test_gpu_conv.py::test_expand_reduce_chain_f32 compares both with fp64 torch
// at the direct kernel's bar.
#include "pwchain_f32.h"

#include "conv_mfma.h"

namespace mp {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef PWC_ABLATE
#define PWC_ABLATE 0  // diagnostic builds (tools/variant_builds.sh): 1 no y stores, 2 no MFMA, 4 no residual loads; results wrong, timings meaningful
#endif

__device__ __forceinline__ f32x16 pwc_fake_mfma(float a, float b, f32x16 c) { c[0] += a * b; return c; }
#if PWC_ABLATE & 2
#define PWC_MFMA(a, b, c) pwc_fake_mfma(a, b, c)
#else
#define PWC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#endif

constexpr int kXP = 96;                  // floats per k row of a staged input tile
constexpr int kExWave = 3 * 16 * 64;     // floats one wave hands over per tile: three quadrants x 16 registers x 64 lanes
constexpr int kExBuf = 4 * kExWave;      // the exchange buffer
constexpr int kSs = 5 * 256 + 2 * 64;    // scale3 | shift3 | scale_d | shift_d | (spare) ... | scale1 | shift1
constexpr int kSsD = 512, kSs1 = 1024;   // offsets of the down-sample and reduce tables

__device__ __forceinline__ void chain_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS traffic done; global loads / stores stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// DS = the residual is the down-sample conv of x0 (computed here); RED = the reduce conv of the next block follows
template <bool DS, bool RED>
__global__ __launch_bounds__(256, 1) void expand_reduce_f32_kernel(const PwChainF32Params p) {
    constexpr int KR = DS ? 128 : 64;        // staged k rows: mid | x0
    constexpr int kXBuf = KR * kXP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ lds_x = smem;                // [2][KR][kXP]
    float* __restrict__ lds_ex = smem + 2 * kXBuf;   // [4 waves][3 quadrants][16][64]
    float* __restrict__ lds_ss = lds_ex + kExBuf;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    lds_ss[tid] = p.scale3[tid];
    lds_ss[256 + tid] = p.shift3[tid];
    if (DS) {
        lds_ss[kSsD + tid] = p.scale_d[tid];
        lds_ss[kSsD + 256 + tid] = p.shift_d[tid];
    }
    if (RED && tid < 64) {
        lds_ss[kSs1 + tid] = p.scale1[tid];
        lds_ss[kSs1 + 64 + tid] = p.shift1[tid];
    }

    // ---- the stationary weights (A operands).  Packed 1x1 weights are k-major: w3[k][256], wd[k][256], w1[k][64]
    float a3[2][32], ad[DS ? 2 : 1][DS ? 32 : 1];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            a3[rt][s] = p.w3[(2 * s + h) * 256 + 64 * wave + 32 * rt + l31];
            if constexpr (DS) ad[rt][s] = p.wd[(2 * s + h) * 256 + 64 * wave + 32 * rt + l31];
        }
    float a1[RED ? 2 : 1][RED ? 2 : 1][RED ? 16 : 1];  // [reduce row tile][the wave's expand row tile][k-step = accumulator register of that tile]
    if constexpr (RED) {
#pragma unroll
        for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
            for (int rt1 = 0; rt1 < 2; ++rt1)
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    a1[rt2][rt1][s] = p.w1[(64 * wave + 32 * rt1 + 8 * (s >> 2) + 4 * h + (s & 3)) * 64 + 32 * rt2 + l31];
    }

    const unsigned plane = (unsigned)p.HW * 4u;  // bytes of one channel plane
    const __amdgpu_buffer_rsrc_t rs_mid = make_rsrc(p.mid, (size_t)p.N * 64 * plane);
    const __amdgpu_buffer_rsrc_t rs_x0 = make_rsrc(DS ? p.x0 : p.mid, (size_t)p.N * 64 * plane);
    const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(DS ? p.mid : p.res, (size_t)p.N * (DS ? 64 : 256) * plane);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y, (size_t)p.N * 256 * plane);
    const __amdgpu_buffer_rsrc_t rs_z = make_rsrc(RED ? p.z : p.y, (size_t)p.N * (RED ? 64 : 256) * plane);

    // input staging: thread -> channel k = tid / 4, sixteen pixels 16 (tid % 4) ..: four 16-byte loads, 256-byte runs per channel
    // (DS: the same of x0 into rows 64 ..)
    const int xk = tid >> 2, xq = tid & 3;
    auto x_offset = [&](int tile) {
        const int n = tile / p.tiles_per_img, p0 = (tile - n * p.tiles_per_img) * 64;
        return (unsigned)((n * 64 + xk) * p.HW + p0 + 16 * xq) * 4u;
    };
    f32x4 xv[DS ? 8 : 4];
    auto x_load = [&](unsigned off) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xv[j] = buf_load4(rs_mid, off + 16u * j);
            if constexpr (DS) xv[4 + j] = buf_load4(rs_x0, off + 16u * j);
        }
    };
    auto x_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f32x4*>(lds_x + buf * kXBuf + xk * kXP + 16 * xq + 4 * j) = xv[j];
            if constexpr (DS) *reinterpret_cast<f32x4*>(lds_x + buf * kXBuf + (64 + xk) * kXP + 16 * xq + 4 * j) = xv[4 + j];
        }
    };

    int tile = blockIdx.x;
    if (tile >= p.total_tiles) return;  // (the grid is at most the tile count)
    x_load(x_offset(tile));
    x_store(0);
    chain_barrier();  // also: the scale / shift tables

    const int b_off = h * kXP + l31;
    for (int it = 0; tile < p.total_tiles; ++it, tile += gridDim.x) {
        const int cur = it & 1;
        const int n = tile / p.tiles_per_img, p0 = (tile - n * p.tiles_per_img) * 64;
        const bool has_next = tile + (int)gridDim.x < p.total_tiles;  // workgroup-uniform

        // ---- request of this tile: the residual (lands under the expand MFMAs)
        // lane part of a y / residual address: pixel p0 + 32 pt + l31 of the cout rows 4 h .. (the uniform part - the cout - is the scalar offset)
        unsigned v_y[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) v_y[pt] = (unsigned)((n * 256 + 4 * h) * p.HW + p0 + 32 * pt + l31) * 4u;
        f32x16 rr[2][2];
        if constexpr (!DS) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned s_off = (unsigned)(64 * wave + 32 * rt + 8 * (r >> 2) + (r & 3)) * plane;
                        rr[rt][pt][r] = (PWC_ABLATE & 4) ? (float)(v_y[pt] + s_off)
                                                         : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, v_y[pt], s_off, 0));
                    }
        }
        if (!RED && has_next) x_load(x_offset(tile + gridDim.x));

        // DS: the expand pass runs per pixel half (32 pixels = one MFMA column block) with the down-sample GEMM woven into it - four
        // independent accumulators either way (two alternating ones left the matrix pipe waiting on its own results: 292 -> 335 us) -
        // so that the residual is 32 registers at a time next to the 192 of the three resident weight matrices.
        constexpr int PH = DS ? 1 : 2;  // pixel halves per expand pass
        const float* __restrict__ xs = lds_x + cur * kXBuf + b_off;
        f32x16 acc1[2][2];
        [[maybe_unused]] f32x16 acc2[RED ? 2 : 1][RED ? 2 : 1];
        if constexpr (RED) {
#pragma unroll
            for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[rt2][pt][r] = 0.f;
        }
#pragma unroll
        for (int pt0 = 0; pt0 < 2; pt0 += PH) {
            // (DS: the next input tile is requested between the halves and lands under the second half's MFMAs)
            if (DS && RED && pt0 == 1 && has_next) x_load(x_offset(tile + gridDim.x));
            f32x16 r2[2][PH];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ph = 0; ph < PH; ++ph)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc1[rt][pt0 + ph][r] = 0.f;
                        if constexpr (DS) r2[rt][ph][r] = 0.f;
                    }
            // ---- expand: 64 couts of this wave x 32 PH pixels, K = 64 (DS: and the block's down-sample conv on the x0 rows of the tile)
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                float b[PH];
#pragma unroll
                for (int ph = 0; ph < PH; ++ph) b[ph] = xs[2 * s * kXP + 32 * (pt0 + ph)];
                [[maybe_unused]] float bd = 0.f;
                if constexpr (DS) bd = xs[(64 + 2 * s) * kXP + 32 * pt0];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                    for (int ph = 0; ph < PH; ++ph) acc1[rt][pt0 + ph] = PWC_MFMA(a3[rt][s], b[ph], acc1[rt][pt0 + ph]);
                    if constexpr (DS) r2[rt][0] = PWC_MFMA(ad[rt][s], bd, r2[rt][0]);
                }
            }
            if constexpr (DS) {  // the residual = the down-sample conv's output (no ReLU)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int row = 64 * wave + 32 * rt + 8 * g + 4 * h;
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(lds_ss + kSsD + row), sh = *reinterpret_cast<const f32x4*>(lds_ss + kSsD + 256 + row);
#pragma unroll
                        for (int e = 0; e < 4; ++e) r2[rt][0][4 * g + e] = r2[rt][0][4 * g + e] * sc[e] + sh[e];
                    }
            } else {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ph = 0; ph < PH; ++ph) r2[rt][ph] = rr[rt][pt0 + ph];
            }
            // ---- y = relu(acc * scale3 + shift3 + res): stored, and kept in the accumulators as the reduce conv's B operand
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = 64 * wave + 32 * rt + 8 * g + 4 * h;  // four consecutive couts
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(lds_ss + row), sh = *reinterpret_cast<const f32x4*>(lds_ss + 256 + row);
#pragma unroll
                    for (int ph = 0; ph < PH; ++ph)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = fmaxf(acc1[rt][pt0 + ph][4 * g + e] * sc[e] + sh[e] + r2[rt][ph][4 * g + e], 0.f);
                            acc1[rt][pt0 + ph][4 * g + e] = v;
                            if (RED && (PWC_ABLATE & 1)) continue;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, v_y[pt0 + ph],
                                                                  (unsigned)(64 * wave + 32 * rt + 8 * g + e) * plane, 0);
                        }
                }
            if constexpr (DS && RED) {
                // ---- reduce of THIS half (its y half dies here: the register budget of the down-sample form; two accumulators)
#pragma unroll
                for (int rt1 = 0; rt1 < 2; ++rt1)
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        acc2[0][pt0] = PWC_MFMA(a1[0][rt1][s], acc1[rt1][pt0][s], acc2[0][pt0]);
                        acc2[1][pt0] = PWC_MFMA(a1[1][rt1][s], acc1[rt1][pt0][s], acc2[1][pt0]);
                    }
            }
        }
        if constexpr (RED && !DS) {
            // ---- the next input tile is requested here (its 16 registers were the residual's until now) and lands under the
            //      reduce MFMAs;  reduce: this wave's K slice (its own 64 channels of y) for all 64 couts x 64 pixels
            if (has_next) x_load(x_offset(tile + gridDim.x));
#pragma unroll
            for (int rt1 = 0; rt1 < 2; ++rt1)
#pragma unroll
                for (int s = 0; s < 16; ++s)
#pragma unroll
                    for (int rt2 = 0; rt2 < 2; ++rt2) {
                        acc2[rt2][0] = PWC_MFMA(a1[rt2][rt1][s], acc1[rt1][0][s], acc2[rt2][0]);
                        acc2[rt2][1] = PWC_MFMA(a1[rt2][rt1][s], acc1[rt1][1][s], acc2[rt2][1]);
                    }
        }
        if constexpr (!RED) {
            if (has_next) x_store(cur ^ 1);
            chain_barrier();  // the next tile's input is in place; every wave is done with this one's
            continue;
        } else {
            // ---- hand-over: the next input tile, and the three quadrants this wave does not own (quadrant q = 2 rt2 + pt belongs to
            //      wave q).  The exchange buffer is single: the barrier in front of the writes says every wave has read the last tile's
            if (has_next) x_store(cur ^ 1);
            f32x4* __restrict__ ex4 = reinterpret_cast<f32x4*>(lds_ex);
            if (it > 0) chain_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q == wave) continue;  // wave-uniform
                const int slot = q < wave ? q : q - 1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& a = acc2[q >> 1][q & 1];
                    ex4[((wave * 3 + slot) * 4 + g) * 64 + lane] = (f32x4){a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                }
            }
            f32x16 mine;
            if (wave == 0) mine = acc2[0][0];
            else if (wave == 1) mine = acc2[0][1];
            else if (wave == 2) mine = acc2[1][0];
            else mine = acc2[1][1];
            chain_barrier();
            // ---- z quadrant of this wave: the four K-slice partials in wave order
            f32x16 tot;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x16 val;
                if (u == wave) {
                    val = mine;
                } else {
                    const int slot = wave < u ? wave : wave - 1;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 t = ex4[((u * 3 + slot) * 4 + g) * 64 + lane];
                        val[4 * g] = t[0]; val[4 * g + 1] = t[1]; val[4 * g + 2] = t[2]; val[4 * g + 3] = t[3];
                    }
                }
                if (u == 0) tot = val;
                else tot = tot + val;
            }
            const int rt2 = wave >> 1, ptz = wave & 1;
            const unsigned v_z = (unsigned)((n * 64 + 4 * h) * p.HW + p0 + 32 * ptz + l31) * 4u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = 32 * rt2 + 8 * g + 4 * h;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(lds_ss + kSs1 + row), sh = *reinterpret_cast<const f32x4*>(lds_ss + kSs1 + 64 + row);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaxf(tot[4 * g + e] * sc[e] + sh[e], 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_z, v_z, (unsigned)(32 * rt2 + 8 * g + e) * plane, 0);
                }
            }
        }
    }
}


// The same chain with EIGHT waves (two per SIMD): wave = (pixel half pw, cout quarter cw).  A wave keeps the same 128 weight registers
// but half the accumulators (its 64 couts x 32 pixels), which is what fits 256 registers per wave - and a second resident wave is what
// the four-wave form lacks: its epilogue, hand-over and memory waits (288 us per launch against a 164 us MFMA-only and a 188 us
// memory-only time) now run under the other wave's MFMAs.  Two accumulators alternate per wave, four per SIMD.  The K-slice partials
// of z are folded in the same wave order: z and y are bit-identical to the four-wave form's.  (The down-sample form keeps four waves:
// its three weight matrices are 192 registers.)
// PW = 2: the eight-wave form above (64-pixel tiles, one workgroup per CU).  PW = 1: FOUR waves on 32-pixel tiles, TWO workgroups per
// CU: the same two waves per SIMD, but the two pixel halves are separate workgroups - the barriers of the K-slice hand-over hold the
// eight waves of one workgroup in step (all in the MFMA phases together, all in the epilogue together); two workgroups drift apart.
template <bool RED, int PW>
__global__ __launch_bounds__(256 * PW, 3 - PW) void expand_reduce_f32_w8_kernel(const PwChainF32Params p) {
    constexpr int TPX = 32 * PW;  // pixels per tile
    // row pitch of the staged tile: the two k rows of a fetch on disjoint bank halves - 96 floats for 64-pixel rows, 32 for 32-pixel rows
    constexpr int XP = PW == 2 ? kXP : 32;
    constexpr int kXBuf = 64 * XP;
    constexpr int EXB = PW == 2 ? 1 : 2;  // exchange buffers: the 32-pixel form has the LDS for two (no second barrier per tile)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ lds_x = smem;                // [2][64][kXP]
    float* __restrict__ lds_ex = smem + 2 * kXBuf;   // [4 PW waves][3 owners][2][64 lanes] float4
    float* __restrict__ lds_ss = lds_ex + kExBuf / 2 * PW * EXB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 3, pw = wave >> 2;
    const int tiles_per_img = p.tiles_per_img * (2 / PW), total_tiles = p.total_tiles * (2 / PW);
    const int l31 = lane & 31, h = lane >> 5;

    if (tid < 256) {
        lds_ss[tid] = p.scale3[tid];
        lds_ss[256 + tid] = p.shift3[tid];
    }
    if (RED && tid < 64) {
        lds_ss[kSs1 + tid] = p.scale1[tid];
        lds_ss[kSs1 + 64 + tid] = p.shift1[tid];
    }
    float a3[2][32];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int s = 0; s < 32; ++s) a3[rt][s] = p.w3[(2 * s + h) * 256 + 64 * cw + 32 * rt + l31];
    float a1[RED ? 2 : 1][RED ? 2 : 1][RED ? 16 : 1];
    if constexpr (RED) {
#pragma unroll
        for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
            for (int rt1 = 0; rt1 < 2; ++rt1)
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    a1[rt2][rt1][s] = p.w1[(64 * cw + 32 * rt1 + 8 * (s >> 2) + 4 * h + (s & 3)) * 64 + 32 * rt2 + l31];
    }
    const unsigned plane = (unsigned)p.HW * 4u;
    const __amdgpu_buffer_rsrc_t rs_mid = make_rsrc(p.mid, (size_t)p.N * 64 * plane);
    const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(p.res, (size_t)p.N * 256 * plane);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y, (size_t)p.N * 256 * plane);
    const __amdgpu_buffer_rsrc_t rs_z = make_rsrc(RED ? p.z : p.y, (size_t)p.N * (RED ? 64 : 256) * plane);

    // input staging: thread -> channel k = tid / (4 PW), eight pixels 8 (tid % (4 PW)) ..: two 16-byte loads
    const int xk = tid / (4 * PW), xq = tid % (4 * PW);
    auto x_offset = [&](int tile) {
        const int n = tile / tiles_per_img, p0 = (tile - n * tiles_per_img) * TPX;
        return (unsigned)((n * 64 + xk) * p.HW + p0 + 8 * xq) * 4u;
    };
    f32x4 xv[2];
    auto x_load = [&](unsigned off) __attribute__((always_inline)) {
        xv[0] = buf_load4(rs_mid, off);
        xv[1] = buf_load4(rs_mid, off + 16u);
    };
    auto x_store = [&](int buf) __attribute__((always_inline)) {
        *reinterpret_cast<f32x4*>(lds_x + buf * kXBuf + xk * XP + 8 * xq) = xv[0];
        *reinterpret_cast<f32x4*>(lds_x + buf * kXBuf + xk * XP + 8 * xq + 4) = xv[1];
    };

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    x_load(x_offset(tile));
    x_store(0);
    chain_barrier();

    const int b_off = h * XP + 32 * pw + l31;
    for (int it = 0; tile < total_tiles; ++it, tile += gridDim.x) {
        const int cur = it & 1;
        const int n = tile / tiles_per_img, p0 = (tile - n * tiles_per_img) * TPX;
        const bool has_next = tile + (int)gridDim.x < total_tiles;
        const unsigned v_y = (unsigned)((n * 256 + 4 * h) * p.HW + p0 + 32 * pw + l31) * 4u;
        f32x16 rr[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned s_off = (unsigned)(64 * cw + 32 * rt + 8 * (r >> 2) + (r & 3)) * plane;
                rr[rt][r] = (PWC_ABLATE & 4) ? (float)(v_y + s_off) : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, v_y, s_off, 0));
            }
        if (!RED && has_next) x_load(x_offset(tile + gridDim.x));
        // ---- expand: 64 couts x 32 pixels of this wave, K = 64
        const float* __restrict__ xs = lds_x + cur * kXBuf + b_off;
        f32x16 acc1[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[rt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float b = xs[2 * s * XP];
            acc1[0] = PWC_MFMA(a3[0][s], b, acc1[0]);
            acc1[1] = PWC_MFMA(a3[1][s], b, acc1[1]);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = 64 * cw + 32 * rt + 8 * g + 4 * h;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(lds_ss + row), sh = *reinterpret_cast<const f32x4*>(lds_ss + 256 + row);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaxf(acc1[rt][4 * g + e] * sc[e] + sh[e] + rr[rt][4 * g + e], 0.f);
                    acc1[rt][4 * g + e] = v;
                    if (RED && (PWC_ABLATE & 1)) continue;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, v_y, (unsigned)(64 * cw + 32 * rt + 8 * g + e) * plane, 0);
                }
            }
        if constexpr (!RED) {
            if (has_next) x_store(cur ^ 1);
            chain_barrier();
            continue;
        } else {
            if (has_next) x_load(x_offset(tile + gridDim.x));
            // ---- reduce: this wave's K slice (its own 64 channels of y) for all 64 couts x its 32 pixels
            f32x16 acc2[2];
#pragma unroll
            for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[rt2][r] = 0.f;
#pragma unroll
            for (int rt1 = 0; rt1 < 2; ++rt1)
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    acc2[0] = PWC_MFMA(a1[0][rt1][s], acc1[rt1][s], acc2[0]);
                    acc2[1] = PWC_MFMA(a1[1][rt1][s], acc1[rt1][s], acc2[1]);
                }
            // ---- hand-over inside the pixel half: owner o of (row tile o / 2, registers 8 (o % 2) .. + 7) is wave (pw, o)
            if (has_next) x_store(cur ^ 1);
            f32x4* __restrict__ ex4 = reinterpret_cast<f32x4*>(lds_ex) + (EXB == 2 ? cur * (kExBuf / 8) : 0);
            if (EXB == 1 && it > 0) chain_barrier();  // every wave has read the last tile's partials (two buffers: the barrier two tiles on says so)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o == cw) continue;  // wave-uniform
                const int slot = o < cw ? o : o - 1;
                const f32x16& a = acc2[o >> 1];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int r0 = 8 * (o & 1) + 4 * gg;
                    ex4[(((pw * 4 + cw) * 3 + slot) * 2 + gg) * 64 + lane] = (f32x4){a[r0], a[r0 + 1], a[r0 + 2], a[r0 + 3]};
                }
            }
            float mine[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                mine[i] = cw == 0 ? acc2[0][i] : cw == 1 ? acc2[0][8 + i] : cw == 2 ? acc2[1][i] : acc2[1][8 + i];
            chain_barrier();
            float tot[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float val[8];
                if (u == cw) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) val[i] = mine[i];
                } else {
                    const int slot = cw < u ? cw : cw - 1;
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const f32x4 t = ex4[(((pw * 4 + u) * 3 + slot) * 2 + gg) * 64 + lane];
                        val[4 * gg] = t[0]; val[4 * gg + 1] = t[1]; val[4 * gg + 2] = t[2]; val[4 * gg + 3] = t[3];
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) tot[i] = u == 0 ? val[i] : tot[i] + val[i];
            }
            const int rt2 = cw >> 1, hh = cw & 1;
            const unsigned v_z = (unsigned)((n * 64 + 4 * h) * p.HW + p0 + 32 * pw + l31) * 4u;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int g = 2 * hh + gg, row = 32 * rt2 + 8 * g + 4 * h;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(lds_ss + kSs1 + row), sh = *reinterpret_cast<const f32x4*>(lds_ss + kSs1 + 64 + row);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaxf(tot[4 * gg + e] * sc[e] + sh[e], 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_z, v_z, (unsigned)(32 * rt2 + 8 * g + e) * plane, 0);
                }
            }
        }
    }
}

}  // namespace

int pwchain32_build(const float* mid, const float* res, const float* x0, const float* packed_wd, const float* scale_d, const float* shift_d,
                    const float* packed_w3, const float* scale3, const float* shift3, const float* packed_w1, const float* scale1,
                    const float* shift1, float* y, float* z, int n, int cm, int ce, int cr, int h, int w, PwChainF32Launch& L) {
    if (!mid || !packed_w3 || !scale3 || !shift3 || !y) return MP_ERR_NULL;
    const bool ds = x0 != nullptr, red = packed_w1 != nullptr;
    if (ds ? (res || !packed_wd || !scale_d || !shift_d) : (!res || packed_wd || scale_d || shift_d)) return MP_ERR_NULL;  // exactly one residual source
    if (red ? (!scale1 || !shift1 || !z) : (scale1 || shift1 || z)) return MP_ERR_NULL;
    if (n <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    // built for the widths of HRNet's stage 1 (hrnet.py:440-470: Bottleneck 64 -> 256, expansion 4) and planes of whole 64-pixel tiles
    if (cm != 64 || ce != 256 || (red && cr != 64) || ((h * w) & 63)) return MP_ERR_UNSUPPORTED;
    if (ds && !red) return MP_ERR_UNSUPPORTED;  // (no caller: the first Bottleneck is followed by another)
    if ((long long)n * 256 * h * w * 4 >= 0x7FFFFFF0LL) return MP_ERR_UNSUPPORTED;  // 32-bit byte offsets
    PwChainF32Params& p = L.p;
    p.mid = mid; p.res = res; p.x0 = x0; p.wd = packed_wd; p.scale_d = scale_d; p.shift_d = shift_d;
    p.w3 = packed_w3; p.scale3 = scale3; p.shift3 = shift3; p.w1 = packed_w1; p.scale1 = scale1; p.shift1 = shift1;
    p.y = y; p.z = z; p.N = n; p.HW = h * w; p.tiles_per_img = p.HW / 64; p.total_tiles = n * p.tiles_per_img;
    L.ds = ds; L.red = red;
    // forms: 4 = four waves, 64-pixel tiles, one workgroup per CU (the down-sample form's only one); 8 = eight waves; 2 = four waves on
    // 32-pixel tiles, two workgroups per CU.  Measured at N = 128 (64x48): chain 300 / 285 / 272 us for forms 4 / 8 / 2, expand conv
    // alone 185 / 165 / 174 us
    L.form = ds ? 4 : red ? 2 : 8;
    if (const char* e = knob("MP_PWCHAIN32_WAVES")) {  // experiments / tests
        const int v = atoi(e);
        if (!ds && (v == 2 || v == 4 || v == 8)) L.form = v;
    }
    int cus = 256;
    if (const char* e = knob("MP_PWCHAIN32_WGS")) cus = atoi(e) > 0 ? atoi(e) : cus;  // experiments
    const int tiles = L.form == 2 ? 2 * p.total_tiles : p.total_tiles, slots = L.form == 2 ? 2 * cus : cus;
    L.grid = tiles < slots ? tiles : slots;
    L.lds_bytes = L.form == 4 ? (size_t)(2 * (ds ? 128 : 64) * kXP + kExBuf + kSs) * 4
                              : L.form == 8 ? (size_t)(2 * 64 * kXP + kExBuf + kSs) * 4 : (size_t)(2 * 64 * 32 + kExBuf + kSs) * 4;
    return MP_OK;
}

int pwchain32_launch(const PwChainF32Launch& L, hipStream_t s) {
    if (g_dry_launch) return MP_OK;
    auto go = [&](auto kern) {
        static AttrOnce attr_once;  // (one per instantiation of this lambda)
        if (attr_once.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(kern, dim3(L.grid), dim3(256), L.lds_bytes, s, L.p);
        return check_launch();
    };
    auto go8 = [&](auto kern, int threads) {
        static AttrOnce attr_once;
        if (attr_once.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(kern, dim3(L.grid), dim3(threads), L.lds_bytes, s, L.p);
        return check_launch();
    };
    if (L.ds) return go(expand_reduce_f32_kernel<true, true>);
    if (L.form == 8) return L.red ? go8(expand_reduce_f32_w8_kernel<true, 2>, 512) : go8(expand_reduce_f32_w8_kernel<false, 2>, 512);
    if (L.form == 2) return L.red ? go8(expand_reduce_f32_w8_kernel<true, 1>, 256) : go8(expand_reduce_f32_w8_kernel<false, 1>, 256);
    if (L.red) return go(expand_reduce_f32_kernel<false, true>);
    return go(expand_reduce_f32_kernel<false, false>);
}

}  // namespace mp

using namespace mp;

extern "C" int mp_expand_reduce_fwd(const float* mid, const float* res, const float* x0, const float* packed_wd, const float* scale_d,
                                    const float* shift_d, const float* packed_w3, const float* scale3, const float* shift3,
                                    const float* packed_w1, const float* scale1, const float* shift1, float* y, float* z, int n, int cm,
                                    int ce, int cr, int h, int w, mp_stream_t stream) {
    PwChainF32Launch L{};
    const int rc = pwchain32_build(mid, res, x0, packed_wd, scale_d, shift_d, packed_w3, scale3, shift3, packed_w1, scale1, shift1, y, z, n, cm,
                                   ce, cr, h, w, L);
    if (rc != MP_OK) return rc;
    return pwchain32_launch(L, as_stream(stream));
}
