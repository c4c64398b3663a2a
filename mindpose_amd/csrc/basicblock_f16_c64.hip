// Fused fp16 BasicBlock for the 64-channel branch (hrnet.py:30-83 BasicBlock, :202-241 branch 1 of stages 2 - 4:
// relu(bn2(conv2(relu(bn1(conv1 x)))) + x), both convs 3x3 stride 1) - round 4.
//
// Why: at N = 128 the 64 -> 64 @32x24 layer is 7.25 GFLOP = 2.9 us of matrix pipe, but every launch of the weights-in-registers
// kernel costs 10.8 us in the network - a 5 - 6 k-cycle prologue (DMA issue, first weight fragments, address tables), the epilogue
// and the launch boundary are most of it (round-4 stamps).  Fusing the block's two convs halves those fixed costs
// and keeps the intermediate tensor out of HBM.  The 32-channel block (basicblock_f16_v2.hip) is bound by its 96 KB tiles; here
// the tiles are small and the structure is different:
//   * one workgroup = one band of R output rows of one image, ALL 64 channels; two workgroups per CU (72 KB LDS, < 256 registers);
//   * the four waves split the work 2 x 2: a wave owns a PAIR of cout tiles (32 couts: one 16-byte channel-block element per lane
//     and pixel) of BOTH convs for half of the band's pixels.  Its weight fragments live in a ring of one k-step (9 taps x 2 tiles =
//     72 registers) that walks conv1 k-step 0 -> 1 -> conv2 k-step 0 -> 1: a tap's fragments are replaced by the next element's right
//     behind the MFMAs that consumed them - conv2's weights arrive under conv1;
//   * the input tile (rows y0 - 2 ... y0 + R + 1, eight channel planes) enters LDS once by LDS-DMA (per-plane descriptors: rows
//     outside the image arrive as zeros), conv1 computes the R + 2 intermediate rows into a second LDS tile (fp32 accumulate, scale /
//     shift, ReLU, ONE rounding to fp16 - what the two-launch path stores to HBM; rows outside the image are written as zeros: they
//     are conv2's padding), conv2 reads it, takes the identity from the staged input tile and stores 8 bytes per lane;
//   * one ds_read_b128 feeds TWO MFMAs (128 of the LDS array's 256 B/clk at full MFMA rate; a first version with one cout tile per
//     wave - one read per MFMA on all eight resident waves - sat on the LDS array: 20 us per block at N = 128, no better than two
//     launches), through a rolling window of eight operands in flight whose order is pinned by scheduling fences.
// Same operand mapping, k order (k-steps ascending, taps ascending) and epilogue arithmetic as conv_f16_kernel: bit-identical to two
// mp_f16_conv2d_fwd launches (tests/test_gpu_f16.py::test_fused_basicblock_equals_two_convs).
#include <type_traits>

#include "conv_f16.h"
#include "conv_f16_dev.h"

namespace mp {

namespace {

constexpr int kC64MaxPieces = 8;  // DMA pieces (64 x 16 B) per input plane
constexpr int kC64Win = 8;        // pixel-operand fragments in flight

// NQ = k-steps of 32 channels (C = 32 NQ: 64 or 128); the workgroup has 2 NQ waves - two pixel halves x NQ cout pairs (NQ = 2: four
// waves, two workgroups per CU; NQ = 4: eight waves, one workgroup per CU, two waves per SIMD).  PS1 / PS2 = pixel tiles of 16 PER
// WAVE for the intermediate / output band
template <int NQ, int PS1, int PS2>
__global__ __launch_bounds__(128 * NQ, NQ == 2 ? 2 : 1) void basicblock_f16_c64_kernel(const BlockF16Params p) {
    constexpr int T = 9, NPL = 4 * NQ, CS = 2, C = 32 * NQ, NTHREADS = 128 * NQ;  // NPL 8-channel planes; two cout tiles per wave
    extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
    u32x4* __restrict__ lds_in = smem16;                      // [8][plane_in]: pixel (r, x) at r Wp + x + 1, row 0 = image row y0 - 2
    u32x4* __restrict__ lds_mid = smem16 + NPL * p.plane_in;  // [8][plane_mid]: row 0 = image row y0 - 1; + one dummy element behind
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 1, wc = wave >> 1;  // pixel half, cout pair (32 couts = one PAIR of cout tiles: 16-byte elements per lane)
    const int lq = lane >> 4, lr = lane & 15;

    int b = blockIdx.x;
    {
        const int nb = p.total_blocks, q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, j = b >> 3;
        b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const int n = b / p.tiles_y, y0 = (b - n * p.tiles_y) * p.R;
    const int HW = p.H * p.W, P = p.Wp;
    const unsigned plane_bytes = (unsigned)HW * 16u;

    // ---- input tile by LDS-DMA: wave w stages planes w and w + 2 NQ; the slot -> (row, column) map is the same for every plane:
    //      decoded once (the division is a quarter-rate multiply)
    {
        const int ppp = p.plane_in >> 6;
        const char* img = reinterpret_cast<const char*>(p.x) + (size_t)n * NPL * plane_bytes;
        const int rows_in = p.R + 4;
        unsigned piece_off[kC64MaxPieces];
#pragma unroll
        for (int s = 0; s < kC64MaxPieces; ++s) {
            const unsigned slot = (unsigned)(s * 64 + lane);
            const unsigned r = __umulhi(slot, p.magic_w);  // / (W + 1)
            const int c = (int)(slot - r * P) - 1;
            piece_off[s] = (r < (unsigned)rows_in && c >= 0) ? (unsigned)(((int)r + y0 - 2) * p.W + c) * 16u : kOob;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pl = wave + 2 * NQ * j;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(img + (size_t)pl * plane_bytes, plane_bytes);
#pragma unroll
            for (int s = 0; s < kC64MaxPieces; ++s) {
                if (s >= ppp) break;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds_in + pl * p.plane_in + s * 64), 16,
                                                         piece_off[s], 0, 0, 0);
            }
        }
    }
    // ---- weight fragments of this wave's cout pair: a ring of ONE k-step (9 taps x 2 cout tiles = 72 registers).  The ring walks
    //      conv1 k-step 0 -> ... -> conv1 k-step NQ - 1 -> conv2 k-step 0 -> ...: the fragment of tap t is replaced by the next
    //      sequence element's right behind the MFMAs that consumed it, i.e. one k-step (9 taps) of prefetch distance
    const unsigned w_bytes = (unsigned)(NQ * T * 4 * C * 16);
    const __amdgpu_buffer_rsrc_t rs_w1 = make_rsrc(p.w1, w_bytes), rs_w2 = make_rsrc(p.w2, w_bytes);
    unsigned a_off[CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) a_off[cs] = (unsigned)(lq * C + wc * 32 + f16_a_row<CS>(cs, lr)) * 16u;
    constexpr unsigned kTapBytes = 4u * (unsigned)C * 16u;
    u32x4 A[T][CS];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) A[t][cs] = __builtin_amdgcn_raw_buffer_load_b128(rs_w1, a_off[cs] + (unsigned)t * kTapBytes, 0, 0);

    // ---- the intermediate tile's halo column (one zero slot between rows, + the slot behind the last row): written once
    {
        const int per_plane = p.R + 3;
        const u32x4 zero = (u32x4){0u, 0u, 0u, 0u};
        for (int i = tid; i < NPL * per_plane; i += NTHREADS) {
            const int pl = i / per_plane, r = i - pl * per_plane;
            lds_mid[pl * p.plane_mid + r * P] = zero;
        }
    }
    // ---- conv1: intermediate pixel px = 16 (wp PS1 + ps) + lr -> (row rm, column x); its window origin in the input tile is
    //      rm Wp + x.  Registers are what this kernel is short of (two workgroups per CU = 256 per lane): the lane's slot in the
    //      intermediate tile is b1 + a lane constant, "row inside the image" / "real pixel" are one bit per pixel tile
    unsigned b1[PS1];
    unsigned in_mask = 0, ok_mask = 0;
#pragma unroll
    for (int ps = 0; ps < PS1; ++ps) {
        const unsigned px = (unsigned)((wp * PS1 + ps) * 16 + lr);
        const bool ok = px < (unsigned)p.M1;
        const unsigned rm = fastdiv(ok ? px : 0u, p.W, p.magic_rw);
        const unsigned x = (ok ? px : 0u) - rm * p.W;
        b1[ps] = (unsigned)(lq * p.plane_in + rm * P + x) * 16u;
        const int yy = y0 - 1 + (int)rm;
        in_mask |= (yy >= 0 && yy < p.H) ? 1u << ps : 0u;   // else the row is conv2's zero padding
        ok_mask |= ok ? 1u << ps : 0u;
    }
    // byte address of the lane's 16 bytes (one channel block of one pixel) in the intermediate tile = b1 - lq plane_in 16 + this
    const unsigned m_delta = (unsigned)(NPL * p.plane_in + (wc * 4 + lq) * p.plane_mid + 1 - lq * p.plane_in) * 16u;
    const unsigned dummy = (unsigned)(NPL * (p.plane_in + p.plane_mid)) * 16u;
    // the DMA pieces are older than the 18 weight loads: all but 18 done = this wave's part of the tile has landed
    __builtin_amdgcn_s_waitcnt(0x0F70 | (18 & 15) | ((18 >> 4) << 14));
    __syncthreads();

    const char* lbase = reinterpret_cast<const char*>(smem16);
    const u32x2 none = (u32x2){0u, 0u};
    // one convolution's MFMA stream over the wave's PS pixel tiles: operand i = (q T + t) PS + ps feeds CS MFMAs; a rolling window
    // of kC64Win operands in flight; the weight ring refilled from (rs_next, q_next) tap by tap (q_next < 0: nothing follows)
    auto conv = [&](auto& acc, const unsigned* bsrc, int plane, const __amdgpu_buffer_rsrc_t& rs_same, const __amdgpu_buffer_rsrc_t& rs_other,
                    bool has_other, auto ps_tag) __attribute__((always_inline)) {
        constexpr int PS = decltype(ps_tag)::value;
        constexpr int M = NQ * T * PS;
        u32x4 win[kC64Win];
        auto b_at = [&](int i) __attribute__((always_inline)) {
            const int ps = i % PS, qt = i / PS, q = qt / T, t = qt - q * T;
            return *reinterpret_cast<const u32x4*>(lbase + bsrc[ps] + (unsigned)(q * 4 * plane + (t / 3) * P + (t % 3)) * 16u);
        };
#pragma unroll
        for (int i = 0; i < kC64Win; ++i) win[i] = b_at(i);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const int ps = i % PS, qt = i / PS, q = qt / T, t = qt - q * T;
#pragma unroll
            for (int cs = 0; cs < CS; ++cs)
                acc[ps][cs] = (q == 0 && t == 0)
                                  ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[t][cs]), __builtin_bit_cast(f16x8, win[i % kC64Win]),
                                                                           (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[t][cs]), __builtin_bit_cast(f16x8, win[i % kC64Win]),
                                                                           acc[ps][cs], 0, 0, 0);
            if (i + kC64Win < M) win[i % kC64Win] = b_at(i + kC64Win);
            if (ps == PS - 1) {
                // this tap's fragments are free: the next k-step's (same conv, k-step 1 - or the other conv's k-step 0) take their place
                if (q + 1 < NQ) {
#pragma unroll
                    for (int cs = 0; cs < CS; ++cs)
                        A[t][cs] = __builtin_amdgcn_raw_buffer_load_b128(rs_same, a_off[cs] + (unsigned)((q + 1) * T + t) * kTapBytes, 0, 0);
                } else if (has_other) {
#pragma unroll
                    for (int cs = 0; cs < CS; ++cs)
                        A[t][cs] = __builtin_amdgcn_raw_buffer_load_b128(rs_other, a_off[cs] + (unsigned)t * kTapBytes, 0, 0);
                }
            }
            // source order IS the schedule: the MFMAs of operand i, then the read of operand i + 8 into the slot it freed.  Left alone
            // hipcc regroups the reads right in front of their MFMAs (lgkmcnt(1) behind every read: every operand's LDS latency exposed)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        f32x4 acc[PS1][CS];
        conv(acc, b1, p.plane_in, rs_w1, rs_w2, true, std::integral_constant<int, PS1>{});
        // epilogue 1: folded BatchNorm + ReLU, one rounding, the lane's 16 bytes (8 couts of one pixel) into the intermediate tile
        f32x4 sc[CS], sh[CS];
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            sc[cs] = *reinterpret_cast<const f32x4*>(p.scale1 + wc * 32 + f16_d_cout<CS>(cs, lq));
            sh[cs] = *reinterpret_cast<const f32x4*>(p.shift1 + wc * 32 + f16_d_cout<CS>(cs, lq));
        }
#pragma unroll
        for (int ps = 0; ps < PS1; ++ps) {
            u32x2 lo = f16_pack4(f16_epi4(acc[ps][0], sc[0], sh[0], false, none, false, none, 1));
            u32x2 hi = f16_pack4(f16_epi4(acc[ps][1], sc[1], sh[1], false, none, false, none, 1));
            if (!((in_mask >> ps) & 1u)) lo = hi = none;
            *reinterpret_cast<u32x4*>(const_cast<char*>(lbase) + (((ok_mask >> ps) & 1u) ? b1[ps] + m_delta : dummy)) = (u32x4){lo.x, lo.y, hi.x, hi.y};
        }
    }
    __syncthreads();
    {
        // ---- conv2: output pixel px -> (row ro, column x); window origin in the intermediate tile ro Wp + x
        unsigned b2[PS2], o_off[PS2];
        const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, (size_t)p.N * NPL * plane_bytes);
        // the identity's 16 bytes in the staged input tile = b2 (an intermediate-tile address) + this
        const unsigned id_delta = (unsigned)((wc * 4 + lq) * p.plane_in + 2 * P + 1) * 16u - (unsigned)(NPL * p.plane_in + lq * p.plane_mid) * 16u;
#pragma unroll
        for (int ps = 0; ps < PS2; ++ps) {
            const unsigned px = (unsigned)((wp * PS2 + ps) * 16 + lr);
            const bool ok = px < (unsigned)p.M2;
            const unsigned ro = fastdiv(ok ? px : 0u, p.W, p.magic_rw);
            const unsigned x = (ok ? px : 0u) - ro * p.W;
            b2[ps] = (unsigned)(NPL * p.plane_in + lq * p.plane_mid + ro * P + x) * 16u;
            const int yy = y0 + (int)ro;
            o_off[ps] = (ok && yy < p.H) ? (unsigned)((n * NPL + wc * 4 + lq) * HW + yy * p.W + x) * 16u : kOob;
        }
        f32x4 acc[PS2][CS];
        conv(acc, b2, p.plane_mid, rs_w2, rs_w2, false, std::integral_constant<int, PS2>{});
        // epilogue 2: folded BatchNorm + identity (from the staged input tile) + ReLU, one rounding, 16-byte stores
        f32x4 sc[CS], sh[CS];
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            sc[cs] = *reinterpret_cast<const f32x4*>(p.scale2 + wc * 32 + f16_d_cout<CS>(cs, lq));
            sh[cs] = *reinterpret_cast<const f32x4*>(p.shift2 + wc * 32 + f16_d_cout<CS>(cs, lq));
        }
#pragma unroll
        for (int ps = 0; ps < PS2; ++ps) {
            const u32x4 idn = *reinterpret_cast<const u32x4*>(lbase + b2[ps] + id_delta);
            const u32x2 lo = f16_pack4(f16_epi4(acc[ps][0], sc[0], sh[0], true, (u32x2){idn.x, idn.y}, false, none, 1));
            const u32x2 hi = f16_pack4(f16_epi4(acc[ps][1], sc[1], sh[1], true, (u32x2){idn.z, idn.w}, false, none, 1));
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){lo.x, lo.y, hi.x, hi.y}, rs_o, o_off[ps], 0, 0);
        }
    }
}

template <int NQ, int PS1, int PS2>
int launch_c64(const BlockF16Params& p, size_t lds_bytes, hipStream_t s) {
    auto kern = basicblock_f16_c64_kernel<NQ, PS1, PS2>;
    static AttrOnce attr_set_once;
    if (attr_set_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(p.total_blocks), dim3(128 * NQ), lds_bytes, s, p);
    return check_launch();
}

}  // namespace

// geometry: exactly 64 or 128 channels; a band of R rows whose R + 2 intermediate rows fill at most 2 x PS1 pixel tiles of 16 (2 x PS2
// for the output): 64 channels 8 / 6 tiles per wave (two workgroups per CU), 128 channels 4 / 3 (one eight-wave workgroup per CU)
bool blockf16_c64_build(const void* x, const void* w1, const float* scale1, const float* shift1, const void* w2, const float* scale2,
                        const float* shift2, void* out, int n, int c, int h, int w, int rows, BlockF16Launch& L) {
    if ((c != 64 && c != 128) || x == out) return false;
    if (const char* e = knob("MP_F16_BLOCK_C64"))
        if (atoi(e) == 0) return false;
    const int npl = c / 8, ps1 = c == 64 ? 8 : 4, ps2 = c == 64 ? 6 : 3;
    if ((size_t)n * npl * h * w * 16 > 0x7FFFFFF0u) return false;
    BlockF16Params p{};
    p.x = x; p.w1 = w1; p.w2 = w2; p.scale1 = scale1; p.shift1 = shift1; p.scale2 = scale2; p.shift2 = shift2; p.out = out;
    p.N = n; p.H = h; p.W = w;
    p.Wp = w + 1;
    int best = 0;
    for (int R = (rows > 0 ? rows : 12); R >= 1; --R) {
        if (R > h && R > 1) continue;
        if ((R + 2) * w > 32 * ps1 || R * w > 32 * ps2) continue;
        const int plane_in = round_up((R + 4) * p.Wp + 1, 64);
        if (plane_in / 64 > kC64MaxPieces) continue;
        const size_t bytes = ((size_t)npl * (plane_in + round_up((R + 2) * p.Wp + 1, 16)) + 1) * 16;
        if (bytes > (size_t)(c == 64 ? 78 : 150) * 1024) continue;  // 64 channels: two workgroups per CU
        best = R;
        break;
    }
    if (best == 0 || (rows > 0 && best != rows)) return false;
    p.R = best;
    p.plane_in = round_up((best + 4) * p.Wp + 1, 64);
    p.plane_mid = round_up((best + 2) * p.Wp + 1, 16);
    p.M1 = (best + 2) * w;
    p.M2 = best * w;
    p.tiles_y = (h + best - 1) / best;
    p.tiles_total = p.tiles_y * n;
    p.tiles_per_wg = 1;
    p.total_blocks = p.tiles_total;
    p.magic_w = magic_of((unsigned)p.Wp);   // DMA slot -> row
    p.magic_rw = magic_of((unsigned)w);     // pixel -> row
    L.p = p;
    L.small = c == 64 ? 4 : 5;  // marks this kernel (0 / 1: first structure, 2: second structure of the 32-channel block)
    L.lds_bytes = ((size_t)npl * (p.plane_in + p.plane_mid) + 1) * 16;
    return true;
}

int blockf16_c64_launch(const BlockF16Launch& L, hipStream_t s) {
    return L.small == 5 ? launch_c64<4, 4, 3>(L.p, L.lds_bytes, s) : launch_c64<2, 8, 6>(L.p, L.lds_bytes, s);
}

}  // namespace mp
