// Fused BasicBlock of the 32-channel branch, second structure (same contract and bit-identical results as basicblock_f16.hip):
//
//     out = relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x)          x, out: channel-blocked fp16 [N][4][H][W][8]
//
// What changed against the first kernel, and why (round-2 measurements: 29 us per block = 0.20 of the fp16 MFMA peak, the
// dominant fp16 launch; per row band two weight sets swapped through LDS, four barriers, register-staged input rows, 8-byte
// stores):
//   * BOTH weight sets live in REGISTERS for the whole life of the persistent workgroup (2 x 9 taps x 2 cout tiles x 16 B per lane
//     = 144 VGPRs, fetched once from the packed lane-linear weights): no weight traffic, no weight LDS reads, no swap barriers;
//   * 512-thread workgroups, ONE per CU = two waves per SIMD that share nothing but the two LDS tiles: while one wave is in an
//     epilogue (LDS writes / global stores) its SIMD partner is in an MFMA loop;
//   * input row bands arrive by LDS-DMA (buffer_load ... lds; halo column, rows outside the image and padding arrive as zeros
//     through the range check), double-buffered: the next band flies under the current band's two convolutions;
//   * cout tiles are paired (conv_f16_dev.h): the intermediate goes to LDS with one ds_write_b128 per pixel, the identity comes
//     back with one ds_read_b128, the result leaves with one 16-byte store;
//   * two barriers per band (intermediate tile complete / tiles free).
// LDS image of a tile: [plane][row][W + 1] 16-byte elements + 1: one zero column between rows is the right halo of row r and the
// left halo of row r + 1 (pixel (r, x) at r * (W + 1) + x + 1).
#include <stdlib.h>

#include <type_traits>

#include "conv_f16.h"
#include "conv_f16_dev.h"

namespace mp {

#if defined(MP_BLOCK_STAMPS) && MP_BLOCK_STAMPS
unsigned long long* g_block_stamp_buf = nullptr;
size_t g_block_stamp_bytes = 0;
#endif

namespace {

#ifndef MP_BLOCK_STAMPS
#define MP_BLOCK_STAMPS 0  // 1: s_memtime phase stamps of waves 0 and 4 into BlockF16Params::dbg (never in the product build)
#endif
#if MP_BLOCK_STAMPS
#define BLOCK_STAMP(i)                                                                                      \
    do {                                                                                                    \
        if (p.dbg && (wave & 3) == 0) {                                                                     \
            unsigned long long t_;                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            if (lane == 0) p.dbg[((size_t)blockIdx.x * 2 + (wave >> 2)) * 16 + (i)] = t_;                   \
        }                                                                                                   \
    } while (0)
#else
#define BLOCK_STAMP(i) do { } while (0)
#endif

template <int WAVES, int PS1, int PS2>
__global__ __launch_bounds__(WAVES * 64, 2) void basicblock_f16_v2_kernel(const BlockF16Params p) {
    constexpr int CS = 2;
    constexpr int kV2Waves = WAVES;
    extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
    u32x4* __restrict__ lds_in = smem16;                          // [2][4][plane_in]
    u32x4* __restrict__ lds_mid = smem16 + 2 * 4 * p.plane_in;    // [4][plane_mid]
    f32x4* __restrict__ lds_bn = reinterpret_cast<f32x4*>(lds_mid + 4 * p.plane_mid);  // [scale1|shift1|scale2|shift2] x 32 fp32
    const int dummy_mid = 4 * p.plane_mid + 32;                   // unit (from lds_mid) that masked writes land in

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const int P = p.Wp;  // W + 1

    int b = blockIdx.x;
    {
        const int nb = p.total_blocks, q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, j = b >> 3;
        b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const int t_begin = b * p.tiles_per_wg, t_end = min(t_begin + p.tiles_per_wg, p.tiles_total);
    const int HW = p.H * p.W;
    BLOCK_STAMP(0);

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, (size_t)p.N * 4 * HW * 16);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, (size_t)p.N * 4 * HW * 16);

    // LDS-DMA of band t into input buffer `buf`: every slot of the image is written, data or (range check) zero.  A wave issues
    // pieces (64 consecutive slots) wave, wave + WAVES, ...; what a lane's slot maps to - plane, tile row, column - does not
    // depend on the band, so it is decoded ONCE (two multiply-high divisions per piece cost more than the DMA issue itself when
    // redone per band: 1.6 - 2.1 k cycles per band in the phase stamps) and a band only adds its origin and checks the row.
    const int rows_in = p.R + 4;
    constexpr int kMaxPieces = WAVES == 8 ? 6 : 8;  // blockf16_v2_build admits only tiles of at most WAVES * kMaxPieces pieces
    unsigned piece_rel[kMaxPieces];  // byte offset relative to (image n, row yb); kOob = halo column / padding slot
    int piece_row[kMaxPieces];
    {
        const int img_slots = rows_in * P;
#pragma unroll
        for (int i = 0; i < kMaxPieces; ++i) {
            const unsigned s = (unsigned)((wave + i * kV2Waves) * 64 + lane);
            const unsigned pl = fastdiv(s, p.plane_in, p.magic_rw);
            const unsigned rem = s - pl * p.plane_in;
            const unsigned r = fastdiv(rem, P, p.magic_w);
            const int c = (int)(rem - r * P) - 1;
            const bool ok = pl < 4u && rem < (unsigned)img_slots && c >= 0;
            piece_rel[i] = ok ? (pl * HW + r * p.W + c) * 16u : kOob;
            piece_row[i] = (int)r;
        }
    }
    const int n_pieces = (4 * p.plane_in + 63) / 64;  // plane_in is a multiple of 16: 4 planes are a whole number of pieces
    auto dma_band = [&](int t, int buf) {
        const int ty = t % p.tiles_y, n = t / p.tiles_y;
        const int yb = ty * p.R - 2;
        const unsigned base = (unsigned)((n * 4 * HW + yb * p.W) * 16);
        u32x4* dst = lds_in + buf * 4 * p.plane_in;
#pragma unroll
        for (int i = 0; i < kMaxPieces; ++i) {
            const int piece = wave + i * kV2Waves;
            if (piece >= n_pieces) break;  // wave-uniform
            const int yin = yb + piece_row[i];
            const unsigned off = (piece_rel[i] != kOob && yin >= 0 && yin < p.H) ? base + piece_rel[i] : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(dst + piece * 64), 16, off, 0, 0, 0);
        }
    };

    // ---- tile-independent pixel addressing
    int m_slot[PS1], m_row[PS1];  // conv1 pixel: slot r * P + x in a tile plane (operand base; + 1 = its own slot), row; -1 = none
#pragma unroll
    for (int ps = 0; ps < PS1; ++ps) {
        const unsigned pl = (unsigned)((wave * PS1 + ps) * 16 + lr);
        const unsigned r = fastdiv(pl, p.W, p.magic_wo);
        const unsigned col = pl - r * p.W;
        const bool ok = pl < (unsigned)p.M1;
        m_slot[ps] = ok ? (int)(r * P + col) : -1;
        m_row[ps] = (int)r;
    }
    int o_slot[PS2], o_rc[PS2];  // conv2 pixel: slot o * P + x; (row << 16 | col)
#pragma unroll
    for (int ps = 0; ps < PS2; ++ps) {
        const unsigned pl = (unsigned)((wave * PS2 + ps) * 16 + lr);
        const unsigned r = fastdiv(pl, p.W, p.magic_wo);
        const unsigned col = pl - r * p.W;
        const bool ok = pl < (unsigned)p.M2;
        o_slot[ps] = ok ? (int)(r * P + col) : -1;
        o_rc[ps] = (int)((r << 16) | col);
    }
    // a wave whose pixel tiles all lie beyond the band (conv2 has fewer pixels than conv1) skips the MFMA loop: wave-uniform
    const bool wave_has1 = wave * PS1 * 16 < p.M1, wave_has2 = wave * PS2 * 16 < p.M2;

    // one 3x3 conv over an LDS tile with the weights in registers: 9 taps, pixel operands fetched one tap ahead
    auto mma9 = [&](const u32x4* __restrict__ lin, auto& acc, const auto& b_off, const u32x4 (&A)[9][CS], auto ps_tag) {
        constexpr int PS = decltype(ps_tag)::value;
        u32x4 bv[PS];
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) bv[ps] = lin[b_off[ps]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tn = (t + 1 < 9) ? t + 1 : 0;  // the final prefetch re-reads a valid position (discarded)
            const int in_off = (tn / 3) * P + (tn % 3);
            u32x4 bn[PS];
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) bn[ps] = lin[b_off[ps] + in_off];
#pragma unroll
            for (int ps = 0; ps < PS; ++ps)
#pragma unroll
                for (int cs = 0; cs < CS; ++cs)
                    acc[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[t][cs]),
                                                                          __builtin_bit_cast(f16x8, bv[ps]), acc[ps][cs], 0, 0, 0);
            {
                constexpr int NM = PS * CS;
#pragma unroll
                for (int i = 0; i < PS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NM - PS, 0);
            }
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) bv[ps] = bn[ps];
        }
    };

    // the first band's DMA goes out BEFORE the weight fragments: its HBM latency hides under their 36 loads per lane
    if (t_begin < t_end) dma_band(t_begin, 0);
    // ---- both weight sets: ONE copy per workgroup by LDS-DMA into the (still unused) second input buffer, from there into every
    //      wave's registers (lane-linear packed layout [tap][4][32 couts] x 16 B; rows paired, conv_f16_dev.h).  Eight waves
    //      fetching their 36 fragments each straight from L2 moved 288 KB through the CU's vector-memory path: 4.5 k of the 9.3 k
    //      prologue cycles in the phase stamps.
    u32x4 A1[9][CS], A2[9][CS];
    u32x4* __restrict__ lds_wstage = lds_in + 4 * p.plane_in;  // 2 x 1152 elements <= 4 * plane_in (checked by blockf16_v2_build)
    {
        const __amdgpu_buffer_rsrc_t rs_w1 = make_rsrc(p.w1, (size_t)9 * 4 * 32 * 16);
        const __amdgpu_buffer_rsrc_t rs_w2 = make_rsrc(p.w2, (size_t)9 * 4 * 32 * 16);
        for (int piece = wave; piece < 36; piece += kV2Waves) {  // 18 pieces of 64 elements per weight set
            const bool second = piece >= 18;
            const unsigned off = (unsigned)(((second ? piece - 18 : piece) * 64 + lane) * 16);
            if (second)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (__attribute__((address_space(3))) void*)(lds_wstage + piece * 64), 16, off, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (__attribute__((address_space(3))) void*)(lds_wstage + piece * 64), 16, off, 0, 0, 0);
        }
    }
    {
        // the intermediate tile's halo slots are never written again: zero the tile once; folded BatchNorm parameters to LDS
        const int n16 = 4 * p.plane_mid + 33;
        const u32x4 zero = (u32x4){0u, 0u, 0u, 0u};
        for (int i = tid; i < n16; i += WAVES * 64) lds_mid[i] = zero;
        __syncthreads();
        if (tid < 32) {
            const float* src = tid < 8 ? p.scale1 : tid < 16 ? p.shift1 : tid < 24 ? p.scale2 : p.shift2;
            lds_bn[tid] = *reinterpret_cast<const f32x4*>(src + 4 * (tid & 7));
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // first band, both weight sets and the BatchNorm parameters are in LDS
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            const int e = (t * 4 + lq) * 32 + f16_a_row<CS>(cs, lr);
            A1[t][cs] = lds_wstage[e];
            A2[t][cs] = lds_wstage[1152 + e];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // every wave holds its fragments: the staging area is an input buffer again
    BLOCK_STAMP(1);

    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        const int ty = t % p.tiles_y, n = t / p.tiles_y;
        const int y0 = ty * p.R;
        const u32x4* __restrict__ in_cur = lds_in + cur * 4 * p.plane_in;
        const bool rec = t == t_begin + 1;  // stamps: the second band of the run (steady state)
        if (rec) BLOCK_STAMP(2);
        if (t + 1 < t_end) dma_band(t + 1, cur ^ 1);  // flies under this band's two convolutions
        if (rec) BLOCK_STAMP(3);

        // ---- conv1 + bn1 + relu over the R + 2 intermediate rows -> lds_mid; rows outside the image are conv2's zero padding
        if (wave_has1) {
            f32x4 acc[PS1][CS];
#pragma unroll
            for (int ps = 0; ps < PS1; ++ps)
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) acc[ps][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
            int b1_off[PS1];
#pragma unroll
            for (int ps = 0; ps < PS1; ++ps) b1_off[ps] = lq * p.plane_in + (m_slot[ps] >= 0 ? m_slot[ps] : 0);
            mma9(in_cur, acc, b1_off, A1, std::integral_constant<int, PS1>{});
            if (rec) BLOCK_STAMP(4);
            // lane = couts 8 lq .. 8 lq + 7 (channel block lq) of its pixel: scale / shift of those couts
            const f32x4 sc0 = lds_bn[2 * lq], sc1 = lds_bn[2 * lq + 1], sh0 = lds_bn[8 + 2 * lq], sh1 = lds_bn[8 + 2 * lq + 1];
            const u32x2 none = (u32x2){0u, 0u};
#pragma unroll
            for (int ps = 0; ps < PS1; ++ps) {
                u32x2 lo = f16_pack4(f16_epi4(acc[ps][0], sc0, sh0, false, none, false, none, 1));
                u32x2 hi = f16_pack4(f16_epi4(acc[ps][1], sc1, sh1, false, none, false, none, 1));
                const int ym = y0 - 1 + m_row[ps];
                if (ym < 0 || ym >= p.H) { lo = none; hi = none; }
                const int slot = m_slot[ps] >= 0 ? lq * p.plane_mid + m_slot[ps] + 1 : dummy_mid;
                lds_mid[slot] = (u32x4){lo.x, lo.y, hi.x, hi.y};
            }
        }
        // the intermediate tile is complete.  A raw barrier behind an LDS-only wait: __syncthreads() would drain vmcnt and with it
        // the next band's DMA, which then has only conv1 to hide under - all CUs burst their bands at once, a band takes longer
        // to arrive than conv1 runs (the first version of this kernel waited here: 23 us per block)
        if (rec) BLOCK_STAMP(5);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (rec) BLOCK_STAMP(6);

        // ---- conv2 + bn2 + identity + relu over the R output rows -> HBM
        if (wave_has2) {
            f32x4 acc[PS2][CS];
#pragma unroll
            for (int ps = 0; ps < PS2; ++ps)
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) acc[ps][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
            int b2_off[PS2];
#pragma unroll
            for (int ps = 0; ps < PS2; ++ps) b2_off[ps] = lq * p.plane_mid + (o_slot[ps] >= 0 ? o_slot[ps] : 0);
            mma9(lds_mid, acc, b2_off, A2, std::integral_constant<int, PS2>{});
            if (rec) BLOCK_STAMP(7);
            const f32x4 sc0 = lds_bn[16 + 2 * lq], sc1 = lds_bn[16 + 2 * lq + 1], sh0 = lds_bn[24 + 2 * lq], sh1 = lds_bn[24 + 2 * lq + 1];
            const unsigned img = (unsigned)(n * 4 + lq) * HW * 16u;
            const u32x2 none = (u32x2){0u, 0u};
#pragma unroll
            for (int ps = 0; ps < PS2; ++ps) {
                const int orow = o_rc[ps] >> 16, ocol = o_rc[ps] & 0xFFFF;
                // identity: the input tile's own pixel (tile row orow + 2), same channel block
                const u32x4 idn = in_cur[lq * p.plane_in + (o_slot[ps] >= 0 ? o_slot[ps] : 0) + 2 * P + 1];
                const u32x2 lo = f16_pack4(f16_epi4(acc[ps][0], sc0, sh0, true, (u32x2){idn.x, idn.y}, false, none, 1));
                const u32x2 hi = f16_pack4(f16_epi4(acc[ps][1], sc1, sh1, true, (u32x2){idn.z, idn.w}, false, none, 1));
                const int yo = y0 + orow;
                const unsigned off = (o_slot[ps] >= 0 && yo < p.H) ? img + (unsigned)(yo * p.W + ocol) * 16u : kOob;
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){lo.x, lo.y, hi.x, hi.y}, rs_o, off, 0, 0);
            }
        }
        // both tiles are free (the next band may overwrite the intermediate, the band after it this input buffer) and the next
        // band's rows have landed: the DMA pieces are OLDER than this band's PS2 stores, so all but the PS2 youngest operations
        // are waited for - the stores themselves drain under the next band
        if (rec) BLOCK_STAMP(8);
        if (wave_has2) {
            __builtin_amdgcn_s_waitcnt(0x0F70 | PS2);  // vmcnt(PS2), expcnt / lgkmcnt not waited
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (rec) BLOCK_STAMP(9);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (rec) BLOCK_STAMP(10);
    }
    BLOCK_STAMP(11);
}

template <int WAVES, int PS1, int PS2>
int launch_block_v2(const BlockF16Params& p, size_t lds_bytes, hipStream_t s) {
    auto kern = basicblock_f16_v2_kernel<WAVES, PS1, PS2>;
    static AttrOnce attr_set_once;
    if (attr_set_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(p.total_blocks), dim3(WAVES * 64), lds_bytes, s, p);
    return check_launch();
}

}  // namespace

// geometry of the second structure; false = shape not covered (the first kernel serves it)
bool blockf16_v2_build(const void* x, const void* w1, const float* scale1, const float* shift1, const void* w2, const float* scale2,
                       const float* shift2, void* out, int n, int c, int h, int w, int rows, BlockF16Launch& L) {
    if (const char* e = knob("MP_F16_BLOCK_V2"))
        if (atoi(e) == 0) return false;
    if (c <= 24 || c > 32 || x == out) return false;
    if ((size_t)n * 4 * h * w * 16 > 0x7FFFFFF0u) return false;
    BlockF16Params p{};
    p.x = x; p.w1 = w1; p.w2 = w2; p.scale1 = scale1; p.shift1 = shift1; p.scale2 = scale2; p.shift2 = shift2; p.out = out;
    p.N = n; p.H = h; p.W = w;
    p.Wp = w + 1;
    // eight waves, one workgroup per CU (PS1 = 4, PS2 = 3: up to 8 rows of 48).  (A four-wave / two-workgroups-per-CU shape of the same
    // code measured 26.6 us against 23.3 us at N = 128 in round 2 and was removed in round 4.)
    const int shape = 0;
    const int waves = 8, PS1 = 4, PS2 = 3;
    int best = 0;
    for (int R = (rows > 0 ? rows : (shape ? 4 : 8)); R >= 1; --R) {
        if (R > h && R > 1) continue;
        if ((R + 2) * w > waves * PS1 * 16 || R * w > waves * PS2 * 16) continue;
        const size_t bytes = ((size_t)4 * (2 * round_up((R + 4) * p.Wp + 1, 16) + round_up((R + 2) * p.Wp + 1, 16)) + 33 + 32) * 16;
        if (bytes > (shape ? 78 : 150) * 1024) continue;
        if (4 * round_up((R + 4) * p.Wp + 1, 16) < 2 * 1152) continue;  // the second input buffer stages both weight sets once
        if ((4 * round_up((R + 4) * p.Wp + 1, 16) + 63) / 64 > waves * (shape ? 8 : 6)) continue;  // DMA pieces per wave
        best = R;
        break;
    }
    if (best == 0 || (rows > 0 && best != rows)) return false;
    p.R = best;
    p.plane_in = round_up((best + 4) * p.Wp + 1, 16);
    p.plane_mid = round_up((best + 2) * p.Wp + 1, 16);
    p.M1 = (best + 2) * w;
    p.M2 = best * w;
    p.tiles_y = (h + best - 1) / best;
    p.tiles_total = p.tiles_y * n;
    int groups = shape ? 512 : 256;  // workgroups resident at once
    if (const char* e = knob("MP_F16_BLOCK_GROUPS")) {  // tests: force long tile runs on small problems
        const int v = atoi(e);
        if (v >= 1) groups = v;
    }
    p.tiles_per_wg = (p.tiles_total + groups - 1) / groups;
    p.total_blocks = (p.tiles_total + p.tiles_per_wg - 1) / p.tiles_per_wg;
    p.magic_wo = magic_of((unsigned)w);            // pixel -> row
    p.magic_w = magic_of((unsigned)p.Wp);          // slot -> row
    p.magic_rw = magic_of((unsigned)p.plane_in);   // slot -> plane
#if MP_BLOCK_STAMPS
    p.dbg = (g_block_stamp_buf && (size_t)p.total_blocks * 2 * 16 * 8 <= g_block_stamp_bytes) ? g_block_stamp_buf : nullptr;
#endif
    L.p = p;
    L.small = 2 + shape;  // 2 / 3 mark the second structure (eight / four waves)
    L.lds_bytes = ((size_t)4 * (2 * p.plane_in + p.plane_mid) + 33 + 32) * 16;
    return true;
}

int blockf16_v2_launch(const BlockF16Launch& L, hipStream_t s) {
    return launch_block_v2<8, 4, 3>(L.p, L.lds_bytes, s);
}

}  // namespace mp

#if MP_BLOCK_STAMPS
extern "C" int mp_debug_set_block_stamp_buffer(void* dev_ptr, size_t bytes) {  // diagnostic build only (tools/block_probe.py)
    mp::g_block_stamp_buf = reinterpret_cast<unsigned long long*>(dev_ptr);
    mp::g_block_stamp_bytes = dev_ptr ? bytes : 0;
    return MP_OK;
}
#endif
