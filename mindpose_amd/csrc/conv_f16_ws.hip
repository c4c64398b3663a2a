// fp16-MFMA 3x3 stride-1 convolution, WEIGHT-STATIONARY persistent form (round 4) - the HRNet branch convs with 32 - 128 channels
// (hrnet.py:30-83 BasicBlock, :202-241 branches; W48 cfg :670-718).
//
// Why another structure: round-3 counters of the weights-in-registers kernel (conv_f16_wreg.hip) on W48's 96 -> 96 @48x36 layer -
// a wave lives 18.7 k cycles for 2.6 k cycles of MFMA, issues ~1000 VALU instructions for its 162 MFMAs (slot decoding, offsets,
// descriptors: all per short-lived workgroup), and every workgroup re-streams its weight slice through the CU's vector-memory path
// (43 - 85 of 64 B/clk).  Here a workgroup lives for the whole launch:
//   * ONE workgroup per CU (256 threads, one wave per SIMD, up to 512 registers per lane).  The wave's weight fragments - ALL k of
//     its CSW cout tiles: NQ k-steps x 9 taps x CSW x 16 B per lane = 72 NQ CSW registers - are loaded ONCE and stay in registers:
//     in steady state there is no weight traffic at all and no per-tile weight prologue;
//   * pixel tiles (R full-width rows of one image, all input channels) stream through a TWO-stage LDS ring filled by LDS-DMA
//     (buffer_load ... lds): tile t + 1 flies while tile t is on the matrix pipe; what a DMA piece maps to (plane, row, column) is
//     decoded once per kernel, a tile only adds its row origin, and rows outside the image / the halo column arrive as zeros through
//     the range check of a PER-PLANE buffer descriptor (no compare per piece);
//   * every B fragment (one ds_read_b128) feeds CSW >= 2 MFMAs, so the LDS read rate stays <= 128 B/clk per CU;
//   * ONE barrier per tile; the tile's stores drain under the next tile (counted vmcnt: the DMA pieces are older than the stores).
// Same operand mapping, k order (k-steps ascending, taps ascending) and epilogue arithmetic as conv_f16_kernel / conv_f16_wreg_kernel:
// outputs are bit-identical to those (tests/test_gpu_f16.py).  Cout slices (n_ct > 1) re-read the input tile from L2.
// LDS image of a tile: [plane][row][W + 1] 16-byte elements, plane pitch a multiple of 64 elements (a DMA piece never straddles
// planes): one zero column between rows is the right halo of row r and the left halo of row r + 1.
#include <type_traits>

#include "conv_f16.h"
#include "conv_f16_dev.h"

namespace mp {

#if defined(MP_WS_STAMPS) && MP_WS_STAMPS
unsigned long long* g_ws_stamp_buf = nullptr;
size_t g_ws_stamp_bytes = 0;
#endif

namespace {

#ifndef MP_WS_ABLATE
#define MP_WS_ABLATE 0  // kernel work only (tools/ws_ablate.sh; results are WRONG): 1 no epilogue arithmetic, 2 no DMA of the next tile,
#endif                  // 4 no residual loads, 8 no pixel-operand reads, 16 no stores, 32 no accumulator copy
#ifndef MP_WS_STAMPS
#define MP_WS_STAMPS 0  // 1: s_memtime phase stamps of wave 0 into g_ws_stamp_buf (tools/ws_probe.py; never in the product build)
#endif
#if MP_WS_STAMPS
#define WS_STAMP(i)                                                                          \
    do {                                                                                     \
        if (dbg && wave == 0 && (i) < 64) {                                                  \
            unsigned long long t_;                                                           \
            __builtin_amdgcn_sched_barrier(0);                                               \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
            __builtin_amdgcn_sched_barrier(0);                                               \
            if (lane == 0) dbg[(size_t)blockIdx.x * 64 + (i)] = t_;                          \
        }                                                                                    \
    } while (0)
#else
#define WS_STAMP(i) do { } while (0)
#endif

constexpr int kWsMaxPieces = 8;  // DMA pieces (64 x 16 B) per plane: f16_configure_ws admits no more
constexpr int kWsAgprRegs = 252;  // AGPRs handed out by hand: the accumulators first, weight fragments in what is left

// The MFMA as inline assembly: hipcc keeps MFMA source operands in VGPRs and uses AGPRs only as spill space (a 324-register weight set
// then costs four v_accvgpr_read per MFMA and the spill traffic wrecks the operand prefetch - the first build of this kernel waited
// lgkmcnt(0) right behind every ds_read).  The hardware takes the A operand from an AGPR tuple directly; as an asm operand of class
// "a" the fragment is loaded into AGPRs (buffer_load ... a[..]) and stays there.  Hazards inside these statements: back-to-back MFMAs
// on independent accumulators need no wait states, the same-accumulator chain (SrcC = vDst of the previous one) needs none either; the
// epilogue's first VALU read of an accumulator sits behind mfma_results_ready().
// The ACCUMULATORS are AGPR operands too ("+a"): the two accumulator sets of the pipelined tile loop plus most weight fragments fill the
// 256 AGPRs, the VGPR half keeps the remaining fragments and everything the VALU touches (the epilogue reads an accumulator through
// v_accvgpr_read, in the shadow of the next tile's MFMAs).
__device__ __forceinline__ void mfma_a(f32x4& acc, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x4& acc, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_a0(f32x4& acc, const u32x4& a, const u32x4& b) {  // first MFMA of a chain: C = 0
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&a"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v0(f32x4& acc, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&a"(acc) : "v"(a), "v"(b));
}
// accumulator (AGPR tuple) -> VGPRs, HERE: left to the compiler the copy floats to wherever the register allocator likes it (it read
// whole accumulator sets at the top of the tile loop and spilled them)
__device__ __forceinline__ f32x4 acc_read(const f32x4& acc) {
    f32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v.x) : "a"(acc.x));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v.y) : "a"(acc.y));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v.z) : "a"(acc.z));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v.w) : "a"(acc.w));
    return v;
}
__device__ __forceinline__ void load_a(u32x4& dst, unsigned voff, const u32x4& rsrc) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=a"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
// (fragments kept in VGPRs are loaded by the COMPILER-TRACKED builtin at their use site: a register the allocator decides to spill or
// copy is then waited for first.  An asm load into a VGPR is invisible to the waitcnt pass - round 4's statistics builds stored such
// registers to scratch before the data had arrived.  The asm AGPR loads stay: the AGPR budget is handed out by hand (kWsAgprRegs), the
// allocator never moves those tuples, and the build fails if any instantiation gets scratch memory - csrc/Makefile.)

template <int NQ, int CSW, int WAVES_P, int PS, int OCC, int STATS, bool RES>
__global__ __launch_bounds__(256, OCC) void conv_f16_ws_kernel(const ConvF16Params p
#if MP_WS_STAMPS
                                                               , unsigned long long* dbg
#endif
) {
    constexpr int T = 9;
    constexpr int WAVES_C = 4 / WAVES_P;
    constexpr int CT = 16 * CSW * WAVES_C;  // couts per workgroup
    constexpr int NP = CSW / 2, NS = CSW - 2 * NP;
    constexpr int NST = PS * (NP + NS);  // stores per tile and wave
    constexpr int NF = NQ * T * CSW;     // weight fragments of a wave
    constexpr int kAgprFrags = (kWsAgprRegs / OCC - 4 * PS * CSW) / 4;  // 4 AGPRs per fragment behind the PS CSW accumulators of 4
    constexpr int NFA = NF < kAgprFrags ? NF : kAgprFrags, NFV = NF - NFA;
    extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
    u32x4* __restrict__ lds_in = smem16;  // [2][PK][plane]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp_i = wave % WAVES_P, wc_i = wave / WAVES_P;
    const int lq = lane >> 4, lr = lane & 15;
    WS_STAMP(0);

    int b = blockIdx.x;
    {
        const int nb = p.total_blocks, q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, j = b >> 3;
        b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const int ct = b % p.n_ct;  // the cout slices of one tile run sit next to each other: they share the input through one L2
    const int grp = b / p.n_ct;
    const int t_begin = grp * p.tiles_per_wg, t_end = min(t_begin + p.tiles_per_wg, p.tiles_total);
    const int HW = p.H * p.W;
    const int P = p.Wp;  // row pitch W + 1
    const int buf_elems = p.in_buf;
    const unsigned plane_bytes = (unsigned)HW * 16u;

    // ---- LDS-DMA.  Wave w stages planes w, w + 4, ...; piece s of a plane covers its slots 64 s ...: the slot -> (row, column) map
    //      is the same for every plane and tile, decoded once
    const int ppp = p.upc;  // pieces per plane
    unsigned piece_rel[kWsMaxPieces];  // byte offset inside the image plane relative to the tile's first staged row; kOob = zero slot
#pragma unroll
    for (int s = 0; s < kWsMaxPieces; ++s) {
        const unsigned slot = (unsigned)(s * 64 + lane);
        const unsigned r = __umulhi(slot, p.magic_ncols);  // / (W + 1), W >= 1
        const int c = (int)(slot - r * P) - 1;
        piece_rel[s] = (slot < (unsigned)p.img_plane && c >= 0) ? (r * p.W + c) * 16u : kOob;
    }
    auto tile_pos = [&](int t, int& n, int& y0) __attribute__((always_inline)) {
        n = p.tiles_y == 1 ? t : (int)__umulhi((unsigned)t, p.magic_rows);
        y0 = (t - n * p.tiles_y) * p.R;
    };
    auto dma_tile = [&](int t, int buf) __attribute__((always_inline)) {
        int n, y0;
        tile_pos(t, n, y0);
        const unsigned row_off = (unsigned)((y0 - 1) * p.W * 16);  // first staged row = y0 - 1 (-1: wraps out of range)
        const char* img = reinterpret_cast<const char*>(p.x) + (size_t)n * p.C8in * plane_bytes;
        u32x4* dst = lds_in + buf * buf_elems;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int pl = wave + 4 * j;
            if (pl >= p.C8in) break;  // wave-uniform
            // descriptor of image plane (n, pl) alone: rows above / below the image fall outside it and arrive as zeros
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(img + pl * plane_bytes, plane_bytes);
#pragma unroll
            for (int s = 0; s < kWsMaxPieces; ++s) {
                if (s >= ppp) break;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + pl * p.plane + s * 64), 16,
                                                         piece_rel[s] + row_off, 0, 0, 0);
            }
        }
    };
    if (t_begin < t_end) dma_tile(t_begin, 0);  // flies under the weight loads

    // ---- weight operand: ALL k of this wave's couts, from the packed weights [kq][T][4][Cout_pad16] x 16 B, for the kernel's life:
    //      fragment f = (q * T + t) * CSW + cs; the first NFA in AGPR tuples, the rest in VGPRs
    u32x4 Aa[NFA], Av[NFV ? NFV : 1];
    {
        u32x4 rs_w;
        const size_t w_bytes = (size_t)NQ * T * 4 * p.Cout_pad16 * 16;
        rs_w.x = __builtin_amdgcn_readfirstlane((unsigned)(size_t)p.wp);
        rs_w.y = __builtin_amdgcn_readfirstlane((unsigned)((size_t)p.wp >> 32)) & 0xFFFFu;
        rs_w.z = __builtin_amdgcn_readfirstlane((unsigned)w_bytes);
        rs_w.w = 0x00020000u;
        const unsigned tap_bytes = 4u * p.Cout_pad16 * 16u;
        unsigned goff[CSW];
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) {
            const int co = ct * CT + wc_i * CSW * 16 + f16_a_row<CSW>(cs, lr);  // < Cout_pad16: every wave owns real cout tiles
            goff[cs] = (unsigned)(lq * p.Cout_pad16 + co) * 16u;
        }
        asm volatile("s_nop 4" ::: "memory");  // v_readfirstlane -> descriptor read by the first load
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const unsigned off = goff[f % CSW] + (unsigned)(f / CSW) * tap_bytes;
            if (f < NFA) load_a(Aa[f], off, rs_w);
        }
        // the VGPR fragments: AFTER every asm load, so that a compiler-placed vmcnt(k) - which counts only the loads the compiler
        // knows, all of them younger than the asm loads - never under-waits
        asm volatile("" ::: "memory");
        const __amdgpu_buffer_rsrc_t rs_wv = make_rsrc(p.wp, w_bytes);
#pragma unroll
        for (int f = NFA; f < NF; ++f)
            Av[f - NFA] = __builtin_amdgcn_raw_buffer_load_b128(rs_wv, goff[f % CSW] + (unsigned)(f / CSW) * tap_bytes, 0, 0);
    }
    // the zero planes behind the last channel block (Cin not a multiple of 32) are never staged: cleared once, in both buffers
    {
        const int pad0 = p.C8in * p.plane, padn = (p.PK - p.C8in) * p.plane;
        const u32x4 zero = (u32x4){0u, 0u, 0u, 0u};
        for (int i = tid; i < padn; i += 256) {
            lds_in[pad0 + i] = zero;
            lds_in[buf_elems + pad0 + i] = zero;
        }
    }

    // ---- tile-independent pixel addressing
    unsigned b_addr[PS];   // byte address (buffer 0) of the pixel's window origin in plane lq
    unsigned pix_rel[PS];  // output byte offset relative to (image, first row of the tile); kInv = padding lane
    int y_rel[PS];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
        const unsigned pl0 = (unsigned)((wp_i * PS + ps) * 16 + lr);
        const bool in_tile = pl0 < (unsigned)p.RWo;
        const unsigned pl = in_tile ? pl0 : 0u;
        const unsigned y = fastdiv(pl, p.W, p.magic_wo);
        const unsigned xx = pl - y * p.W;
        b_addr[ps] = (unsigned)(lq * p.plane + y * P + xx) * 16u;
        pix_rel[ps] = in_tile ? (y * p.W + xx) * 16u : kInv;
        y_rel[ps] = (int)y;
    }
    // folded-BatchNorm scale / shift of the workgroup's couts: in LDS behind the ring ([CT] scale, [CT] shift), fetched by the
    // epilogue item that needs them (registers are what this kernel is short of)
    float* __restrict__ lds_bn = reinterpret_cast<float*>(smem16 + 2 * buf_elems);
    if (tid < CT) {
        lds_bn[tid] = p.scale[ct * CT + tid];  // Cout_pad16 is a multiple of CT: in range
        lds_bn[CT + tid] = p.shift[ct * CT + tid];
    }
    unsigned co_off[CSW], bn_off[CSW];
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs) {
        const int cl = wc_i * CSW * 16 + f16_d_cout<CSW>(cs, lq);
        const int co = ct * CT + cl;
        const bool ok = co < p.C8out * 8;
        bn_off[cs] = (unsigned)cl;
        co_off[cs] = ok ? (unsigned)(co >> 3) * plane_bytes + ((co >> 2) & 1) * 8u : kInv;
    }
    const size_t o_bytes = (size_t)p.N * p.C8out * plane_bytes;
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, o_bytes);
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.res1 ? p.res1 : p.out, p.res1 ? o_bytes : 0);  // absent: zero-length, nothing fetched

    f32x4 st_a[STATS ? CSW : 1], st_b[STATS ? CSW : 1];
    if constexpr (STATS) {
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) st_a[cs] = st_b[cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_z = make_rsrc(STATS == 2 ? p.st_z : p.out, STATS == 2 ? o_bytes : 0);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(STATS == 2 && p.st_y ? p.st_y : p.out, STATS == 2 && p.st_y ? o_bytes : 0);

    // ---- the SIDE JOBS of a tile: everything but its MFMAs.  A wave is alone on its SIMD, so whatever it issues between two tiles'
    //      MFMA loops leaves the matrix pipe idle (round-4 stamps, 48 -> 48 @96x72: 4.3 k cycles of MFMA per tile, 1.0 k issuing the
    //      next tile's DMA pieces, 2.0 k of epilogue VALU work).  What fits in an MFMA's shadow is small: the MFMA holds the SIMD's
    //      vector issue for 8 of its 16 cycles, a plain VALU instruction costs 4, a scalar one ~6, and costs ADD - two VALU
    //      instructions per MFMA are free, a third is paid in full (tools/probes/mfma_shadow.hip; MI355X_MICROARCH constants table).
    //      The jobs are therefore cut into MICRO-OPS of at most two vector instructions - a DMA piece, a residual load, one
    //      pk_fma pair, two conversions, two ReLUs, a packed rounding, a store - and micro-op u of tile t - 1 (DMA: of tile t + 1)
    //      sits behind MFMA number u M / NU of tile t.  The accumulators live in AGPRs and are copied to VGPRs at the tile's end.
    //      List order = issue order of the vector-memory operations: DMA pieces, then loads, then stores - at the top of the next
    //      tile everything older than the NST stores has landed (counted vmcnt).
    constexpr int NDMA = NQ * kWsMaxPieces;              // DMA slots (pieces beyond the plane's size are skipped)
    constexpr int NLD = NST * (STATS == 2 ? 3 : 1);      // residual (+ z, y) loads
    constexpr int UH = RES ? 7 : 4;                      // micro-ops of one half (4 couts of one pixel): fma | cvt cvt add | relu relu | pack
    constexpr int UP = 2 * UH + 1, US = UH + 1;          // ... of a paired / single output group (+ the store)
    constexpr int NEPI = PS * (NP * UP + NS * US);
    constexpr int M = NQ * T * PS * CSW;                 // MFMAs per tile

    u32x4 r1p[NP ? NP : 1][PS];
    u32x2 r1s[PS];
    u32x4 zp[(STATS == 2 && NP) ? NP : 1][PS], yp[(STATS == 2 && NP) ? NP : 1][PS];
    u32x2 zs[PS], ys[PS];
    u32x2 e_lo, e_hi;  // the output group in flight (micro-ops of a group are consecutive)
    f32x4 e_v;
    f32x2 e_r01, e_r23;
    unsigned pix_prev[PS];  // output offsets of the tile whose side jobs are running; kInv: none (loads return zero, stores drop)
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) pix_prev[ps] = kInv;
    const float relu_floor = p.relu ? 0.f : -__builtin_inff();  // max(v, floor): the ReLU without a branch

    // scale / shift registers: `lo` serves the even cout tile of a pair (and the single last tile), `hi` the odd one.  Each is
    // reloaded from LDS right behind the LAST micro-op that used its value (the groups of one cout tile are consecutive), i.e. many
    // MFMAs ahead of the next use: the epilogue never waits on LDS.  A sequence of one value is never reloaded
    constexpr int L_LO = NP + NS, L_HI = NP;
    f32x4 sc_lo, sh_lo, sc_hi, sh_hi;
    auto bn_load_lo = [&](int i) __attribute__((always_inline)) {
        const int cs = i < NP ? 2 * i : CSW - 1;
        sc_lo = *reinterpret_cast<const f32x4*>(lds_bn + bn_off[cs]);
        sh_lo = *reinterpret_cast<const f32x4*>(lds_bn + CT + bn_off[cs]);
    };
    auto bn_load_hi = [&](int i) __attribute__((always_inline)) {
        sc_hi = *reinterpret_cast<const f32x4*>(lds_bn + bn_off[2 * i + 1]);
        sh_hi = *reinterpret_cast<const f32x4*>(lds_bn + CT + bn_off[2 * i + 1]);
    };
    // one load: l < NP PS pairs (16 B) then NS PS singles (8 B); STATS == 2: then the same for z, then for y
    auto side_load = [&](int l) __attribute__((always_inline)) {
        const int which = l / NST, k = l - which * NST;
        const bool pair = k < NP * PS;
        const int j = pair ? k / PS : 0, ps = pair ? k - j * PS : k - NP * PS;
        const unsigned off = co_off[pair ? 2 * j : CSW - 1] + pix_prev[ps];
        if (which == 0) {
            if constexpr (RES) {
                if (pair) r1p[j][ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, off, 0, 0);
                else r1s[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs_r, off, 0, 0);
            }
        } else if (which == 1) {
            if (pair) zp[j][ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, off, 0, 0);
            else zs[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs_z, off, 0, 0);
        } else {
            if (pair) yp[j][ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, off, 0, 0);
            else ys[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs_y, off, 0, 0);
        }
    };
    // micro-op k of one half: 4 couts of one pixel, accumulator `a`, residual halves `r`; the arithmetic of f16_epi4 / f16_pack4
    // (fma(acc, scale, shift), + residual, ReLU, one rounding) instruction by instruction
    auto half_op = [&](int k, const f32x4& a, const f32x4& sc, const f32x4& sh, unsigned r0, unsigned r1, u32x2& out) __attribute__((always_inline)) {
        const int step = RES ? k : (k == 0 ? 0 : k + 3);  // without a residual: fma, relu, relu, pack
        if (step == 0) {
            const f32x2 v01 = __builtin_elementwise_fma((f32x2){a.x, a.y}, (f32x2){sc.x, sc.y}, (f32x2){sh.x, sh.y});
            const f32x2 v23 = __builtin_elementwise_fma((f32x2){a.z, a.w}, (f32x2){sc.z, sc.w}, (f32x2){sh.z, sh.w});
            e_v = (f32x4){v01.x, v01.y, v23.x, v23.y};
        } else if (step == 1) {
            const f16x2 h = __builtin_bit_cast(f16x2, r0);
            e_r01 = (f32x2){(float)h.x, (float)h.y};
        } else if (step == 2) {
            const f16x2 h = __builtin_bit_cast(f16x2, r1);
            e_r23 = (f32x2){(float)h.x, (float)h.y};
        } else if (step == 3) {
            e_v = (f32x4){e_v.x + e_r01.x, e_v.y + e_r01.y, e_v.z + e_r23.x, e_v.w + e_r23.y};
        } else if (step == 4) {
            asm volatile("v_max_f32 %0, %1, %0" : "+v"(e_v.x) : "v"(relu_floor));
            asm volatile("v_max_f32 %0, %1, %0" : "+v"(e_v.y) : "v"(relu_floor));
        } else if (step == 5) {
            asm volatile("v_max_f32 %0, %1, %0" : "+v"(e_v.z) : "v"(relu_floor));
            asm volatile("v_max_f32 %0, %1, %0" : "+v"(e_v.w) : "v"(relu_floor));
        } else {
            out = f16_pack4(e_v);
        }
    };
    // micro-op e of the previous tile's epilogue (accumulators `ac`): paired groups (j, ps), then single groups (ps)
    auto side_epi = [&](int e, f32x4 (&ac)[PS][CSW]) __attribute__((always_inline)) {
        const bool pair = e < NP * PS * UP;
        const int e2 = pair ? e : e - NP * PS * UP;
        const int grp = e2 / (pair ? UP : US), sub = e2 - grp * (pair ? UP : US);
        const int j = pair ? grp / PS : 0, ps = pair ? grp - j * PS : grp;
        [[maybe_unused]] const bool valid = pix_prev[ps] != kInv;
        if ((MP_WS_ABLATE & 1) && sub < (pair ? 2 * UH : UH)) return;
        if ((MP_WS_ABLATE & 16) && sub >= (pair ? 2 * UH : UH)) return;
        if (pair) {
            if (sub < UH) {
                half_op(sub, ac[ps][2 * j], sc_lo, sh_lo, RES ? r1p[j][ps].x : 0u, RES ? r1p[j][ps].y : 0u, e_lo);
                if (sub == 0 && ps == PS - 1 && L_LO > 1) bn_load_lo((j + 1) % L_LO);
            } else if (sub < 2 * UH) {
                half_op(sub - UH, ac[ps][2 * j + 1], sc_hi, sh_hi, RES ? r1p[j][ps].z : 0u, RES ? r1p[j][ps].w : 0u, e_hi);
                if (sub == UH && ps == PS - 1 && L_HI > 1) bn_load_hi((j + 1) % L_HI);
            } else {
                if constexpr (STATS == 2) {
                    const u32x4 zq = zp[j][ps], yq = yp[j][ps];
                    f16_stats_acc<2>(e_lo, valid, st_a[2 * j], st_b[2 * j], (u32x2){zq.x, zq.y}, (u32x2){yq.x, yq.y}, p.st_relu);
                    f16_stats_acc<2>(e_hi, valid, st_a[2 * j + 1], st_b[2 * j + 1], (u32x2){zq.z, zq.w}, (u32x2){yq.z, yq.w}, p.st_relu);
                } else if constexpr (STATS == 1) {
                    f16_stats_acc<1>(e_lo, valid, st_a[2 * j], st_b[2 * j], e_lo, e_lo, 0);
                    f16_stats_acc<1>(e_hi, valid, st_a[2 * j + 1], st_b[2 * j + 1], e_hi, e_hi, 0);
                }
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){e_lo.x, e_lo.y, e_hi.x, e_hi.y}, rs_o, co_off[2 * j] + pix_prev[ps], 0, 0);
            }
        } else {
            if (sub < UH) {
                half_op(sub, ac[ps][CSW - 1], sc_lo, sh_lo, RES ? r1s[ps].x : 0u, RES ? r1s[ps].y : 0u, e_lo);
                if (sub == 0 && ps == PS - 1 && L_LO > 1) bn_load_lo(0);
            } else {
                if constexpr (STATS == 2) f16_stats_acc<2>(e_lo, valid, st_a[CSW - 1], st_b[CSW - 1], zs[ps], ys[ps], p.st_relu);
                else if constexpr (STATS == 1) f16_stats_acc<1>(e_lo, valid, st_a[CSW - 1], st_b[CSW - 1], e_lo, e_lo, 0);
                __builtin_amdgcn_raw_buffer_store_b64(e_lo, rs_o, co_off[CSW - 1] + pix_prev[ps], 0, 0);
            }
        }
    };

    WS_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first tile and the weight fragments (asm loads: this wait is theirs) have landed
    __syncthreads();
    WS_STAMP(2);
    bn_load_lo(0);
    if constexpr (NP > 0) bn_load_hi(0);

    // one tile: MFMAs into `acc` (AGPRs), side jobs of the previous tile from `prev` (its accumulators, copied to VGPRs at its end)
    f32x4 acc[PS][CSW], prev[PS][CSW];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps)
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) prev[ps][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};  // read (masked) by the first tile's side jobs
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        if (t != t_begin) {
            // tile t's DMA pieces and the residual loads (issued one tile ago) are OLDER than the NST stores that followed them
            __builtin_amdgcn_s_waitcnt(0x0F70 | (NST & 15) | ((NST >> 4) << 14));  // vmcnt(NST)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // ... in every wave, and nobody reads the other buffer any more
            asm volatile("" ::: "memory");
        }
        WS_STAMP(t - t_begin < 6 ? 3 + 2 * (t - t_begin) : 64);
        // DMA of tile t + 1: the descriptors of this wave's planes, once per tile and OUTSIDE the MFMA stream (scalar instructions
        // are the dearest fillers); a DMA micro-op is then m0, one address add, the load
        const bool more = t + 1 < t_end;
        int n1, y1;
        tile_pos(more ? t + 1 : t, n1, y1);
        const unsigned row_off1 = (unsigned)((y1 - 1) * p.W * 16);
        const char* img1 = reinterpret_cast<const char*>(p.x) + (size_t)n1 * p.C8in * plane_bytes;
        // absent plane / no next tile: zero-length descriptor (the piece is still issued: no branch on it in the MFMA stream; it
        // writes zeros into the padding planes, which is what they hold)
        auto plane_rsrc = [&](int j) __attribute__((always_inline)) {
            const int pl = wave + 4 * j;
            return make_rsrc(img1 + (size_t)(pl < p.C8in ? pl : 0) * plane_bytes, more && pl < p.C8in ? plane_bytes : 0);
        };
        // (no arrays of the descriptor type: k-steps beyond NQ repeat the first and are never used)
        const __amdgpu_buffer_rsrc_t rs_dma0 = plane_rsrc(0), rs_dma1 = plane_rsrc(NQ > 1 ? 1 : 0), rs_dma2 = plane_rsrc(NQ > 2 ? 2 : 0),
                                     rs_dma3 = plane_rsrc(NQ > 3 ? 3 : 0);
        u32x4* dst1 = lds_in + (cur ^ 1) * buf_elems;
        auto side_dma = [&](int w) __attribute__((always_inline)) {
            const int j = w / kWsMaxPieces, s = w - j * kWsMaxPieces;
            const int pl = wave + 4 * j;
            if (s < ppp)  // wave-uniform (scalar compare + branch); pl < PK = 4 NQ always
                __builtin_amdgcn_raw_ptr_buffer_load_lds(j == 0 ? rs_dma0 : j == 1 ? rs_dma1 : j == 2 ? rs_dma2 : rs_dma3, (__attribute__((address_space(3))) void*)(dst1 + pl * p.plane + s * 64), 16,
                                                         piece_rel[s] + row_off1, 0, 0, 0);
        };

        int n, y0;
        tile_pos(t, n, y0);
        const int rows_valid = min(p.R, p.H - y0);
        const unsigned tile_o = (unsigned)n * p.C8out * plane_bytes + (unsigned)(y0 * p.W * 16);

        // pixel operands: one address per (k-step, window row, pixel tile) - the window column is the instruction's immediate offset
        const char* lbase = reinterpret_cast<const char*>(lds_in + cur * buf_elems);
        const char* brow[PS];
        u32x4 bv[PS];
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            brow[ps] = lbase + b_addr[ps];
            bv[ps] = *reinterpret_cast<const u32x4*>(brow[ps]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int tp = 0; tp < T; ++tp) {
                // pixel operands of the NEXT tap are requested (one ds_read_b128 each) between this tap's MFMAs; the final prefetch
                // re-reads the first position (discarded).  The asm MFMAs are ordered statements: source order is issue order
                const bool first = q == 0 && tp == 0, last = q == NQ - 1 && tp == T - 1;
                WS_STAMP(t - t_begin == 1 ? 16 + q * T + tp : 64);
                const int qn = last ? 0 : (tp + 1 < T ? q : q + 1), tn = last ? 0 : (tp + 1 < T ? tp + 1 : 0);
                const unsigned off_row = (unsigned)(qn * 4 * p.plane + (tn / 3) * P) * 16u;
                u32x4 bn[PS];
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) {
                    if (tn % 3 == 0) brow[ps] = lbase + b_addr[ps] + off_row;  // a new window row (or k-step): one add
                    if (MP_WS_ABLATE & 8) bn[ps] = bv[ps];
                    else bn[ps] = *reinterpret_cast<const u32x4*>(brow[ps] + (tn % 3) * 16);
#pragma unroll
                    for (int cs = 0; cs < CSW; ++cs) {
                        const int f = (q * T + tp) * CSW + cs;
                        if (f < NFA) {
                            if (first) mfma_a0(acc[ps][cs], Aa[f < NFA ? f : 0], bv[ps]);
                            else mfma_a(acc[ps][cs], Aa[f < NFA ? f : 0], bv[ps]);
                        } else {
                            if (first) mfma_v0(acc[ps][cs], Av[f >= NFA ? f - NFA : 0], bv[ps]);
                            else mfma_v(acc[ps][cs], Av[f >= NFA ? f - NFA : 0], bv[ps]);
                        }
                        // micro-ops in the shadow of this MFMA.  FRONT half of the tile: the previous tile's residual loads, then the
                        // next tile's DMA pieces, spread evenly - every wave-wide 1 KB vector-memory instruction occupies the CU's
                        // 64 B/clk address path for 16 cycles, and four waves issuing them back to back stall each other on it
                        // (ablation: ~100 cycles of wave time per piece at one piece per two MFMAs).  BACK half: the epilogue
                        // micro-ops, evenly spread, each group ending in its store.  All stores are younger than all DMA pieces.
                        const int m = ((q * T + tp) * PS + ps) * CSW + cs;
                        constexpr int NFRONT = NLD + NDMA;
                        constexpr int E0 = NEPI >= M ? 0 : (M - NEPI < M / 2 ? M - NEPI : M / 2);  // first MFMA of the epilogue stretch
                        constexpr int FE = E0 > 0 ? E0 : M / 2;                                      // the front jobs end here
                        constexpr int EMAX = (NEPI + (M - E0) - 1) / (M - E0), FMAX = (NFRONT + FE - 1) / FE;
                        __builtin_amdgcn_sched_barrier(0);
                        if (m < FE) {
                            const int f0 = m * NFRONT / FE, f1 = (m + 1) * NFRONT / FE;
#pragma unroll
                            for (int i = 0; i < FMAX; ++i) {
                                const int f = f0 + i;
                                if (f >= f1) continue;
                                if (f < NLD) {
                                    if (!(MP_WS_ABLATE & 4)) side_load(f);
                                } else if (!(MP_WS_ABLATE & 2)) side_dma(f - NLD);
                            }
                        }
                        if (m >= E0) {
                            const int e0 = (m - E0) * NEPI / (M - E0), e1 = (m + 1 - E0) * NEPI / (M - E0);
#pragma unroll
                            for (int i = 0; i < EMAX; ++i) {
                                const int e = e0 + i;
                                if (e >= e1) continue;
                                side_epi(e, prev);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) bv[ps] = bn[ps];
            }
        }
        // this tile becomes the one whose side jobs run.  MFMA result -> VALU read: a 4-pass XDL op needs its passes to drain (hipcc
        // pads nothing behind an asm statement)
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
#pragma unroll
            for (int cs = 0; cs < CSW; ++cs)
                if (!(MP_WS_ABLATE & 32)) prev[ps][cs] = acc_read(acc[ps][cs]);
            pix_prev[ps] = y_rel[ps] < rows_valid ? tile_o + pix_rel[ps] : kInv;
        }
        WS_STAMP(t - t_begin < 6 ? 4 + 2 * (t - t_begin) : 64);
    }
    // the last tile's side jobs, alone
#pragma unroll
    for (int l = 0; l < NLD; ++l) side_load(l);
#pragma unroll
    for (int e = 0; e < NEPI; ++e) side_epi(e, prev);
    WS_STAMP(15);
    if constexpr (STATS)
        f16_stats_flush<CSW, WAVES_P, WAVES_C>(st_a, st_b, reinterpret_cast<float*>(smem16), p.st_part, p.st_nparts, grp, ct * CT, p.C8out,
                                               wp_i, wc_i, lq, lr);
}

// Builds that do not fit the register file (hipcc gives them scratch memory: 24 - 328 B per lane, measured with
// -Rpass-analysis=kernel-resource-usage) are NOT instantiated: the launch answers "unsupported" and the tuner takes another variant
// of the shape.  A spill is slow here (one wave per SIMD hides nothing) and it is the one place where a compiler-made register copy
// could meet a fragment that an asm load has not delivered yet.  csrc/Makefile fails the build if any instantiated
// conv_f16_ws_kernel reports ScratchSize != 0, so this list cannot go stale silently.
template <int NQ, int CSW, int WAVES_P, int PS, int OCC, int STATS, bool RES>
constexpr bool ws_build_fits() {
    if (NQ == 2 && CSW == 3 && PS == 5) return STATS != 2;                      // 48 ch: 54 fragments + 15 accumulators
    if ((NQ == 2 && CSW == 4) || (NQ == 3 && CSW == 3)) return STATS == 0 || (STATS == 1 && !RES);  // 72 / 81 fragments
    if (NQ == 4 && CSW == 2 && WAVES_P == 1) return STATS != 2 && !RES;         // 72 fragments, 6 pixel tiles
    return true;
}

template <int NQ, int CSW, int WAVES_P, int PS, int OCC, int STATS, bool RES>
int launch_ws_kernel(const ConvF16Params& p, size_t lds_bytes, hipStream_t s) {
    if constexpr (!ws_build_fits<NQ, CSW, WAVES_P, PS, OCC, STATS, RES>()) return MP_ERR_UNSUPPORTED;
    else {
    if (g_dry_launch) return MP_OK;  // mp_f16_conv_supported: the dispatch alone
    auto kern = conv_f16_ws_kernel<NQ, CSW, WAVES_P, PS, OCC, STATS, RES>;
    static AttrOnce attr_set_once;
    if (attr_set_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
#if MP_WS_STAMPS
    hipLaunchKernelGGL(kern, dim3(p.total_blocks), dim3(256), lds_bytes, s, p,
                       (g_ws_stamp_buf && (size_t)p.total_blocks * 64 * 8 <= g_ws_stamp_bytes) ? g_ws_stamp_buf : nullptr);
#else
    hipLaunchKernelGGL(kern, dim3(p.total_blocks), dim3(256), lds_bytes, s, p);
#endif
    return check_launch();
    }
}

template <int NQ, int CSW, int WAVES_P, int PS, int OCC>
int launch_ws(const ConvF16Params& p, size_t lds_bytes, hipStream_t s) {
    // the residual is a template parameter: its loads and the six vector instructions per 4 couts exist only where there is one
    const bool res = p.res1 != nullptr;
#if defined(MP_WS_QUICK) && MP_WS_QUICK  // kernel work: inference builds only (a quarter of the compile time)
    if (p.st_mode != 0) return MP_ERR_UNSUPPORTED;
#else
    if (p.st_mode == 1) return res ? launch_ws_kernel<NQ, CSW, WAVES_P, PS, OCC, 1, true>(p, lds_bytes, s) : launch_ws_kernel<NQ, CSW, WAVES_P, PS, OCC, 1, false>(p, lds_bytes, s);
    if (p.st_mode == 2) return res ? launch_ws_kernel<NQ, CSW, WAVES_P, PS, OCC, 2, true>(p, lds_bytes, s) : launch_ws_kernel<NQ, CSW, WAVES_P, PS, OCC, 2, false>(p, lds_bytes, s);
#endif
    return res ? launch_ws_kernel<NQ, CSW, WAVES_P, PS, OCC, 0, true>(p, lds_bytes, s) : launch_ws_kernel<NQ, CSW, WAVES_P, PS, OCC, 0, false>(p, lds_bytes, s);
}

struct WsShape { int nq, csw, waves_p, ps, occ; };
// F_WS_BASE + i.  nq = k-steps of 32 input channels (Cin in (32 (nq - 1), 32 nq]); a workgroup = 16 csw (4 / waves_p) couts x
// 16 ps waves_p pixels
constexpr WsShape kWsShapes[F_WS_COUNT] = {
    {1, 2, 4, 3, 2},  // 32 channels: 32 couts x 192 px (training convs of the 32-channel branch; inference runs them as fused blocks)
    {2, 3, 4, 5, 1},  // 48 (W48 branch 1): 48 couts x 320 px
    {2, 4, 4, 3, 1},  // 64: 64 couts x 192 px
    {2, 2, 4, 3, 1},  // 64: two cout slices of 32 x 192 px
    {3, 2, 4, 5, 1},  // 96 (W48 branch 2): three cout slices of 32 x 320 px
    {3, 3, 4, 3, 1},  // 96: two cout slices of 48 x 192 px
    {4, 2, 2, 3, 1},  // 128: two cout slices of 64 x 96 px
    {4, 2, 1, 6, 1},  // 128: 128 couts x 96 px
};

}  // namespace

void f16_ws_dims(int v, int& ps, int& csw, int& waves_p) {
    const WsShape& s = kWsShapes[v - F_WS_BASE];
    ps = s.ps; csw = s.csw; waves_p = s.waves_p;
}

// geometry: 3x3 stride-1 pad-1 convolutions with a plain output mapping whose rows tile the pixel tile
bool f16_configure_ws(const mp_conv_desc& d, int variant, ConvF16Launch& L) {
    const WsShape& sh = kWsShapes[variant - F_WS_BASE];
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad_top != 1 || d.pad_left != 1) return false;
    if (d.conv_h != d.h || d.conv_w != d.w || d.out_h != d.h || d.out_w != d.w || d.out_mul != 1 || d.out_off_y != 0 || d.out_off_x != 0) return false;
    if (d.cin <= 32 * (sh.nq - 1) || d.cin > 32 * sh.nq) return false;
    ConvF16Params& p = L.p;
    const int WC = 4 / sh.waves_p;
    const int PT = 16 * sh.ps * sh.waves_p, CT = 16 * sh.csw * WC;
    p.N = d.n; p.H = d.h; p.W = d.w; p.Cout = d.cout;
    p.C8in = (d.cin + 7) / 8;
    p.Cout_pad16 = round_up(d.cout, 16);
    p.C8out = (d.cout + 7) / 8;
    p.Ho = d.h; p.Wo = d.w; p.pad_t = 1; p.pad_l = 1;
    if ((long long)d.n * p.C8in * d.h * d.w * 16 >= 0x7FFFFFF0LL || (long long)d.n * p.C8out * d.h * d.w * 16 >= 0x60000000LL) return false;
    if (p.Cout_pad16 % CT != 0) return false;  // every wave of every workgroup owns CSW real cout tiles
    p.PK = sh.nq * 4;
    p.PKs = p.C8in;
    p.n_chunks = 1;
    if (p.W > PT) return false;
    p.n_ct = p.Cout_pad16 / CT;
    int groups = sh.occ * 256 / p.n_ct;  // workgroups per cout slice, all resident at once
    if (const char* e = knob("MP_F16_WS_GROUPS")) {  // tests: force long tile runs on small problems
        const int v = atoi(e);
        if (v >= 1) groups = v;
    }
    if (groups < 1) return false;
    const int r_max = PT / p.W < p.H ? PT / p.W : p.H;
    // rows per tile: the fewest tiles per workgroup win (every tile costs the same MFMA time, padding lanes included), then the taller
    int best_r = 0, best_tpw = 0;
    for (int R = r_max; R >= 1 && R >= r_max - 3; --R) {
        if (R * p.W * 10 < PT * 7) break;  // more than 30 % padding lanes
        const int plane = round_up((R + 2) * (p.W + 1) + 1, 64);
        if ((size_t)2 * p.PK * plane * 16 > (size_t)(sh.occ == 1 ? 156 : 78) * 1024) continue;
        if (plane / 64 > kWsMaxPieces) continue;
        const int tiles = ((p.H + R - 1) / R) * p.N;
        const int tpw = (tiles + groups - 1) / groups;
        if (best_r == 0 || tpw < best_tpw) { best_r = R; best_tpw = tpw; }
    }
    if (best_r == 0) return false;
    p.R = best_r;
    p.G = 1;
    p.RWo = p.R * p.W;
    p.Rin = p.R + 2;
    p.Wp = p.W + 1;
    p.img_plane = p.Rin * p.Wp;
    p.plane = round_up(p.img_plane + 1, 64);
    p.upc = p.plane / 64;  // DMA pieces per plane
    p.ncols = p.W;
    p.in_buf = p.PK * p.plane;
    p.w_buf = 0;
    p.nbuf = 2;
    p.tiles_y = (p.H + p.R - 1) / p.R;
    p.tiles_n = p.N;
    p.tiles_total = p.tiles_y * p.N;
    p.tiles_per_wg = (p.tiles_total + groups - 1) / groups;
    // a run of ONE tile amortises nothing (the weight fragments would be fetched per tile, as the one-tile kernels do with less LDS)
    if (p.tiles_per_wg < 2) return false;
    p.n_groups = (p.tiles_total + p.tiles_per_wg - 1) / p.tiles_per_wg;
    p.total_blocks = p.n_ct * p.n_groups;
    p.relu = d.relu;
    p.out_h = d.out_h; p.out_w = d.out_w; p.out_mul = 1; p.off_y = 0; p.off_x = 0;
    p.magic_ncols = magic_of((unsigned)p.Wp);      // slot -> row
    p.magic_wo = magic_of((unsigned)p.W);          // pixel -> row
    p.magic_rows = magic_of((unsigned)p.tiles_y);  // tile -> image
    p.magic_rin = p.magic_rwo = 0;
    p.ni_used = p.nw_used = 0;
    L.ks = 3; L.stride = 1; L.variant = variant;
    L.lds_bytes = (size_t)2 * p.in_buf * 16 + (size_t)2 * CT * 4;  // ring + scale / shift of the workgroup's couts
    return true;
}

int f16_ws_launch(const ConvF16Launch& L, hipStream_t s) {
    switch (L.variant - F_WS_BASE) {
        case 0: return launch_ws<1, 2, 4, 3, 2>(L.p, L.lds_bytes, s);
        case 1: return launch_ws<2, 3, 4, 5, 1>(L.p, L.lds_bytes, s);
        case 2: return launch_ws<2, 4, 4, 3, 1>(L.p, L.lds_bytes, s);
        case 3: return launch_ws<2, 2, 4, 3, 1>(L.p, L.lds_bytes, s);
        case 4: return launch_ws<3, 2, 4, 5, 1>(L.p, L.lds_bytes, s);
        case 5: return launch_ws<3, 3, 4, 3, 1>(L.p, L.lds_bytes, s);
        case 6: return launch_ws<4, 2, 2, 3, 1>(L.p, L.lds_bytes, s);
        case 7: return launch_ws<4, 2, 1, 6, 1>(L.p, L.lds_bytes, s);
        default: return MP_ERR_UNSUPPORTED;
    }
}

}  // namespace mp

#if MP_WS_STAMPS
extern "C" int mp_debug_set_ws_stamp_buffer(void* dev_ptr, size_t bytes) {  // diagnostic build only (tools/ws_probe.py)
    mp::g_ws_stamp_buf = reinterpret_cast<unsigned long long*>(dev_ptr);
    mp::g_ws_stamp_bytes = dev_ptr ? bytes : 0;
    return MP_OK;
}
#endif
