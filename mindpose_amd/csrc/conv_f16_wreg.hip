// fp16-MFMA convolution, "weights in registers" form, for the large-K stride-1 layers (HRNet branches with 64 - 384 channels).
//
// Why another structure (DESIGN 4.4, round-1 phase stamps): at fp16 MFMA rates the one-tile / multi-tile kernels are paced by
// everything AROUND the matrix pipe - both operands go global -> VGPR -> ds_write -> barrier -> ds_read, one barrier per 54 MFMAs,
// and a workgroup's life is a serial chain of those phases (pipe 25 % busy).  Here
//   * the weight operand never touches LDS: the four waves of a workgroup own disjoint output channels, so each wave streams ITS
//     A fragments straight from global memory (L2-resident, packed lane-linear) into a register ring one k-step (KS*KS taps)
//     ahead of their use - no staging registers, no ds_write, no sharing, no barrier;
//   * the input tile of ALL input channels is brought into LDS ONCE by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round trip;
//     rows / images outside the tensor and the halo column arrive as zeros through the buffer range check, so there is no zero
//     fill pass either), then read by every wave as the B operand: ONE barrier per workgroup;
//   * the k loop is therefore a pure ds_read_b128 + global A prefetch + MFMA stream in the SAME k order as the other kernels
//     (k-steps ascending, taps ascending): outputs are bit-identical to conv_f16_kernel / conv_f16_mt_kernel.
// LDS image: [plane][image][row][W + h] 16-byte elements + one trailing element, h = KS / 2: ONE zero column between rows serves
// as the right halo of row r and the left halo of row r + 1 (pixel (r, x) at r * (W + h) + x + h).
// Per-CU budget at (6 pixel tiles x 2 cout tiles per wave): LDS reads 128 B/clk of 256, weight stream 43 B/clk of the 64 B/clk
// vector-memory path, 12 MFMAs per tap per wave.
#include <type_traits>

#include "conv_f16.h"
#include "conv_f16_dev.h"

namespace mp {

#if defined(MP_WS_STAMPS) && MP_WS_STAMPS
extern unsigned long long* g_ws_stamp_buf;  // conv_f16_ws.hip (diagnostic build; tools/ws_probe.py reads both kernels' stamps)
extern size_t g_ws_stamp_bytes;
#endif

namespace {

constexpr int kWregMaxPieces = 8;  // DMA pieces (64 x 16 B) per plane of the fast staging form

#ifndef MP_WS_STAMPS
#define MP_WS_STAMPS 0
#endif
#if MP_WS_STAMPS
#define WREG_STAMP(i)                                                                        \
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
#define WREG_STAMP(i) do { } while (0)
#endif

// Wave roles: WAVES_P waves split the workgroup's pixel tiles, 4 / WAVES_P split its couts.  WAVES_P = 1 (every wave all pixels,
// a quarter of the couts) streams each weight byte once per workgroup - the shape for the 128 - 384-channel layers whose weight
// stream is what the vector-memory path carries; with fewer couts (64 / 32-channel layers) the pixel split keeps two cout tiles
// per wave, i.e. half the LDS reads per MFMA, at the price of WAVES_P waves fetching the same (small) weight fragments.
// PRE (training, forward statistics builds of the 3x3 stride-1 form): the tensor at p.x is the RAW output z of the conv below; each
// wave turns the planes it staged into act(z * pre_scale + pre_shift) in place - the BatchNorm apply pass of the layer below, done
// on this conv's operand (hrnet.py:67-72) - and writes the tile's own rows to p.pre_out for the backward pass.
template <int KS, int S, int PS, int CSW, int WAVES_P, int OCC, int STATS = 0, bool PRE = false>
__global__ __launch_bounds__(256, OCC) void conv_f16_wreg_kernel(const ConvF16Params p
#if MP_WS_STAMPS
                                                                 , unsigned long long* dbg
#endif
) {
    constexpr int T = KS * KS;
    constexpr int HALO = KS / 2;
    constexpr bool RES2 = S == 2;  // the stride-2 layers are the exchange unit's down-sampling convs: running sum + identity
    constexpr int WAVES_C = 4 / WAVES_P;
    constexpr int CT = 16 * CSW * WAVES_C;  // couts per workgroup
    extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
    u32x4* __restrict__ lds_in = smem16;  // [PK][plane]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp_i = wave % WAVES_P, wc_i = wave / WAVES_P;
    const int lq = lane >> 4, lr = lane & 15;
    WREG_STAMP(0);

    int b = blockIdx.x;
    {
        const int nb = p.total_blocks, q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, j = b >> 3;
        b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const int ct = b % p.n_ct;
    b /= p.n_ct;
    const int part_idx = b;  // pixel tile = partial-sum slot of the epilogue statistics
    const int ty = b % p.tiles_y, tn = b / p.tiles_y;
    const int n0 = tn * p.G, y0 = ty * p.R;
    const int HW = p.H * p.W;
    const int P = p.Wp;  // row pitch W + HALO
    const int y_in0 = y0 * S - HALO;  // image row of the tile's first staged row

    // ---- input tile: every 16-byte slot of the LDS image is written by LDS-DMA, data or (out of range) zero.  The DMA goes FIRST:
    //      its data comes from HBM and ALL of it must have landed before the first MFMA, the weight fragments behind it come from L2
    //      and are needed tap by tap
    [[maybe_unused]] unsigned piece_rel[kWregMaxPieces];  // fast form: byte offset of a piece's slot inside its image plane; kOob = zero slot
    if (p.upc > 0) {
        // fast form (stride 1, one image per tile, plane pitch a multiple of 64 elements - f16_configure_wreg): a DMA piece never
        // straddles planes, so its slot -> (row, column) map is the same for every plane and is decoded ONCE (the magic
        // divisions are quarter-rate integer multiplies: ~300 cycles per piece in the generic loop below, round-4 stamps); a
        // plane only adds its descriptor - rows above / below the image fall outside it and arrive as zeros
        const unsigned plane_bytes = (unsigned)HW * 16u;
#pragma unroll
        for (int s = 0; s < kWregMaxPieces; ++s) {
            const unsigned slot = (unsigned)(s * 64 + lane);
            const unsigned r = __umulhi(slot, p.magic_ncols);
            const int c = (int)(slot - r * P) - HALO;
            piece_rel[s] = (slot < (unsigned)p.img_plane && c >= 0) ? (unsigned)(((int)r + y_in0) * p.W + c) * 16u : kOob;
        }
        const char* img = reinterpret_cast<const char*>(p.x) + (size_t)n0 * p.C8in * plane_bytes;
        for (int pl = wave; pl < p.PK; pl += 4) {  // wave-uniform
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(img + (size_t)(pl < p.C8in ? pl : 0) * plane_bytes, pl < p.C8in ? plane_bytes : 0);
#pragma unroll
            for (int s = 0; s < kWregMaxPieces; ++s) {
                if (s >= p.upc) break;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds_in + pl * p.plane + s * 64), 16,
                                                         piece_rel[s], 0, 0, 0);
            }
        }
    } else {
        const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, (size_t)p.N * p.C8in * HW * 16);
        const int total = p.PK * p.plane;  // a last partial piece spills (zeros) into the 64-element slack behind the image
        const int img_slots = p.G * p.img_plane;
        for (int s0 = wave * 64; s0 < total; s0 += 256) {
            const unsigned s = (unsigned)(s0 + lane);
            const unsigned pl = fastdiv(s, p.plane, p.magic_upc);
            const unsigned rem = s - pl * p.plane;
            const unsigned g = p.G > 1 ? fastdiv(rem, p.img_plane, p.magic_rin) : 0u;
            const unsigned rr = rem - g * p.img_plane;
            const unsigned r = fastdiv(rr, P, p.magic_ncols);
            const int c = (int)(rr - r * P) - HALO;
            const int yin = y_in0 + (int)r;
            const bool ok = rem < (unsigned)img_slots && c >= 0 && yin >= 0 && yin < p.H && n0 + (int)g < p.N && pl < (unsigned)p.C8in;
            const unsigned off = ok ? ((((unsigned)(n0 + g) * p.C8in + pl) * HW + yin * p.W + c) * 16u) : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(lds_in + s0), 16, off, 0, 0, 0);
        }
    }
    WREG_STAMP(60);

    // ---- weight operand: this wave's couts, straight from the packed weights [kq][T][4][Cout_pad16] x 16 B
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.wp, (size_t)(p.PK >> 2) * T * 4 * p.Cout_pad16 * 16);
    unsigned a_goff[CSW];
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs) {
        const int co = ct * CT + wc_i * CSW * 16 + f16_a_row<CSW>(cs, lr);
        a_goff[cs] = co < p.Cout_pad16 ? (unsigned)(lq * p.Cout_pad16 + co) * 16u : kOob;
    }
    const unsigned tap_bytes = 4u * p.Cout_pad16 * 16u;  // one (k-step, tap): 4 planes x Cout_pad16 x 16 B
    u32x4 A[T][CSW];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs)
            A[t][cs] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_goff[cs] == kOob ? kOob : a_goff[cs] + t * tap_bytes, 0, 0);
    WREG_STAMP(61);
    // ---- pixel operand addresses and the epilogue's pixel offsets (overlap the DMA's flight)
    int b_off[PS];
    unsigned pix_off[PS];
    const int plane_o = p.out_h * p.out_w;
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
        const unsigned pl0 = (unsigned)((wp_i * PS + ps) * 16 + lr);
        const bool in_tile = pl0 < (unsigned)(p.G * p.RWo);
        const unsigned pl = in_tile ? pl0 : 0u;
        const unsigned g = fastdiv(pl, p.RWo, p.magic_rwo);
        const unsigned rem = pl - g * p.RWo;
        const unsigned y = fastdiv(rem, p.Wo, p.magic_wo);
        const unsigned xx = rem - y * p.Wo;
        b_off[ps] = lq * p.plane + g * p.img_plane + (y * S) * P + xx * S;
        const int yy = y0 + (int)y;
        const bool ok = in_tile && n0 + (int)g < p.N && yy < p.Ho;
        pix_off[ps] = ok ? (((unsigned)(n0 + g) * p.C8out * plane_o + (yy * p.out_mul + p.off_y) * p.out_w + xx * p.out_mul + p.off_x) * 16u) : kInv;
    }
    f32x4 sc[CSW], sh[CSW];
    unsigned co_off[CSW];
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs) {
        const int co = ct * CT + wc_i * CSW * 16 + f16_d_cout<CSW>(cs, lq);
        const bool ok = co < p.C8out * 8;
        const int cc = co < p.Cout_pad16 ? co : 0;
        sc[cs] = *reinterpret_cast<const f32x4*>(p.scale + cc);
        sh[cs] = *reinterpret_cast<const f32x4*>(p.shift + cc);
        co_off[cs] = ok ? (unsigned)(co >> 3) * plane_o * 16u + ((co >> 2) & 1) * 8u : kInv;
    }

    f32x4 acc[PS][CSW];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps)
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) acc[ps][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};

    WREG_STAMP(1);
    // this wave's DMA pieces have landed: they are OLDER than the T x CSW weight loads of the first k-step (and whatever else was
    // issued since), so all but that many may still be in flight - the fragments are waited for tap by tap (the compiler counts
    // them); more than 63 cannot be encoded: wait for the excess
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((T * CSW < 63 ? T * CSW : 63) & 15) | (((T * CSW < 63 ? T * CSW : 63) >> 4) << 14));
    if constexpr (PRE) {
        // The planes THIS wave staged are complete in LDS (its own vmcnt): sweep them in place, one 16-byte element (8 channels of a
        // pixel) per lane and piece - the arithmetic of bn16_apply_pre_body (fp32 fma, ReLU, one rounding).  Slots outside the
        // image (halo column, rows above / below, padding planes) hold the DMA's zeros and keep them: the conv's zero padding applies
        // to the ACTIVATION.  The rows [y0, y0 + R) belong to this tile alone: cout slice 0 writes them out.
        const unsigned plane_bytes = (unsigned)HW * 16u;
        const unsigned own_lo = (unsigned)(y0 * p.W) * 16u, own_hi = (unsigned)(min(y0 + p.R, p.H) * p.W) * 16u;
        char* yimg = reinterpret_cast<char*>(p.pre_out) + (size_t)n0 * p.C8in * plane_bytes;
        const float floor_v = p.pre_relu ? 0.f : -__builtin_inff();
        for (int pl = wave; pl < p.C8in; pl += 4) {  // wave-uniform: scale / shift through scalar loads
            float psc[8], psh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                psc[j] = p.pre_scale[pl * 8 + j];
                psh[j] = p.pre_shift[pl * 8 + j];
            }
            const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(yimg + (size_t)pl * plane_bytes, (p.pre_out && ct == 0) ? plane_bytes : 0);
#pragma unroll
            for (int s = 0; s < kWregMaxPieces; ++s) {
                if (s >= p.upc) break;
                u32x4* slot = lds_in + pl * p.plane + s * 64 + lane;
                const unsigned rel = piece_rel[s];
                const f16x8 zv = __builtin_bit_cast(f16x8, *slot);
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (_Float16)fmaxf(__builtin_fmaf((float)zv[j], psc[j], psh[j]), floor_v);
                const u32x4 yv = rel < plane_bytes ? __builtin_bit_cast(u32x4, o) : (u32x4){0u, 0u, 0u, 0u};
                *slot = yv;
                __builtin_amdgcn_raw_buffer_store_b128(yv, rs_y, (rel >= own_lo && rel < own_hi) ? rel : kOob, 0, 0);
            }
        }
    }
    __syncthreads();                                  // ... and every other wave's: the only barrier of the workgroup
    WREG_STAMP(2);

    constexpr int NP = CSW / 2, NS = CSW - 2 * NP;
    const size_t o_bytes = (size_t)p.N * p.C8out * plane_o * 16;
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, o_bytes);
    u32x4 r1p[NP ? NP : 1][PS], r2p[(RES2 && NP) ? NP : 1][PS];  // second residual tensor: stride-2 builds only
    u32x2 r1s[PS], r2s[PS];

    const int nq = p.PK >> 2;
    u32x4 bv[PS];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) bv[ps] = lds_in[b_off[ps]];
    // One k-step = T taps.  The body is STRAIGHT-LINE code (no branch around a load): hipcc's s_waitcnt insertion counts loads
    // exactly only then - with the refill under `if (more)` it fell back to vmcnt(0) at the top of every k-step, i.e. it waited
    // for the whole ring it had just refilled and the prefetch distance collapsed.  REFILL is a compile-time switch: the last
    // k-step is peeled (no refill; the residual tensor is fetched ahead of it instead).
    auto kstep = [&](int q, auto refill_tag) {
        constexpr bool REFILL = decltype(refill_tag)::value;
        const int in_q = q * 4 * p.plane;
        const int qn = REFILL ? q + 1 : q;  // the final prefetch re-reads a valid position (discarded)
        const unsigned wq_next = (unsigned)(q + 1) * T * tap_bytes;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int tn2 = (t + 1 < T) ? t + 1 : 0;
            const int in_off = ((t + 1 < T) ? in_q : qn * 4 * p.plane) + (tn2 / KS) * P + (tn2 % KS);
            u32x4 bn[PS];
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) bn[ps] = lds_in[b_off[ps] + in_off];
#pragma unroll
            for (int ps = 0; ps < PS; ++ps)
#pragma unroll
                for (int cs = 0; cs < CSW; ++cs)
                    acc[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[t][cs]),
                                                                          __builtin_bit_cast(f16x8, bv[ps]), acc[ps][cs], 0, 0, 0);
            {
                constexpr int NM = PS * CSW, NPAIR = PS < NM ? PS : NM;
#pragma unroll
                for (int i = 0; i < NPAIR; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                }
                if (NM > NPAIR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NPAIR, 0);
            }
            // this tap's weight registers are free: refill them for the next k-step (one k-step = T taps of prefetch distance)
            if constexpr (REFILL) {
#pragma unroll
                for (int cs = 0; cs < CSW; ++cs)
                    A[t][cs] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_goff[cs] == kOob ? kOob : a_goff[cs] + wq_next + t * tap_bytes, 0, 0);
            }
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) bv[ps] = bn[ps];
            // nothing crosses a tap boundary: left alone, the scheduler sinks all T x CSW refill loads to the end of the k-step
            // (shorter live ranges), where they are needed again at once - no prefetch distance left
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int q = 0; q + 1 < nq; ++q) {
        kstep(q, std::true_type{});
        WREG_STAMP(3 + q);
    }
    {
        // residual tensor: fetched ahead of the last k-step, so its latency hides under that step's MFMAs.  Unconditional loads
        // (an absent tensor is a zero-length descriptor: the range check answers, nothing is fetched) keep the code branch-free
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.res1 ? p.res1 : p.out, p.res1 ? o_bytes : 0);
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) r1p[j][ps] = __builtin_amdgcn_raw_buffer_load_b128(rs, co_off[2 * j] + pix_off[ps], 0, 0);
        if (NS) {
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) r1s[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs, co_off[CSW - 1] + pix_off[ps], 0, 0);
        }
        if constexpr (RES2) {
            const __amdgpu_buffer_rsrc_t rs2 = make_rsrc(p.res2 ? p.res2 : p.out, p.res2 ? o_bytes : 0);
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) r2p[j][ps] = __builtin_amdgcn_raw_buffer_load_b128(rs2, co_off[2 * j] + pix_off[ps], 0, 0);
            if (NS) {
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) r2s[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs2, co_off[CSW - 1] + pix_off[ps], 0, 0);
            }
        }
    }
    // epilogue statistics (conv_f16_dev.h), backward mode: the BatchNorm's z (and y), fetched with the residual
    f32x4 st_a[STATS ? CSW : 1], st_b[STATS ? CSW : 1];
    u32x4 zp[(STATS && NP) ? NP : 1][PS], yp[(STATS && NP) ? NP : 1][PS];
    u32x2 zs[PS], ys[PS];
    if constexpr (STATS) {
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) {
            st_a[cs] = st_b[cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (STATS == 2) {
            const __amdgpu_buffer_rsrc_t rz = make_rsrc(p.st_z, o_bytes);
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.st_y ? p.st_y : p.st_z, p.st_y ? o_bytes : 0);
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) {
                    zp[j][ps] = __builtin_amdgcn_raw_buffer_load_b128(rz, co_off[2 * j] + pix_off[ps], 0, 0);
                    yp[j][ps] = __builtin_amdgcn_raw_buffer_load_b128(ry, co_off[2 * j] + pix_off[ps], 0, 0);
                }
            if (NS) {
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) {
                    zs[ps] = __builtin_amdgcn_raw_buffer_load_b64(rz, co_off[CSW - 1] + pix_off[ps], 0, 0);
                    ys[ps] = __builtin_amdgcn_raw_buffer_load_b64(ry, co_off[CSW - 1] + pix_off[ps], 0, 0);
                }
            }
        }
    }
    WREG_STAMP(3 + nq - 1);
    kstep(nq - 1, std::false_type{});
    WREG_STAMP(3 + nq);

    // ---- epilogue: scale/shift, residuals, ReLU, one rounding, 16-byte stores per cout-tile pair
    const bool has1 = p.res1 != nullptr, has2 = RES2 && p.res2 != nullptr;
    const u32x2 none = (u32x2){0u, 0u};
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const u32x4 a1 = has1 ? r1p[j][ps] : (u32x4){0u, 0u, 0u, 0u};
            const u32x4 a2 = has2 ? r2p[j][ps] : (u32x4){0u, 0u, 0u, 0u};
            u32x2 lo = f16_pack4(f16_epi4(acc[ps][2 * j], sc[2 * j], sh[2 * j], has1, (u32x2){a1.x, a1.y}, has2, (u32x2){a2.x, a2.y}, p.relu));
            u32x2 hi = f16_pack4(f16_epi4(acc[ps][2 * j + 1], sc[2 * j + 1], sh[2 * j + 1], has1, (u32x2){a1.z, a1.w}, has2, (u32x2){a2.z, a2.w}, p.relu));
            if constexpr (STATS) {
                const bool valid = pix_off[ps] != kInv;
                if constexpr (STATS == 2) {
                    const u32x4 zq = zp[j][ps], yq = yp[j][ps];
                    f16_stats_acc<2>(lo, valid, st_a[2 * j], st_b[2 * j], (u32x2){zq.x, zq.y}, (u32x2){yq.x, yq.y}, p.st_relu);
                    f16_stats_acc<2>(hi, valid, st_a[2 * j + 1], st_b[2 * j + 1], (u32x2){zq.z, zq.w}, (u32x2){yq.z, yq.w}, p.st_relu);
                } else {
                    f16_stats_acc<1>(lo, valid, st_a[2 * j], st_b[2 * j], lo, lo, 0);
                    f16_stats_acc<1>(hi, valid, st_a[2 * j + 1], st_b[2 * j + 1], hi, hi, 0);
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){lo.x, lo.y, hi.x, hi.y}, rs_o, co_off[2 * j] + pix_off[ps], 0, 0);
        }
    if (NS) {
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            u32x2 o = f16_pack4(f16_epi4(acc[ps][CSW - 1], sc[CSW - 1], sh[CSW - 1], has1, has1 ? r1s[ps] : none, has2, has2 ? r2s[ps] : none, p.relu));
            if constexpr (STATS) {
                const bool valid = pix_off[ps] != kInv;
                if constexpr (STATS == 2) f16_stats_acc<2>(o, valid, st_a[CSW - 1], st_b[CSW - 1], zs[ps], ys[ps], p.st_relu);
                else f16_stats_acc<1>(o, valid, st_a[CSW - 1], st_b[CSW - 1], o, o, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b64(o, rs_o, co_off[CSW - 1] + pix_off[ps], 0, 0);
        }
    }
    WREG_STAMP(63);
    if constexpr (STATS)
        f16_stats_flush<CSW, WAVES_P, WAVES_C>(st_a, st_b, reinterpret_cast<float*>(smem16), p.st_part, p.st_nparts, part_idx, ct * CT,
                                               p.C8out, wp_i, wc_i, lq, lr);
}

template <int KS, int S, int PS, int CSW, int WAVES_P, int OCC, int STATS, bool PRE = false>
int launch_wreg_kernel(const ConvF16Params& p, size_t lds_bytes, hipStream_t s) {
    if (g_dry_launch) return MP_OK;  // mp_f16_conv_supported: the dispatch alone
    auto kern = conv_f16_wreg_kernel<KS, S, PS, CSW, WAVES_P, OCC, STATS, PRE>;
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

template <int KS, int S, int PS, int CSW, int WAVES_P, int OCC>
int launch_wreg(const ConvF16Params& p, size_t lds_bytes, hipStream_t s) {
    if (p.pre_scale) {  // BatchNorm apply on the input operand: mp_f16_conv2d_fwd_stats admits it for this form only
        if constexpr (KS == 3 && S == 1) {
            if (p.st_mode == 1 && p.upc > 0) return launch_wreg_kernel<KS, S, PS, CSW, WAVES_P, OCC, 1, true>(p, lds_bytes, s);
        }
        return MP_ERR_UNSUPPORTED;
    }
    if (p.st_mode == 1) return launch_wreg_kernel<KS, S, PS, CSW, WAVES_P, OCC, 1>(p, lds_bytes, s);  // training builds: epilogue statistics
    if (p.st_mode == 2) {
        if constexpr (S == 1) return launch_wreg_kernel<KS, S, PS, CSW, WAVES_P, OCC, 2>(p, lds_bytes, s);
        return MP_ERR_UNSUPPORTED;
    }
    return launch_wreg_kernel<KS, S, PS, CSW, WAVES_P, OCC, 0>(p, lds_bytes, s);
}

template <int KS, int S>
int launch_wreg_ks(const ConvF16Params& p, int variant, size_t lds_bytes, hipStream_t s) {
    switch (variant) {
        case F_WREG_P6C2: return launch_wreg<KS, S, 6, 2, 1, 1>(p, lds_bytes, s);
        case F_WREG_P3C2: return launch_wreg<KS, S, 3, 2, 1, 2>(p, lds_bytes, s);
        case F_WREG_P6C1: return launch_wreg<KS, S, 6, 1, 1, 2>(p, lds_bytes, s);
        case F_WREG_P6C3: return launch_wreg<KS, S, 6, 3, 1, 1>(p, lds_bytes, s);
        case F_WREG_P3C3: return launch_wreg<KS, S, 3, 3, 1, 1>(p, lds_bytes, s);
        case F_WREG_P3C4: return launch_wreg<KS, S, 3, 4, 1, 1>(p, lds_bytes, s);
        case F_WREG_P6C2_W2: return launch_wreg<KS, S, 6, 2, 2, 2>(p, lds_bytes, s);
        case F_WREG_P6C3_W2: return launch_wreg<KS, S, 6, 3, 2, 1>(p, lds_bytes, s);
        case F_WREG_P3C2_W2: return launch_wreg<KS, S, 3, 2, 2, 2>(p, lds_bytes, s);
        case F_WREG_P6C2_W4: return launch_wreg<KS, S, 6, 2, 4, 2>(p, lds_bytes, s);
        case F_WREG_P6C3_W4: return launch_wreg<KS, S, 6, 3, 4, 1>(p, lds_bytes, s);
        case F_WREG_P3C2_W4: return launch_wreg<KS, S, 3, 2, 4, 2>(p, lds_bytes, s);
        case F_WREG_P7C3:
            if constexpr (KS == 3 && S == 1) return launch_wreg<KS, S, 7, 3, 1, 1>(p, lds_bytes, s);
            return MP_ERR_UNSUPPORTED;
        case F_WREG_P4C3:
            if constexpr (KS == 3 && S == 1) return launch_wreg<KS, S, 4, 3, 1, 1>(p, lds_bytes, s);
            return MP_ERR_UNSUPPORTED;
        case F_WREG_P5C4:
            if constexpr (KS == 3 && S == 1) return launch_wreg<KS, S, 5, 4, 1, 1>(p, lds_bytes, s);
            return MP_ERR_UNSUPPORTED;
        default: return MP_ERR_UNSUPPORTED;
    }
}

}  // namespace

void f16_wreg_dims(int v, int& ps, int& csw, int& waves_p) {
    static const int pss[15] = {6, 3, 6, 6, 3, 3, 6, 6, 3, 6, 6, 3, 7, 4, 5};
    static const int css[15] = {2, 2, 1, 3, 3, 4, 2, 3, 2, 2, 3, 2, 3, 3, 4};
    static const int wps[15] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 4, 4, 4, 1, 1, 1};
    ps = pss[f16_wreg_index(v)];
    csw = css[f16_wreg_index(v)];
    waves_p = wps[f16_wreg_index(v)];
}

// geometry: 3x3 pad 1 (stride 1 or 2) and 1x1 pad 0 convolutions whose output rows tile the pixel tile
bool f16_configure_wreg(const mp_conv_desc& d, int variant, ConvF16Launch& L) {
    int PS, CSW, WP;
    f16_wreg_dims(variant, PS, CSW, WP);
    const int KS = d.kh, S = d.stride, halo = KS / 2, T = KS * KS;
    if (!((KS == 3 && (S == 1 || S == 2)) || (KS == 1 && S == 1))) return false;
    if (variant >= F_WREG_P7C3 && !(KS == 3 && S == 1)) return false;  // the round-4 shapes are built for the 3x3 stride-1 branch convs
    if (d.pad_top != halo || d.pad_left != halo) return false;
    if (d.conv_h != (d.h + 2 * halo - KS) / S + 1 || d.conv_w != (d.w + 2 * halo - KS) / S + 1) return false;
    ConvF16Params& p = L.p;
    const int WC = 4 / WP;
    const int PT = 16 * PS * WP, CT = 16 * CSW * WC;
    p.N = d.n; p.H = d.h; p.W = d.w; p.Cout = d.cout;
    p.C8in = (d.cin + 7) / 8;
    p.Cout_pad16 = round_up(d.cout, 16);
    p.C8out = (d.cout + 7) / 8;
    p.Ho = d.conv_h; p.Wo = d.conv_w; p.pad_t = halo; p.pad_l = halo;
    if ((long long)d.n * p.C8in * d.h * d.w * 16 >= 0x7FFFFFF0LL || (long long)d.n * p.C8out * d.out_h * d.out_w * 16 >= 0x60000000LL) return false;
    p.PK = round_up(d.cin, 32) / 8;
    p.PKs = p.C8in < p.PK ? p.C8in : p.PK;
    p.n_chunks = 1;
    if (p.Wo > PT) return false;
    // every wave of every workgroup owns CSW real cout tiles; all-pixels-per-wave shapes are for the layers with a substantial
    // weight stream (the small-K layers are HBM-bound: pixel-split shapes or the tile kernels)
    if ((p.Cout_pad16 / 16) % (WC * CSW) != 0) return false;
    if (WP == 1 && d.cin < 64) return false;
    int R = PT / p.Wo;
    if (R > p.Ho) R = p.Ho;
    p.R = R;
    p.G = 1;
    if (R == p.Ho) {
        p.G = PT / (p.Ho * p.Wo);
        if (p.G > p.N) p.G = p.N;
        if (p.G < 1) p.G = 1;
    }
    p.RWo = p.R * p.Wo;
    if (p.G * p.RWo * 10 < PT * 7) return false;  // more than 30 % padding lanes: another tile shape fits better
    p.Rin = (p.R - 1) * S + KS;
    p.Wp = p.W + halo;  // row pitch: one shared zero column
    p.img_plane = p.Rin * p.Wp;
    // plane stride: == 0 (mod 16) for stride 1, odd for stride 2 (lanes then read every other element): conflict-free ds_read_b128
    p.plane = S == 1 ? round_up(p.G * p.img_plane + halo, 16) : ((p.G * p.img_plane + halo) | 1);
    p.ncols = p.W;
    p.upc = 0;
    if (S == 1 && p.G == 1 && round_up(p.img_plane + halo, 64) / 64 <= 8 && !knob("MP_F16_WREG_SLOW_DMA")) {
        // fast staging form: plane pitch a multiple of 64 elements, upc = DMA pieces per plane
        p.plane = round_up(p.img_plane + halo, 64);
        p.upc = p.plane / 64;
    }
    p.in_buf = p.PK * p.plane;
    p.w_buf = 0;
    p.nbuf = 1;
    p.n_ct = (p.Cout_pad16 + CT - 1) / CT;
    p.tiles_y = (p.G > 1 || p.R >= p.Ho) ? 1 : (p.Ho + p.R - 1) / p.R;
    p.tiles_n = (p.N + p.G - 1) / p.G;
    p.tiles_total = p.tiles_y * p.tiles_n;
    p.relu = d.relu;
    p.out_h = d.out_h; p.out_w = d.out_w; p.out_mul = d.out_mul; p.off_y = d.out_off_y; p.off_x = d.out_off_x;
    p.magic_upc = magic_of(p.plane);       // slot -> plane
    p.magic_rin = magic_of(p.img_plane);   // slot -> image
    p.magic_ncols = magic_of(p.Wp);        // slot -> row
    p.magic_rwo = magic_of(p.RWo);
    p.magic_wo = magic_of(p.Wo);
    p.total_blocks = p.n_ct * p.tiles_y * p.tiles_n;
    p.ni_used = p.nw_used = 0;
    L.ks = KS; L.stride = S; L.variant = variant;
    L.lds_bytes = ((size_t)p.in_buf + 64) * 16;  // + one DMA piece of slack: the last piece may run past an odd-sized image
    (void)T;
    return L.lds_bytes <= (size_t)150 * 1024;
}

int f16_wreg_launch(const ConvF16Launch& L, hipStream_t s) {
    if (L.ks == 3) return L.stride == 1 ? launch_wreg_ks<3, 1>(L.p, L.variant, L.lds_bytes, s) : launch_wreg_ks<3, 2>(L.p, L.variant, L.lds_bytes, s);
    if (L.ks == 1 && L.stride == 1) return launch_wreg_ks<1, 1>(L.p, L.variant, L.lds_bytes, s);
    return MP_ERR_UNSUPPORTED;
}

}  // namespace mp
