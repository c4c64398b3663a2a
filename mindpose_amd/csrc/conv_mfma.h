// Direct (im2col-free) LDS-tiled convolution on the fp32 matrix cores of gfx950.
//
//   implicit GEMM:  M = output pixels (flattened (image, row, col) inside a tile),
//                   N = output channels, K = Cin x KS x KS walked as KS*KS shifted LDS views.
//   MFMA:           v_mfma_f32_16x16x4_f32 - exact fp32 FMA chain, 64 FLOP/clk/SIMD (157 TFLOP/s chip).
//                   lane l feeds A[pixel = l&15][cin = l>>4] and B[cin = l>>4][cout = l&15];
//                   D: lane holds 4 consecutive pixels ((l>>4)*4 + r) of one cout (l&15)
//                   -> the epilogue stores 16 B per lane straight into NCHW rows.
//   workgroup:      256 threads = 4 waves arranged WAVES_P x WAVES_C; a wave owns PS pixel
//                   sub-tiles x CS cout sub-tiles (16 x 16 each) = PS*CS accumulators.
//   LDS:            input tile  [CK cin][G images][Rin rows][Wp cols] incl. zero halo
//                   (cin plane stride == 16 mod 32 floats: the two cin planes a half-wave reads
//                   land on disjoint banks), weight tile [CK/4][KS*KS][4][CT] with a 16-column XOR
//                   swizzle when CT % 32 == 0 (rows kq, kq+1 of one half-wave on disjoint banks).
//   epilogue:       folded BatchNorm scale/shift, up to two residual tensors, ReLU, and an output
//                   mapping that covers plain conv, nearest-upsample-and-add (HRModule fuse rows,
//                   no up-sampled tensor is ever materialised) and the sub-pixel phases of the
//                   4x4 stride-2 transposed convolution.
#pragma once
#include <type_traits>

#include "common.h"

#ifndef MP_CONV_SCHED
#define MP_CONV_SCHED 1
#endif
#ifndef MP_CONV_STAMPS
#define MP_CONV_STAMPS 0  // 1: per-workgroup phase cycle counters into ConvKParams::dbg (never in the product build)
#endif
#if MP_CONV_STAMPS
#define MP_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#else
#define MP_STAMP(var) [[maybe_unused]] const unsigned long long var = 0
#endif

namespace mp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Raw buffer access: the hardware range check makes invalid lanes free - an offset >= num_records loads 0 /
// drops the store - so masked staging and epilogue traffic needs no branches (and no s_waitcnt between loads).
constexpr unsigned kOob = 0x80000000u;  // every descriptor below spans < 2 GiB
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes < 0x7FFFFFF0u ? bytes : 0x7FFFFFF0u), 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}

struct ConvKParams {
    const float* x;
    const float* wp;  // packed weights [Cin_pad4/4][T][4][Cout_pad16]
    const float* scale;
    const float* shift;
    const float* res1;
    const float* res2;
    float* out;
    int N, Cin, H, W;
    int Cout, Cout_pad16, Cin_pad4;
    int Ho, Wo;  // conv output extent
    int pad_t, pad_l;
    int R, G;          // tile = G images x R conv-output rows x Wo columns
    int Rin, Wp;       // LDS rows / pitch per image
    int img_plane;     // Rin * Wp
    int cin_plane;     // >= G * img_plane, == 16 (mod 32)
    int CK;            // cin per chunk (multiple of 4, divides Cin_pad4)
    int n_chunks;
    int n_ct;          // cout tiles
    int tiles_y;       // row bands per image (1 when G > 1)
    int tiles_n;       // image groups
    int ncols;         // input columns copied per row
    int vec;           // 1: staging unit = 4 consecutive columns (float4 global load)
    int upr;           // staging units per input row (ncols / 4 or ncols)
    int upc;           // staging units per input channel (G * Rin * upr)
    int in_buf;        // floats per input buffer  (CK * cin_plane)
    int w_buf;         // floats per weight buffer (CK * T * CT)
    int nbuf;          // 2 = double-buffered chunks, 1 = single chunk
    int out_h, out_w, out_mul, out_rep, off_y, off_x;
    int relu;
    unsigned magic_upr, magic_upc, magic_rin, magic_rwo, magic_wo;  // fast-division multipliers
    int RWo;           // R * Wo
    int total_blocks;
    unsigned long long* dbg;  // diagnostic builds (MP_CONV_STAMPS) only: 8 x u64 per workgroup
};

__device__ __forceinline__ unsigned fastdiv(unsigned e, unsigned d, unsigned magic) {
    // magic = floor(2^32 / d) + 1; exact while e * d < 2^32 (all uses here: e, d < 2^16)
    return d == 1 ? e : __umulhi(e, magic);
}

// Software-pipelined workgroup:
//   prologue   zero both input buffers (halo), build the per-thread staging tables, stage chunk 0
//   per chunk  issue the global loads of chunk c+1 into registers  ->  MFMA over chunk c (operand
//              fragments prefetched one k-step ahead, so an MFMA never waits on LDS)  ->  write the
//              registers into the other LDS buffer  ->  ONE barrier
// NI / NW = staging units (16 B) per thread per chunk for input / weights; the host picks CK to fit.
template <int KS, int S, int PS, int CS, int WAVES_P, int WAVES_C, int NI, int NW, bool VEC, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_mfma_kernel(const ConvKParams p) {
    static_assert(WAVES_P * WAVES_C == 4, "4 waves per workgroup");
    constexpr int T = KS * KS;
    constexpr int CT = 16 * CS * WAVES_C;
    constexpr int C4 = CT / 4;
    constexpr bool SWZ = (CT % 32) == 0;
    constexpr bool SCHED = MP_CONV_SCHED != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ lds_in = smem;                       // [nbuf][in_buf]
    float* __restrict__ lds_w = smem + p.nbuf * p.in_buf;    // [nbuf][w_buf]

    MP_STAMP(t_start);
    [[maybe_unused]] unsigned long long s_load = 0, s_comp = 0, s_store = 0, s_bar = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp_i = wave % WAVES_P, wc_i = wave / WAVES_P;
    const int lq = lane >> 4, lr = lane & 15;

    // XCD-aware tile id: blocks b and b+8 share an XCD (round-robin dispatch), so give every XCD a
    // contiguous run of tiles; cout tiles of one pixel tile (same input) then share one L2.
    int b = blockIdx.x;
    {
        const int nb = p.total_blocks, q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, j = b >> 3;
        b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const int ct = b % p.n_ct;
    b /= p.n_ct;
    const int ty = b % p.tiles_y, tn = b / p.tiles_y;
    const int n0 = tn * p.G, y0 = ty * p.R;
    const int y_in0 = y0 * S - p.pad_t;
    const int HW = p.H * p.W;

    // zero the input buffers once: halo positions are never overwritten by the chunk copies
    {
        const int n4 = (p.nbuf * p.in_buf) >> 2;
        float4* z = reinterpret_cast<float4*>(lds_in);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid; i < n4; i += 256) z[i] = zero;
    }

    // per-thread staging tables (same for every chunk): source offset relative to the chunk base,
    // destination offset inside the input buffer with the chunk-local cin in the top bits
    unsigned isrc[NI];  // byte offset from the image-group base, kOob when the unit is outside the image
    int idst[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const unsigned u = tid + 256 * i;
        isrc[i] = kOob;
        idst[i] = 0;
        if (u < (unsigned)(p.CK * p.upc)) {
            const unsigned c = fastdiv(u, p.upc, p.magic_upc);
            const unsigned rem = u - c * p.upc;
            const unsigned gr = fastdiv(rem, p.upr, p.magic_upr);
            const unsigned xu = rem - gr * p.upr;
            const unsigned g = p.G > 1 ? fastdiv(gr, p.Rin, p.magic_rin) : 0u;
            const unsigned r = gr - g * p.Rin;
            const int yin = y_in0 + (int)r;
            const unsigned xx = VEC ? xu * 4 : xu;
            if (yin >= 0 && yin < p.H && n0 + (int)g < p.N) {
                isrc[i] = ((g * p.Cin + c) * HW + yin * p.W + xx) * 4u;
                idst[i] = (int)((c << 20) | (c * p.cin_plane + g * p.img_plane + r * p.Wp + p.pad_l + xx));
            }
        }
    }

    // lane-constant LDS offsets of the MFMA operands
    int a_off[PS];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
        unsigned pl = (unsigned)((wp_i * PS + ps) * 16 + lr);
        if (pl >= (unsigned)(p.G * p.RWo)) pl = 0;  // padding lanes read a valid address, result discarded
        const unsigned g = fastdiv(pl, p.RWo, p.magic_rwo);
        const unsigned rem = pl - g * p.RWo;
        const unsigned y = fastdiv(rem, p.Wo, p.magic_wo);
        const unsigned xx = rem - y * p.Wo;
        a_off[ps] = lq * p.cin_plane + g * p.img_plane + (y * S) * p.Wp + xx * S;
    }
    int b_off[CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        int col = (wc_i * CS + cs) * 16 + lr;
        if (SWZ) col ^= (lq & 1) << 4;
        b_off[cs] = lq * CT + col;
    }

    f32x4 acc[PS][CS];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) acc[ps][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* __restrict__ xg = p.x + (size_t)n0 * p.Cin * HW;
    const float* __restrict__ wg = p.wp + ct * CT;
    const int w_rows = (p.CK >> 2) * T * 4;  // weight rows (of CT floats) per chunk

    using in_t = typename std::conditional<VEC, f32x4, float>::type;
    in_t vin[NI];
    f32x4 vw[NW];
    const int n_img = min(p.G, p.N - n0);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(xg, (size_t)n_img * p.Cin * HW * 4);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(wg, ((size_t)(p.Cin_pad4 >> 2) * T * 4 * p.Cout_pad16 - (size_t)ct * CT) * 4);
    unsigned wsrc[NW];  // byte offset inside one weight chunk, kOob beyond the chunk / the padded couts
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int u = tid + 256 * i;
        const int row = u / C4, c4 = (u - row * C4) << 2;
        wsrc[i] = (row < w_rows && ct * CT + c4 < p.Cout_pad16) ? (unsigned)(row * p.Cout_pad16 + c4) * 4u : kOob;
    }
    // All NI + NW loads issue back to back (range-checked buffer loads, no branches) and stay in flight
    // across the MFMA loop; the first use is stage_store after the loop.
    auto stage_load = [&](int ch) {
        const int c0 = ch * p.CK;
        const unsigned xo = (unsigned)c0 * HW * 4u;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            unsigned off = isrc[i] + xo;
            if (p.Cin & 3) off = (c0 + (idst[i] >> 20) < p.Cin) ? off : kOob;  // zero-padded cin tail
            if constexpr (VEC) vin[i] = buf_load4(rs_x, off);
            else vin[i] = buf_load1(rs_x, off);
        }
        const unsigned wo = (unsigned)(c0 >> 2) * T * 4u * p.Cout_pad16 * 4u;
#pragma unroll
        for (int i = 0; i < NW; ++i) vw[i] = buf_load4(rs_w, wsrc[i] + wo);
    };
    auto stage_store = [&](int ch, int buf) {
        const int c0 = ch * p.CK;
        float* __restrict__ din = lds_in + buf * p.in_buf;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (isrc[i] != kOob && c0 + (idst[i] >> 20) < p.Cin) {
                float* d = din + (idst[i] & 0xFFFFF);
                if constexpr (VEC) { d[0] = vin[i].x; d[1] = vin[i].y; d[2] = vin[i].z; d[3] = vin[i].w; }
                else d[0] = vin[i];
            }
        }
        float* __restrict__ dw = lds_w + buf * p.w_buf;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int u = tid + 256 * i;
            const int row = u / C4;
            int c4 = (u - row * C4) << 2;
            if (row < w_rows) {
                if (SWZ) c4 ^= (row & 1) << 4;
                *reinterpret_cast<f32x4*>(dw + row * CT + c4) = vw[i];
            }
        }
    };

    stage_load(0);
    __syncthreads();  // zero fill complete before the first copy lands
    stage_store(0, 0);
    __syncthreads();
    MP_STAMP(t_pro);

    const int nq = p.CK >> 2;
    for (int ch = 0; ch < p.n_chunks; ++ch) {
        const int buf = p.nbuf == 2 ? (ch & 1) : 0;
        const bool more = ch + 1 < p.n_chunks;
        MP_STAMP(t0);
        if (more) stage_load(ch + 1);  // global loads fly while this chunk computes
        MP_STAMP(t1);
        const float* __restrict__ lin = lds_in + buf * p.in_buf;
        const float* __restrict__ lw = lds_w + buf * p.w_buf;
        // ---- MFMA over this chunk, fragments prefetched one k-step ahead
        float av[PS], bv[CS];
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) av[ps] = lin[a_off[ps]];
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) bv[cs] = lw[b_off[cs]];
        if constexpr (KS <= 3) {
            for (int q = 0; q < nq; ++q) {
                const int in_q = q * 4 * p.cin_plane;
                const int w_q = q * T * 4 * CT;
                const int qn = min(q + 1, nq - 1);  // the final prefetch re-reads a valid k-step (discarded)
    #pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int tn = (t + 1 < T) ? t + 1 : 0;
                    const int in_off = ((t + 1 < T) ? in_q : qn * 4 * p.cin_plane) + (tn / KS) * p.Wp + (tn % KS);
                    const int w_off = ((t + 1 < T) ? w_q : qn * T * 4 * CT) + tn * 4 * CT;
                    float an[PS], bn[CS];
    #pragma unroll
                    for (int ps = 0; ps < PS; ++ps) an[ps] = lin[a_off[ps] + in_off];
    #pragma unroll
                    for (int cs = 0; cs < CS; ++cs) bn[cs] = lw[b_off[cs] + w_off];
    #pragma unroll
                    for (int ps = 0; ps < PS; ++ps)
    #pragma unroll
                        for (int cs = 0; cs < CS; ++cs)
                            acc[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ps], bv[cs], acc[ps][cs], 0, 0, 0);
                    // pin the software pipeline: this k-step's MFMAs with the NEXT k-step's LDS reads spread
                    // between them (an fp32 MFMA occupies the pipe 32 cycles but the issue port only 8)
                    if (SCHED) {
                        constexpr int NR = PS + CS, NM = PS * CS, NPAIR = NR < NM ? NR : NM;
    #pragma unroll
                        for (int i = 0; i < NPAIR; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                        }
                        if (NM > NPAIR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NPAIR, 0);
                        if (NR > NPAIR) __builtin_amdgcn_sched_group_barrier(0x100, NR - NPAIR, 0);
                    }
    #pragma unroll
                    for (int ps = 0; ps < PS; ++ps) av[ps] = an[ps];
    #pragma unroll
                    for (int cs = 0; cs < CS; ++cs) bv[cs] = bn[cs];
                }
            }

        } else {
            // large kernels (7x7 stem): rows walked at run time, columns unrolled; no pinned pipeline
            // (a fully unrolled 49-tap body costs minutes of compile time for one launch per forward)
            for (int q = 0; q < nq; ++q) {
                for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx) {
                        const int in_off = q * 4 * p.cin_plane + dy * p.Wp + dx;
                        const int w_off = (q * T + dy * KS + dx) * 4 * CT;
#pragma unroll
                        for (int ps = 0; ps < PS; ++ps) av[ps] = lin[a_off[ps] + in_off];
#pragma unroll
                        for (int cs = 0; cs < CS; ++cs) bv[cs] = lw[b_off[cs] + w_off];
#pragma unroll
                        for (int ps = 0; ps < PS; ++ps)
#pragma unroll
                            for (int cs = 0; cs < CS; ++cs)
                                acc[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ps], bv[cs], acc[ps][cs], 0, 0, 0);
                    }
                }
            }
        }
        MP_STAMP(t2);
        if (more) stage_store(ch + 1, buf ^ 1);
        MP_STAMP(t3);
        if (more) __syncthreads();
        MP_STAMP(t4);
        s_load += t1 - t0; s_comp += t2 - t1; s_store += t3 - t2; s_bar += t4 - t3;
    }
    MP_STAMP(t_epi);

    // ---- epilogue: BN scale/shift (+res1) (+res2) (+ReLU) -> NCHW, with the output mapping
    const int plane_o = p.out_h * p.out_w;
    // 4 consecutive flattened tile pixels are 16 contiguous, aligned bytes of an NCHW plane when rows are a multiple of 4
    // wide - or when the tile holds whole images whose plane is (8x6 maps: 48 floats) and the mapping is the plain one
    const bool vec_ok = ((p.Wo & 3) == 0) ||
                        (p.R == p.Ho && ((p.Ho * p.Wo) & 3) == 0 && p.out_mul == 1 && p.out_rep == 1 && p.out_w == p.Wo &&
                         p.out_h == p.Ho && p.off_x == 0 && p.off_y == 0);
    if (vec_ok && p.out_mul == 1 && p.out_rep == 1) {
        // plain mapping: 16 B per lane.  All residual loads are issued first (clamped address + select, no
        // branches), then combined and stored, so their latencies overlap instead of adding up.
        const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out + (size_t)n0 * p.Cout * plane_o, (size_t)n_img * p.Cout * plane_o * 4);
        unsigned pix_off[PS];  // float offset inside the image group, kOob for padding lanes
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const unsigned p4 = (unsigned)((wp_i * PS + ps) * 16 + lq * 4);
            const unsigned pc = p4 < (unsigned)(p.G * p.RWo) ? p4 : 0u;
            const unsigned g = fastdiv(pc, p.RWo, p.magic_rwo);
            const unsigned rem = pc - g * p.RWo;
            const unsigned y = fastdiv(rem, p.Wo, p.magic_wo);
            const unsigned xx = rem - y * p.Wo;
            const int yy = y0 + y;
            const bool ok = p4 < (unsigned)(p.G * p.RWo) && n0 + (int)g < p.N && yy < p.Ho;
            pix_off[ps] = ok ? g * p.Cout * plane_o + (yy + p.off_y) * p.out_w + xx + p.off_x : kOob;
        }
        float sc[CS], sh[CS];
        unsigned co_off[CS];
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            const int co = ct * CT + (wc_i * CS + cs) * 16 + lr;
            const int cc = co < p.Cout ? co : 0;
            sc[cs] = p.scale[cc];
            sh[cs] = p.shift[cc];
            co_off[cs] = co < p.Cout ? (unsigned)cc * plane_o : kOob;
        }
        f32x4 r1[CS][PS], r2[CS][PS];
        if (p.res1) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.res1 + (size_t)n0 * p.Cout * plane_o, (size_t)n_img * p.Cout * plane_o * 4);
#pragma unroll
            for (int cs = 0; cs < CS; ++cs)
#pragma unroll
                for (int ps = 0; ps < PS; ++ps)
                    r1[cs][ps] = buf_load4(rs, ((co_off[cs] | pix_off[ps]) & kOob) ? kOob : (co_off[cs] + pix_off[ps]) * 4u);
        }
        if (p.res2) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.res2 + (size_t)n0 * p.Cout * plane_o, (size_t)n_img * p.Cout * plane_o * 4);
#pragma unroll
            for (int cs = 0; cs < CS; ++cs)
#pragma unroll
                for (int ps = 0; ps < PS; ++ps)
                    r2[cs][ps] = buf_load4(rs, ((co_off[cs] | pix_off[ps]) & kOob) ? kOob : (co_off[cs] + pix_off[ps]) * 4u);
        }
#pragma unroll
        for (int cs = 0; cs < CS; ++cs)
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) {
                f32x4 v = acc[ps][cs] * sc[cs] + sh[cs];
                if (p.res1) v += r1[cs][ps];
                if (p.res2) v += r2[cs][ps];
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                buf_store4(rs_o, ((co_off[cs] | pix_off[ps]) & kOob) ? kOob : (co_off[cs] + pix_off[ps]) * 4u, v);
            }
    } else {
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        const int co = ct * CT + (wc_i * CS + cs) * 16 + lr;
        const bool co_ok = co < p.Cout;
        const float sc = co_ok ? p.scale[co] : 0.f;
        const float sh = co_ok ? p.shift[co] : 0.f;
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const unsigned p4 = (unsigned)((wp_i * PS + ps) * 16 + lq * 4);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ps][cs][r] * sc + sh;
            if (!co_ok) continue;
            if (vec_ok) {
                if (p4 >= (unsigned)(p.G * p.RWo)) continue;
                const unsigned g = fastdiv(p4, p.RWo, p.magic_rwo);
                const unsigned rem = p4 - g * p.RWo;
                const unsigned y = fastdiv(rem, p.Wo, p.magic_wo);
                const unsigned xx = rem - y * p.Wo;
                const int n = n0 + g, yy = y0 + y;
                if (n >= p.N || yy >= p.Ho) continue;
                const size_t base = ((size_t)n * p.Cout + co) * plane_o;
                if (p.out_mul == 1 && p.out_rep == 1) {
                    const size_t o = base + (size_t)(yy + p.off_y) * p.out_w + xx + p.off_x;
                    float4 r4 = make_float4(v[0], v[1], v[2], v[3]);
                    if (p.res1) { const float4 t4 = *reinterpret_cast<const float4*>(p.res1 + o); r4.x += t4.x; r4.y += t4.y; r4.z += t4.z; r4.w += t4.w; }
                    if (p.res2) { const float4 t4 = *reinterpret_cast<const float4*>(p.res2 + o); r4.x += t4.x; r4.y += t4.y; r4.z += t4.z; r4.w += t4.w; }
                    if (p.relu) { r4.x = fmaxf(r4.x, 0.f); r4.y = fmaxf(r4.y, 0.f); r4.z = fmaxf(r4.z, 0.f); r4.w = fmaxf(r4.w, 0.f); }
                    *reinterpret_cast<float4*>(p.out + o) = r4;
                } else if (p.out_rep == p.out_mul && (p.out_rep == 2 || p.out_rep == 4 || p.out_rep == 8)) {
                    // nearest up-sample by s (HRModule fuse rows): 4 pixels -> 4s contiguous outputs per row, s rows.
                    // Per output row: all residual loads first (range-checked buffer loads), then the stores.
                    const int s = p.out_rep;
                    const size_t img_base = (size_t)n * p.Cout * plane_o;
                    const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + img_base, (size_t)p.Cout * plane_o * 4);
                    const __amdgpu_buffer_rsrc_t r1s = make_rsrc((p.res1 ? p.res1 : p.out) + img_base, (size_t)p.Cout * plane_o * 4);
                    const __amdgpu_buffer_rsrc_t r2s = make_rsrc((p.res2 ? p.res2 : p.out) + img_base, (size_t)p.Cout * plane_o * 4);
                    const f32x4 vv = (f32x4){v[0], v[1], v[2], v[3]};
                    for (int a = 0; a < s; ++a) {
                        const unsigned o = ((unsigned)co * plane_o + (unsigned)(yy * s + a + p.off_y) * p.out_w + xx * s + p.off_x) * 4u;
                        for (int jb = 0; jb < s; jb += 4) {  // batches of <= 4 x 16 B keep the register footprint small
                            f32x4 t1[4], t2[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (jb + j < s) {
                                    if (p.res1) t1[j] = buf_load4(r1s, o + 16u * (jb + j));
                                    if (p.res2) t2[j] = buf_load4(r2s, o + 16u * (jb + j));
                                }
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (jb + j < s) {
                                    const int jj = s == 2 ? 0 : (s == 4 ? jb + j : ((jb + j) >> 1));
                                    const float e = (jj == 0) ? vv.x : (jj == 1) ? vv.y : (jj == 2) ? vv.z : vv.w;
                                    f32x4 r4 = (f32x4){e, e, e, e};
                                    if (s == 2) r4 = (j == 0) ? (f32x4){vv.x, vv.x, vv.y, vv.y} : (f32x4){vv.z, vv.z, vv.w, vv.w};
                                    if (p.res1) r4 += t1[j];
                                    if (p.res2) r4 += t2[j];
                                    if (p.relu) { r4.x = fmaxf(r4.x, 0.f); r4.y = fmaxf(r4.y, 0.f); r4.z = fmaxf(r4.z, 0.f); r4.w = fmaxf(r4.w, 0.f); }
                                    buf_store4(ro, o + 16u * (jb + j), r4);
                                }
                            }
                        }
                    }
                } else {
                    for (int r = 0; r < 4; ++r) {
                        for (int a = 0; a < p.out_rep; ++a)
                            for (int bb = 0; bb < p.out_rep; ++bb) {
                                const size_t o = base + (size_t)(yy * p.out_mul + p.off_y + a) * p.out_w +
                                                 (size_t)(xx + r) * p.out_mul + p.off_x + bb;
                                float e = (r == 0) ? v[0] : (r == 1) ? v[1] : (r == 2) ? v[2] : v[3];
                                if (p.res1) e += p.res1[o];
                                if (p.res2) e += p.res2[o];
                                if (p.relu) e = fmaxf(e, 0.f);
                                p.out[o] = e;
                            }
                    }
                }
            } else {
                for (int r = 0; r < 4; ++r) {
                    const unsigned pl = p4 + r;
                    if (pl >= (unsigned)(p.G * p.RWo)) continue;
                    const unsigned g = fastdiv(pl, p.RWo, p.magic_rwo);
                    const unsigned rem = pl - g * p.RWo;
                    const unsigned y = fastdiv(rem, p.Wo, p.magic_wo);
                    const unsigned xx = rem - y * p.Wo;
                    const int n = n0 + g, yy = y0 + y;
                    if (n >= p.N || yy >= p.Ho) continue;
                    const size_t base = ((size_t)n * p.Cout + co) * plane_o;
                    const float e0 = (r == 0) ? v[0] : (r == 1) ? v[1] : (r == 2) ? v[2] : v[3];
                    for (int a = 0; a < p.out_rep; ++a)
                        for (int bb = 0; bb < p.out_rep; ++bb) {
                            const size_t o = base + (size_t)(yy * p.out_mul + p.off_y + a) * p.out_w +
                                             (size_t)xx * p.out_mul + p.off_x + bb;
                            float e = e0;
                            if (p.res1) e += p.res1[o];
                            if (p.res2) e += p.res2[o];
                            if (p.relu) e = fmaxf(e, 0.f);
                            p.out[o] = e;
                        }
                }
            }
        }
    }
    }
#if MP_CONV_STAMPS
    {
        MP_STAMP(t_end);
        if (p.dbg && tid == 0) {
            unsigned long long* d = p.dbg + (size_t)blockIdx.x * 8;
            d[0] = t_end - t_start; d[1] = t_pro - t_start; d[2] = s_load; d[3] = s_comp; d[4] = s_store; d[5] = s_bar;
            d[6] = t_end - t_epi; d[7] = t_start;
        }
    }
#endif
}

// tile variants: index -> (PS, CS, WAVES_P, WAVES_C); CT = 16*CS*WAVES_C, PT = 16*PS*WAVES_P
// tile variants: cout tile x pixel tile (+ _L light / _H regular where both exist); CT = 16*CS*WAVES_C, PT = 16*PS*WAVES_P
enum ConvVariant {
    V_CT32_PT192 = 0,   // light (3 workgroups / CU)
    V_CT64_PT192 = 1,
    V_CT48_PT192 = 2,
    V_CT64_PT96 = 3,
    V_CT32_PT96 = 4,
    V_CT64_PT96_L = 5,  // light
    V_CT32_PT192_H = 6, // regular
    V_CT32_PT96_L = 7,  // light
    V_COUNT = 8
};

inline void variant_dims(int v, int& ct, int& pt) {
    static const int cts[V_COUNT] = {32, 64, 48, 64, 32, 64, 32, 32};
    static const int pts[V_COUNT] = {192, 192, 192, 96, 96, 96, 192, 96};
    ct = cts[v];
    pt = pts[v];
}
inline bool variant_light(int v) {
    static const bool l[V_COUNT] = {true, false, false, false, false, true, false, true};
    return l[v];
}

// Staging units (16 B) per thread per chunk and the occupancy a kernel is compiled for.  LIGHT kernels take small
// chunks (few staging registers, <= 52 KiB LDS) and run THREE workgroups per CU, so one workgroup's prologue / staging /
// epilogue hides under the others' MFMA phases; the regular ones keep big chunks and two workgroups.  Which is faster
// depends on the layer (K, map size, batch): the plan-time autotuner measures both.
constexpr int stage_ni(int ks, bool light, bool vec) { return vec ? (light ? 4 : 8) : (light ? 10 : 8); }
constexpr int stage_nw(int ks, bool light) { return ks == 7 ? 13 : (light ? 3 : 6); }
constexpr int stage_occ(bool light) { return light ? 3 : 2; }

template <int KS, int S, int PS, int CS, int WAVES_P, int WAVES_C, bool VEC, bool LIGHT>
int launch_variant(const ConvKParams& p, size_t lds_bytes, hipStream_t s) {
    auto kern = conv_mfma_kernel<KS, S, PS, CS, WAVES_P, WAVES_C, stage_ni(KS, LIGHT, VEC), stage_nw(KS, LIGHT), VEC, stage_occ(LIGHT)>;
    static AttrOnce attr_set_once;
    if (attr_set_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(p.total_blocks), dim3(256), lds_bytes, s, p);
    return check_launch();
}

template <int KS, int S, bool VEC>
int launch_ks_v(const ConvKParams& p, int variant, size_t lds_bytes, hipStream_t s) {
    switch (variant) {
        case V_CT32_PT192: return launch_variant<KS, S, 3, 2, 4, 1, VEC, true>(p, lds_bytes, s);
        case V_CT64_PT192: return launch_variant<KS, S, 3, 4, 4, 1, VEC, false>(p, lds_bytes, s);
        case V_CT48_PT192: return launch_variant<KS, S, 3, 3, 4, 1, VEC, false>(p, lds_bytes, s);
        case V_CT64_PT96: return launch_variant<KS, S, 3, 2, 2, 2, VEC, false>(p, lds_bytes, s);
        case V_CT32_PT96: return launch_variant<KS, S, 3, 1, 2, 2, VEC, false>(p, lds_bytes, s);
        case V_CT64_PT96_L: return launch_variant<KS, S, 3, 2, 2, 2, VEC, true>(p, lds_bytes, s);
        case V_CT32_PT192_H: return launch_variant<KS, S, 3, 2, 4, 1, VEC, false>(p, lds_bytes, s);
        case V_CT32_PT96_L: return launch_variant<KS, S, 3, 1, 2, 2, VEC, true>(p, lds_bytes, s);
        default: return MP_ERR_UNSUPPORTED;
    }
}

template <int KS, int S>
int launch_ks(const ConvKParams& p, int variant, size_t lds_bytes, hipStream_t s) {
    return p.vec ? launch_ks_v<KS, S, true>(p, variant, lds_bytes, s) : launch_ks_v<KS, S, false>(p, variant, lds_bytes, s);
}

// one translation unit per kernel size (parallel compilation)
int launch_conv_k1(const ConvKParams& p, int stride, int variant, size_t lds_bytes, hipStream_t s);
int launch_conv_k2(const ConvKParams& p, int stride, int variant, size_t lds_bytes, hipStream_t s);
int launch_conv_k3s1(const ConvKParams& p, int variant, size_t lds_bytes, hipStream_t s);
int launch_conv_k3s2(const ConvKParams& p, int variant, size_t lds_bytes, hipStream_t s);
int launch_conv_k7(const ConvKParams& p, int stride, int variant, size_t lds_bytes, hipStream_t s);

}  // namespace mp
