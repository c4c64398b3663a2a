// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores of gfx950: the 3x3 stride-1 padding-1 convolutions of the HRNet
// branches (hrnet.py:51-64, 202-241 - 85 % of the network's FLOP) with 16 multiplications per 2x2 output tile and (cin, cout)
// pair instead of 36.  Everything is fp32: input transform  V = B^T d B  (adds only), 16 independent GEMMs
// M_xi[tile][cout] = sum_cin V_xi[tile][cin] * U_xi[cin][cout]  on v_mfma_f32_16x16x4_f32, output transform  Y = A^T M A
// (adds only), then the same folded-BatchNorm / residual / ReLU epilogue as the direct kernel (conv_mfma.h).
//
//   workgroup   256 threads, one image band of TR x TW <= 48 output tiles (2x2 pixels each) x 32 output channels, all of Cin
//               in chunks of 8 channels; a run of up to 8 consecutive tiles per workgroup.  TEAMS = 2: 512 threads, two
//               four-wave teams on one band and 64 output channels - shared raw rows / transform / V, see the kernel comment
//   per chunk   (1) raw input rows  global -> registers -> LDS  [8][R+2][W+4] (zero halo, double-buffered; the loads of chunk
//                   c+2 fly under the arithmetic of chunk c)
//               (2) input transform: a thread takes two horizontally adjacent tiles of one channel (4 x 6 patch: b128 + b64
//                   LDS reads per row), writes V[xi][cin][tile] (tile pitch 48 == 16 mod 32: the four cin rows of an MFMA
//                   operand fetch sit on disjoint banks); woven between the MFMAs of the previous chunk
//               (3) MFMA: wave w owns xi = 4w .. 4w+3 for all 3 x 2 (tile block, cout block) pairs = 24 accumulators;
//                   U comes straight from global memory / L2 into registers one chunk ahead (no LDS copy of the weights)
//   epilogue    the column step of Y = A^T M A on the wave's own accumulators (it owns row w of every 4x4 M), the two values
//               per row of both 16-channel halves through LDS in one exchange, the row step + scale/shift (+res1)(+res2)(+ReLU)
//               by the reading thread: four tiles of one cout = 2x2 pixels each, 16-byte stores when they are one pixel row.
//   LDS         73 KiB -> two workgroups per CU (TEAMS = 2: 126 KiB, one): while one transforms, the other feeds the matrix cores.
#include <stdlib.h>

#include "conv_mfma.h"
#include "conv_wino.h"

namespace mp {

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int kCK = 8;    // input channels per chunk
constexpr int kTP = 48;   // tile pitch of V (floats), tiles per workgroup <= 48
constexpr int kXP = 52;   // tile pitch of the accumulator exchange
constexpr int kVFloats = 16 * kCK * kTP;
constexpr int kXFloats = 16 * 16 * kXP;

__device__ __forceinline__ void wg_barrier() {
    // LDS traffic of this wave done, then the workgroup barrier; global loads stay in flight (a __syncthreads() would drain them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// GROUP: a band = G whole small images (W % 4 != 0 maps, e.g. 8x6: 12 tiles per image, four images per workgroup); the raw
// planes are staged as contiguous float4 units whose four elements may straddle rows, and the two tiles of a transform item
// are consecutive tiles, not neighbours in a row.
// TEAMS = 2: 512 threads = two teams of four waves on ONE band: the teams share the raw rows, the input transform (done by team 0)
// and V, and each computes its own 32 output channels of a 64-channel cout tile - the transform work per MFMA halves.
template <int NI, bool QROW, bool GROUP, int TEAMS>
__global__ __launch_bounds__(256 * TEAMS, TEAMS == 1 ? 2 : 1) void conv_wino_f32_kernel(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int raw_buf = kCK * p.cin_plane + 4;  // + one float4 that absorbs the stores of threads without a staging unit
    float* __restrict__ lds_raw = smem;                 // [2][raw_buf]
    float* __restrict__ lds_v = smem + 2 * raw_buf;     // [2][16][kCK][kTP]
    const int tid_wg = threadIdx.x;
    const int team = TEAMS == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid_wg >> 8);
    float* __restrict__ lds_x = lds_v + team * kXFloats;  // epilogue: [16][16][kXP] per team over the V buffers (never over the raw
                                                          // buffers: their zero halo columns must survive into the next tile)

    MP_STAMP(t_start);
    [[maybe_unused]] unsigned long long s_xf = 0, s_b1 = 0, s_mm = 0, s_st = 0, s_b2 = 0, s_pro = 0, s_ep = 0;
    const int tid = tid_wg & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // thread / wave inside the team
    const int lq = lane >> 4, lr = lane & 15;

    // A workgroup walks p.tiles_per_wg consecutive tiles (same staging tables, one zero fill; the raw rows of the next tile are
    // requested before the epilogue of the current one, so only the first tile of a workgroup sees the HBM latency).
    int wg = blockIdx.x;
    {   // XCD-aware workgroup id (blocks b, b+8, ... share an XCD): neighbouring tiles (cout tiles of one band) share an L2
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = wg & 7, j = wg >> 3;
        wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const int HW = p.H * p.W;
    int ct, n, y0, n_img;  // coordinates of the tile whose INPUT side is being worked on
    auto tile_coords = [&](int tile) {
        int b = tile;
        ct = b % p.n_ct;
        b /= p.n_ct;
        const int band = GROUP ? 0 : b % p.bands;
        n = GROUP ? b * p.G : b / p.bands;  // GROUP: b = image group, n = its first image
        y0 = band * p.R;
        n_img = GROUP ? min(p.G, p.N - n) : 1;
    };
    const int tile_first = wg * p.tiles_per_wg;
    tile_coords(tile_first);

    // staging tables: float4 units of the chunk's rows.  Branch-free use: a unit outside the image loads zeros (range-checked
    // buffer load) and stores them - halo rows are rewritten with zeros, harmless - and a thread without a unit stores its
    // zeros into the spare float4 behind the buffer.
    unsigned isrc[NI], irel[NI];  // irel: tile-independent part (kOob: no unit); isrc: + the tile's row validity
    int irow[NI], idst[NI];
    int idst4[GROUP ? NI : 1][4];  // GROUP: one LDS offset per element of a unit
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const unsigned u = tid_wg + 256 * TEAMS * i;  // every thread of the workgroup stages
        irel[i] = kOob;
        irow[i] = 0;
        idst[i] = kCK * p.cin_plane;
        if constexpr (GROUP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) idst4[i][j] = kCK * p.cin_plane + j;
            if (u < (unsigned)(kCK * p.upc)) {  // upc = G * HW / 4 units per channel, upr = HW / 4 per image
                const unsigned c = fastdiv(u, p.upc, p.magic_upc);
                const unsigned rem = u - c * p.upc;
                const unsigned g = fastdiv(rem, p.upr, p.magic_upr);
                const unsigned k4 = (rem - g * p.upr) * 4;
                irel[i] = ((g * p.Cin + c) * HW + k4) * 4u;  // images past the batch: beyond the resource, zeros
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned k = k4 + j, row = fastdiv(k, p.W, p.magic_w), col = k - row * p.W;
                    idst4[i][j] = (int)(c * p.cin_plane + g * p.img_plane + (row + 1) * p.Wp + 1 + col);
                }
            }
        } else if (u < (unsigned)(kCK * p.upc)) {
            const unsigned c = fastdiv(u, p.upc, p.magic_upc);
            const unsigned rem = u - c * p.upc;
            const unsigned r = fastdiv(rem, p.upr, p.magic_upr);
            const unsigned xu = rem - r * p.upr;
            irel[i] = (c * HW + xu * 4) * 4u;
            irow[i] = (int)r;
            idst[i] = (int)(c * p.cin_plane + r * p.Wp + 1 + xu * 4);
        }
    }
    auto tile_rows = [&]() {  // the rows of the current tile's band that lie inside the image
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int yin = y0 - 1 + irow[i];
            isrc[i] = (GROUP || (yin >= 0 && yin < p.H)) && irel[i] != kOob ? irel[i] + (GROUP ? 0u : (unsigned)yin * p.W * 4u) : kOob;
        }
    };
    tile_rows();
    __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x + (size_t)n * p.Cin * HW, (size_t)n_img * p.Cin * HW * 4);
    const __amdgpu_buffer_rsrc_t rs_u = make_rsrc(p.u, (size_t)(p.Cin >> 2) * 16 * 4 * p.Cout_pad16 * 4);

    // transform item of this thread: (cin, pair of adjacent tiles); the threads beyond the item count repeat the first items
    // (same values to the same addresses) so that the transform needs no branch
    const int pairs = p.M >> 1;
    int xf_raw = 0, xf_rawb = 0, xf_v = 0;  // patch of the first tile, of the second tile (GROUP), V offset of the pair
    {
        const int items = kCK * pairs;  // <= 192
        const unsigned t = (unsigned)tid % (unsigned)items;
        const unsigned c = fastdiv(t, pairs, p.magic_pairs);
        const unsigned tile0 = (t - c * pairs) * 2;
        auto patch = [&](unsigned tile) {
            unsigned g = 0, r = tile;
            if constexpr (GROUP) {
                g = fastdiv(tile, p.tpi, p.magic_tpi);
                r = tile - g * p.tpi;
            }
            const unsigned ty = fastdiv(r, p.TW, p.magic_tw), tx = r - ty * p.TW;
            return (int)(c * p.cin_plane + g * p.img_plane + (2 * ty) * p.Wp + 2 * tx);
        };
        xf_raw = patch(tile0);
        xf_rawb = GROUP ? patch(tile0 + 1) : xf_raw + 2;
        xf_v = c * kTP + tile0;
    }

    // U fragments of this wave: xi = 4 * wave + i, cout block nb, k-step q of the chunk.  U is stored [cin][cout][xi]: the four
    // xi of a wave are ONE 16-byte load per (nb, q) - 4 loads per chunk instead of 16
    unsigned u_off[2];
    auto tile_u = [&]() {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int co = (ct * TEAMS + team) * 32 + nb * 16 + lr;
            u_off[nb] = co < p.Cout_pad16 ? (unsigned)((lq * p.Cout_pad16 + co) * 16 + wave * 4) * 4u : kOob;
        }
    };
    tile_u();
    const unsigned u_q = (unsigned)(4 * p.Cout_pad16 * 16) * 4u;  // next k-step (4 input channels)
    // ONE register set for the U fragments: the k-step q half of chunk c + 1 is loaded into the registers of the k-step q half of
    // chunk c right after the MFMAs that consumed it (its latency runs under the other half) - no second set, no copies
    float ucur[4][2][2];
    auto load_u_half = [&](int ch, int q) {
        const unsigned base = (unsigned)(ch * 2 + q) * u_q;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const f32x4 v = buf_load4(rs_u, u_off[nb] + base);  // kOob + offset stays out of range
#pragma unroll
            for (int i = 0; i < 4; ++i) ucur[i][nb][q] = v[i];
        }
    };

    f32x4 vin[NI];
    auto stage_load = [&](int ch) {
        const unsigned xo = (unsigned)(ch * kCK) * HW * 4u;
#pragma unroll
        for (int i = 0; i < NI; ++i) vin[i] = buf_load4(rs_x, isrc[i] + xo);  // kOob + offset stays out of range
    };
    auto stage_store = [&](int buf) {
        float* __restrict__ d0 = lds_raw + buf * raw_buf;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if constexpr (GROUP) {
                d0[idst4[i][0]] = vin[i].x; d0[idst4[i][1]] = vin[i].y; d0[idst4[i][2]] = vin[i].z; d0[idst4[i][3]] = vin[i].w;
            } else {
                float* d = d0 + idst[i];
                d[0] = vin[i].x; d[1] = vin[i].y; d[2] = vin[i].z; d[3] = vin[i].w;
            }
        }
    };
    // input transform of one chunk: raw[rb] -> V[vb]; split in three parts so that the MFMA stream can be woven between them
    // columns 0..3 = patch of the first tile; the second tile's patch is columns 2..5 (its row neighbour) or, GROUP, its own
    // four columns 4..7 (consecutive tiles need not be neighbours there)
    constexpr int XC = GROUP ? 8 : 6, XB = GROUP ? 4 : 2;
    float xd[4][XC], xt[4][XC];
    auto xf_read = [&](int rb) {
        const float* __restrict__ src = lds_raw + rb * raw_buf + xf_raw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (GROUP) {
                const float* __restrict__ srcb = lds_raw + rb * raw_buf + xf_rawb;
                const float2 a0 = *reinterpret_cast<const float2*>(src + r * p.Wp), a1 = *reinterpret_cast<const float2*>(src + r * p.Wp + 2);
                const float2 b0 = *reinterpret_cast<const float2*>(srcb + r * p.Wp), b1 = *reinterpret_cast<const float2*>(srcb + r * p.Wp + 2);
                xd[r][0] = a0.x; xd[r][1] = a0.y; xd[r][2] = a1.x; xd[r][3] = a1.y;
                xd[r][4] = b0.x; xd[r][5] = b0.y; xd[r][6] = b1.x; xd[r][7] = b1.y;
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(src + r * p.Wp);
                const float2 c2 = *reinterpret_cast<const float2*>(src + r * p.Wp + 4);
                xd[r][0] = a.x; xd[r][1] = a.y; xd[r][2] = a.z; xd[r][3] = a.w; xd[r][4] = c2.x; xd[r][5] = c2.y;
            }
        }
    };
    auto xf_cols = [&]() {
#pragma unroll
        for (int c = 0; c < XC; ++c) {
            xt[0][c] = xd[0][c] - xd[2][c];
            xt[1][c] = xd[1][c] + xd[2][c];
            xt[2][c] = xd[2][c] - xd[1][c];
            xt[3][c] = xd[1][c] - xd[3][c];
        }
    };
    auto xf_rows_write = [&](int vb, int i) {  // row i of the column-transformed patches -> xi = 4 i .. 4 i + 3 of both tiles
        float* __restrict__ dst = lds_v + vb * kVFloats + xf_v;
        float va[4], vbv[4];
        va[0] = xt[i][0] - xt[i][2]; va[1] = xt[i][1] + xt[i][2]; va[2] = xt[i][2] - xt[i][1]; va[3] = xt[i][1] - xt[i][3];
        vbv[0] = xt[i][XB] - xt[i][XB + 2]; vbv[1] = xt[i][XB + 1] + xt[i][XB + 2]; vbv[2] = xt[i][XB + 2] - xt[i][XB + 1];
        vbv[3] = xt[i][XB + 1] - xt[i][XB + 3];
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float2*>(dst + (i * 4 + j) * (kCK * kTP)) = make_float2(va[j], vbv[j]);
    };

    // the raw rows of chunks 0 and 1 of a tile are requested together (second register set) - for the first tile right away,
    // under the zero fill of the halo; for the following tiles before the epilogue of their predecessor
    f32x4 vin1[NI];
    auto tile_request = [&]() {
        const unsigned xo = (unsigned)kCK * HW * 4u;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            vin[i] = buf_load4(rs_x, isrc[i]);
            vin1[i] = buf_load4(rs_x, isrc[i] + xo);
        }
        load_u_half(0, 0);
        load_u_half(0, 1);
    };
    tile_request();
    {   // zero both raw buffers once: the halo columns are never written by the chunk copies
        const int n4 = (2 * raw_buf) >> 2;  // raw_buf is a multiple of 4
        float4* z = reinterpret_cast<float4*>(lds_raw);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid_wg; i < n4; i += 256 * TEAMS) z[i] = zero;
    }
    wg_barrier();  // zero fill complete

    // epilogue constants of this thread (tile-independent part)
    const int plane_o = HW;
    const bool ep_on = tid < 192;
    const unsigned co_l = ep_on ? fastdiv((unsigned)tid, 12, 0x15555556u) : 0u;  // tid / 12
    const unsigned quad = ep_on ? tid - co_l * 12 : 0u;
    const unsigned row_b = (unsigned)p.W * 4u;
    unsigned pix_rel[QROW ? 1 : 4];  // byte offset of a tile's first output pixel relative to the band start; kOob: no such tile
    int pix_row[QROW ? 1 : 4];
#pragma unroll
    for (int e = 0; e < (QROW ? 1 : 4); ++e) {
        const unsigned tile = quad * 4 + e;
        unsigned g = 0, r = tile;
        if constexpr (GROUP) {
            g = fastdiv(tile, p.tpi, p.magic_tpi);
            r = tile - g * p.tpi;
        }
        const unsigned ty = fastdiv(r, p.TW, p.magic_tw);
        const unsigned tx = r - ty * p.TW;
        pix_row[e] = 2 * (int)ty;
        // GROUP: + the image's offset inside the group (an image past the batch is beyond the resources: nothing loaded / stored)
        pix_rel[e] = (ep_on && tile < (unsigned)p.M) ? (unsigned)(g * p.Cout * plane_o + 2 * ty * p.W + 2 * tx) * 4u : kOob;
    }

    for (int it = 0; it < p.tiles_per_wg; ++it) {
    if (tile_first + it >= p.total_blocks) break;  // workgroup-uniform
    MP_STAMP(t_tile);
    f32x4 acc[4][3][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[i][mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    stage_store(0);
    wg_barrier();
    if (team == 0) {
        xf_read(0);
        xf_cols();
#pragma unroll
        for (int i = 0; i < 4; ++i) xf_rows_write(0, i);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) vin[i] = vin1[i];
    stage_store(1);
    wg_barrier();
    MP_STAMP(t_pro);
    s_pro += t_pro - t_tile;

    // chunk ch: the 48 MFMAs of this wave over V[ch & 1], with the input transform of chunk ch + 1 (raw[(ch + 1) & 1] ->
    // V[(ch + 1) & 1]) woven between them; the raw rows of chunk ch + 2 fly in from global memory meanwhile.  The body is
    // branch-free and the same for every chunk (past the last chunk the buffer loads are out of range and return zeros, the
    // transform works on stale rows into a V buffer nobody reads): one loop, no peeled copy, the accumulators stay put.
    const int a_base = wave * 4 * (kCK * kTP) + lq * kTP + lr;
    // (one team: two chunks per trip - the buffer parities are constants; an odd chunk count runs one chunk past the end, zeros
    // times zeros, see above.  Two teams, at the 256-register ceiling, spill with the double-length body: one chunk per trip)
    auto chunk = [&](int ch, const int par) {
        MP_STAMP(t0);
        stage_load(ch + 2);
        const float* __restrict__ vcur = lds_v + par * kVFloats + a_base;
        if (TEAMS == 1 || team == 0) xf_read(par ^ 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* __restrict__ va = vcur + i * (kCK * kTP) + q * 4 * kTP;
                float av[3];
#pragma unroll
                for (int mb = 0; mb < 3; ++mb) av[mb] = va[mb * 16];
#pragma unroll
                for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[i][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], ucur[i][nb][q], acc[i][mb][nb], 0, 0, 0);
                if (TEAMS == 1 || team == 0) {  // (team: wave-uniform)
                    if (q == 0 && i == 1) xf_cols();
                    if (q == 0 && i >= 2) xf_rows_write(par ^ 1, i - 2);
                    if (q == 1 && i < 2) xf_rows_write(par ^ 1, i + 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // the refill below must not move above the MFMAs that read these registers' old values
            load_u_half(ch + 1, q);
        }
        MP_STAMP(t3);
        stage_store(par);
        MP_STAMP(t4);
        wg_barrier();
        MP_STAMP(t5);
        s_mm += t3 - t0; s_st += t4 - t3; s_b2 += t5 - t4;
    };
    if constexpr (TEAMS == 1) {
        for (int ch = 0; ch < p.n_chunks; ch += 2) {
            chunk(ch, 0);
            chunk(ch + 1, 1);
        }
    } else {
        for (int ch = 0; ch < p.n_chunks; ++ch) chunk(ch, ch & 1);
    }
    MP_STAMP(t_epi);

    // ---- accumulators -> LDS -> output transform -> epilogue.  A thread (0..191) takes cout co_l of each 16-channel half and
    // the four tiles 4 quad .. 4 quad + 3; QROW (launch condition: full 48-tile bands, TW % 4 == 0): those
    // four tiles are eight consecutive pixels of two rows -> 16-byte residual loads and stores.  The residual loads are issued
    // before the accumulators go to LDS, so their latency runs under the exchange.
    const int e_ct = ct, e_n = n, e_y0 = y0;
    const size_t img_o = (size_t)n_img * p.Cout * plane_o * 4;
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out + (size_t)e_n * p.Cout * plane_o, img_o);
    const __amdgpu_buffer_rsrc_t rs_r1 = make_rsrc(p.res1 ? p.res1 + (size_t)e_n * p.Cout * plane_o : p.out, p.res1 ? img_o : 0);
    const __amdgpu_buffer_rsrc_t rs_r2 = make_rsrc(p.res2 ? p.res2 + (size_t)e_n * p.Cout * plane_o : p.out, p.res2 ? img_o : 0);
    unsigned pix[QROW ? 1 : 4];  // byte offset of a tile's first output pixel inside a channel plane; kOob: tile / row not there
#pragma unroll
    for (int e = 0; e < (QROW ? 1 : 4); ++e)
        pix[e] = (pix_rel[e] != kOob && e_y0 + pix_row[e] < p.H) ? pix_rel[e] + (unsigned)(e_y0 * p.W) * 4u : kOob;
    // the next tile of this workgroup: its first raw rows and U fragments are requested now and land during the epilogue
    if (it + 1 < p.tiles_per_wg && tile_first + it + 1 < p.total_blocks) {
        tile_coords(tile_first + it + 1);
        tile_rows();
        tile_u();
        rs_x = make_rsrc(p.x + (size_t)n * p.Cin * HW, (size_t)n_img * p.Cin * HW * 4);
        tile_request();
    }
    // Output transform Y = A^T M A in two steps.  A wave owns row i = wave of the 4x4 M of every (tile, cout): the column step
    // P[i][b] = sum_j M[i][j] A[j][b] happens on its own accumulators in registers, so only the 2 (not 4) values per row go
    // through LDS - both 16-channel halves in ONE exchange ([half][row i][b][cout][tile], the size of the old one-half
    // buffer) and two barriers per tile.  The row step Y[a][b] = sum_i A^T[a][i] P[i][b] is done by the reading thread.
    const int co_base = (e_ct * TEAMS + team) * 32;
    f32x4 r1q[2][2][2], r2q[2][2][2];
    float2 r1v[2][4][2], r2v[2][4][2];
    unsigned co_off[2];
    auto res_request = [&](int nb) {
        const int co = co_base + nb * 16 + (int)co_l;
        co_off[nb] = co < p.Cout ? (unsigned)co * plane_o * 4u : kOob;  // kOob + pixel offset stays out of range
        if constexpr (QROW) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const unsigned o = (pix[0] == kOob ? kOob : co_off[nb] + pix[0]) + a * row_b + h2 * 16u;
                    r1q[nb][a][h2] = buf_load4(rs_r1, o);
                    r2q[nb][a][h2] = buf_load4(rs_r2, o);
                }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const unsigned o = (pix[e] == kOob ? kOob : co_off[nb] + pix[e]) + a * row_b;
                    r1v[nb][e][a] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_r1, o, 0, 0));
                    r2v[nb][e][a] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_r2, o, 0, 0));
                }
        }
    };
    res_request(0);  // the residual rows of the first half fly under the exchange, those of the second under the first half
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
            const f32x4 p0 = acc[0][mb][nb] + acc[1][mb][nb] + acc[2][mb][nb];
            const f32x4 p1 = acc[1][mb][nb] - acc[2][mb][nb] - acc[3][mb][nb];
            float* __restrict__ dst = lds_x + (((nb * 4 + wave) * 2) * 16 + lr) * kXP + mb * 16 + lq * 4;
            *reinterpret_cast<f32x4*>(dst) = p0;
            *reinterpret_cast<f32x4*>(dst + 16 * kXP) = p1;
        }
    res_request(1);
    wg_barrier();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        if (co_base + nb * 16 >= p.Cout_pad16) break;  // wave-uniform; no barrier below
        if (ep_on) {
            const int co = co_base + nb * 16 + (int)co_l;
            f32x4 pr[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    pr[i][b] = *reinterpret_cast<const f32x4*>(lds_x + (((nb * 4 + i) * 2 + b) * 16 + co_l) * kXP + quad * 4);
            const int cc = co < p.Cout ? co : 0;
            const float sc = p.scale[cc], sh = p.shift[cc];
            // the four tiles side by side: component e of every vector = tile e; y[a][0] = left pixel of each tile in output
            // row a, y[a][1] = right pixel
            f32x4 y[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                y[0][b] = (pr[0][b] + pr[1][b] + pr[2][b]) * sc + sh;
                y[1][b] = (pr[1][b] - pr[2][b] - pr[3][b]) * sc + sh;
            }
            if constexpr (QROW) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    f32x4 lo = (f32x4){y[a][0][0], y[a][1][0], y[a][0][1], y[a][1][1]};
                    f32x4 hi = (f32x4){y[a][0][2], y[a][1][2], y[a][0][3], y[a][1][3]};
                    lo += r1q[nb][a][0] + r2q[nb][a][0];
                    hi += r1q[nb][a][1] + r2q[nb][a][1];
                    if (p.relu) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { lo[k] = fmaxf(lo[k], 0.f); hi[k] = fmaxf(hi[k], 0.f); }
                    }
                    const unsigned o = (pix[0] == kOob ? kOob : co_off[nb] + pix[0]) + a * row_b;
                    buf_store4(rs_o, o, lo);
                    buf_store4(rs_o, o + 16u, hi);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        float2 v = make_float2(y[a][0][e] + r1v[nb][e][a].x + r2v[nb][e][a].x, y[a][1][e] + r1v[nb][e][a].y + r2v[nb][e][a].y);
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs_o,
                                                              (pix[e] == kOob ? kOob : co_off[nb] + pix[e]) + a * row_b, 0, 0);
                    }
            }
        }
    }
    wg_barrier();  // the exchange buffer is free for the next tile's V
    MP_STAMP(t_tile_end);
    s_ep += t_tile_end - t_epi;
    }  // tiles of this workgroup
#if MP_CONV_STAMPS
    {
        MP_STAMP(t_end);
        if (p.dbg && tid_wg == 0) {
            unsigned long long* d = p.dbg + (size_t)blockIdx.x * 8;
            d[0] = t_end - t_start; d[1] = s_pro; d[2] = s_xf; d[3] = s_b1; d[4] = s_mm; d[5] = s_st; d[6] = s_b2;
            d[7] = s_ep;
        }
    }
#endif
}

// U = G g G^T per (cout, cin), laid out [Cin_pad4][Cout_pad16][16 xi]
__global__ __launch_bounds__(256) void pack_weight_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin,
                                                               int cin_pad4, int cout_pad16, int transposed) {
    const size_t total = (size_t)cin_pad4 * cout_pad16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout_pad16), ci = (int)(i / cout_pad16);
        float u[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) u[k] = 0.f;
        if (co < cout && ci < cin) wino_transform_weight(w, cout, cin, transposed, co, ci, u);
        float4* o4 = reinterpret_cast<float4*>(out + i * 16);
#pragma unroll
        for (int a = 0; a < 4; ++a) o4[a] = make_float4(u[a * 4], u[a * 4 + 1], u[a * 4 + 2], u[a * 4 + 3]);
    }
}

inline unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ULL / d) + 1u; }

}  // namespace

int wino_pack_launch(const float* w, float* packed, int cout, int cin, int transposed, hipStream_t s) {
    if (!w || !packed) return MP_ERR_NULL;
    if (cout <= 0 || cin <= 0) return MP_ERR_SHAPE;
    const int cin_pad4 = (cin + 3) / 4 * 4, cout_pad16 = (cout + 15) / 16 * 16;
    int blocks = (int)(((size_t)cin_pad4 * cout_pad16 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_wino_kernel, dim3(blocks), dim3(256), 0, s, w, packed, cout, cin, cin_pad4, cout_pad16, transposed);
    return check_launch();
}

int wino_configure(const mp_conv_desc* d, WinoLaunch& L) {
    if (!d) return MP_ERR_NULL;
    if (d->n <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return MP_ERR_SHAPE;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad_top != 1 || d->pad_left != 1) return MP_ERR_UNSUPPORTED;
    if (d->conv_h != d->h || d->conv_w != d->w || d->out_h != d->h || d->out_w != d->w) return MP_ERR_UNSUPPORTED;
    if (d->out_mul != 1 || d->out_rep != 1 || d->out_off_y != 0 || d->out_off_x != 0) return MP_ERR_UNSUPPORTED;
    if ((d->w & 1) || (d->h & 1) || (d->cin % kCK) || (d->flags & ~MP_CONV_SHARES_CUS)) return MP_ERR_UNSUPPORTED;
    if ((long long)d->cin * d->h * d->w * 4 >= 0x1FFFFFF0LL || (long long)d->cout * d->h * d->w * 4 >= 0x1FFFFFF0LL) return MP_ERR_UNSUPPORTED;
    WinoParams& p = L.p;
    p.N = d->n; p.Cin = d->cin; p.Cout = d->cout; p.Cout_pad16 = (d->cout + 15) / 16 * 16; p.H = d->h; p.W = d->w;
    p.TW = d->w / 2;
    if (p.TW > kTP) return MP_ERR_UNSUPPORTED;
    p.n_chunks = d->cin / kCK;
    p.tpi = (d->h / 2) * p.TW;
    L.group = (d->w & 3) != 0;
    // two teams (64 output channels per workgroup on one shared input transform) where there are 64 channels to share it, the
    // launch still has a workgroup for every CU, and the CUs are this launch's own: eight waves of 242-256 registers fill a
    // CU's register file, so nothing else is resident beside them - inside a training step (MP_CONV_SHARES_CUS), where
    // BatchNorm launches of sibling streams would otherwise run under the convolution, that costs more than it saves
    L.teams = 1;
    if (!L.group && p.Cout_pad16 >= 64 && !(d->flags & MP_CONV_SHARES_CUS)) {
        const int tr = kTP / p.TW < d->h / 2 ? kTP / p.TW : d->h / 2;
        const long long wgs2 = (long long)((p.Cout_pad16 + 63) / 64) * ((d->h + 2 * tr - 1) / (2 * tr)) * d->n;
        if (wgs2 >= 256) L.teams = 2;
    }
    if (const char* e = knob("MP_WINO_TEAMS")) {  // experiments
        if (atoi(e) == 1) L.teams = 1;
        if (atoi(e) == 2 && !L.group) L.teams = 2;
    }
    p.n_ct = (p.Cout_pad16 + 32 * L.teams - 1) / (32 * L.teams);
    if (L.group) {
        // rows are not 16-byte units: whole small images, staged as contiguous planes (HW % 4 == 0), G of them per workgroup
        if (((d->h * d->w) & 3) || p.tpi > kTP / 2 || (p.tpi & 1)) return MP_ERR_UNSUPPORTED;
        p.G = kTP / p.tpi;
        if (p.G > d->n) p.G = d->n;
        p.TR = d->h / 2;
        p.M = p.G * p.tpi;
        p.R = d->h;
        p.Rin = p.R + 2;
        p.Wp = d->w + 4;  // even: 8-byte patch reads
        p.img_plane = p.Rin * p.Wp;
        p.cin_plane = (p.G * p.img_plane + 3) / 4 * 4;
        p.bands = (d->n + p.G - 1) / p.G;  // image groups
        p.total_blocks = p.n_ct * p.bands;
        p.upr = d->h * d->w / 4;
        p.upc = p.G * p.upr;
    } else {
        p.G = 1;
        p.TR = kTP / p.TW;
        if (p.TR > d->h / 2) p.TR = d->h / 2;
        p.M = p.TR * p.TW;
        p.R = 2 * p.TR;
        p.Rin = p.R + 2;
        p.Wp = d->w + 4;
        p.img_plane = p.Rin * p.Wp;
        p.cin_plane = p.img_plane;
        p.bands = (d->h + p.R - 1) / p.R;
        p.total_blocks = p.n_ct * p.bands * d->n;
        p.upr = d->w / 4;
        p.upc = p.Rin * p.upr;
    }
    if (kCK * p.upc > 3 * 256) return MP_ERR_UNSUPPORTED;
    // tiles per workgroup: about two resident workgroups per CU in ONE round (at most 8 tiles each)
    p.tiles_per_wg = p.total_blocks / (512 / L.teams);
    if (p.tiles_per_wg < 1) p.tiles_per_wg = 1;
    if (p.tiles_per_wg > 8) p.tiles_per_wg = 8;
    if (const char* e = knob("MP_WINO_TILES")) {  // experiments
        const int v = atoi(e);
        if (v >= 1 && v <= 64) p.tiles_per_wg = v;
    }
    p.relu = d->relu;
    p.magic_upr = magic_of(p.upr); p.magic_upc = magic_of(p.upc); p.magic_tw = magic_of(p.TW); p.magic_pairs = magic_of(p.M >> 1);
    p.magic_tpi = magic_of(p.tpi); p.magic_w = magic_of(p.W);
    const size_t raw_bytes = (size_t)2 * (kCK * p.cin_plane + 4) * 4, v_bytes = (size_t)2 * kVFloats * 4, x_bytes = (size_t)L.teams * kXFloats * 4;
    L.lds_bytes = raw_bytes + (v_bytes > x_bytes ? v_bytes : x_bytes);
    L.ni = (kCK * p.upc + 256 * L.teams - 1) / (256 * L.teams);
    if (L.lds_bytes > 150 * 1024) return MP_ERR_UNSUPPORTED;
    return MP_OK;
}

int wino_launch(const WinoLaunch& L0, hipStream_t s) {
    WinoLaunch L = L0;
    L.p.dbg = conv_stamp_buffer((size_t)L.p.total_blocks * 64);
    auto go = [&](auto kern) {
        static AttrOnce attr_once;
        if (attr_once.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(kern, dim3((L.p.total_blocks + L.p.tiles_per_wg - 1) / L.p.tiles_per_wg), dim3(256 * L.teams), L.lds_bytes, s, L.p);
        return check_launch();
    };
    // QROW: the four tiles of an epilogue item are eight consecutive pixels of two rows
    const bool qrow = !L.group && L.p.M == kTP && L.p.TW % 4 == 0 && L.p.H % L.p.R == 0;
    if (L.group) {
        switch (L.ni) {
            case 1: return go(conv_wino_f32_kernel<1, false, true, 1>);
            case 2: return go(conv_wino_f32_kernel<2, false, true, 1>);
            case 3: return go(conv_wino_f32_kernel<3, false, true, 1>);
            default: return MP_ERR_UNSUPPORTED;
        }
    }
    if (L.teams == 2) {
        switch (L.ni) {
            case 1: return qrow ? go(conv_wino_f32_kernel<1, true, false, 2>) : go(conv_wino_f32_kernel<1, false, false, 2>);
            case 2: return qrow ? go(conv_wino_f32_kernel<2, true, false, 2>) : go(conv_wino_f32_kernel<2, false, false, 2>);
            case 3: return qrow ? go(conv_wino_f32_kernel<3, true, false, 2>) : go(conv_wino_f32_kernel<3, false, false, 2>);
            default: return MP_ERR_UNSUPPORTED;
        }
    }
    switch (L.ni) {
        case 1: return qrow ? go(conv_wino_f32_kernel<1, true, false, 1>) : go(conv_wino_f32_kernel<1, false, false, 1>);
        case 2: return qrow ? go(conv_wino_f32_kernel<2, true, false, 1>) : go(conv_wino_f32_kernel<2, false, false, 1>);
        case 3: return qrow ? go(conv_wino_f32_kernel<3, true, false, 1>) : go(conv_wino_f32_kernel<3, false, false, 1>);
        default: return MP_ERR_UNSUPPORTED;
    }
}

}  // namespace mp

using namespace mp;

extern "C" {

size_t mp_conv_winograd_packed_weight_bytes(int cout, int cin) {
    if (cout <= 0 || cin <= 0) return 0;
    return (size_t)((cin + 3) / 4 * 4) * 16 * ((cout + 15) / 16 * 16) * sizeof(float);
}

int mp_conv_winograd_pack_weight(const float* w, float* packed, int cout, int cin, mp_stream_t stream) {
    return wino_pack_launch(w, packed, cout, cin, 0, as_stream(stream));
}

int mp_conv_winograd_supported(const mp_conv_desc* desc) {
    WinoLaunch L{};
    return wino_configure(desc, L);
}

int mp_conv2d_winograd_fwd(const mp_conv_desc* desc, const float* x, const float* packed_u, const float* scale, const float* shift,
                           const float* res1, const float* res2, float* out, mp_stream_t stream) {
    WinoLaunch L{};
    int rc = wino_configure(desc, L);
    if (rc != MP_OK) return rc;
    if (!x || !packed_u || !scale || !shift || !out) return MP_ERR_NULL;
    L.p.x = x; L.p.u = packed_u; L.p.scale = scale; L.p.shift = shift; L.p.res1 = res1; L.p.res2 = res2; L.p.out = out;
    return wino_launch(L, as_stream(stream));
}

}  // extern "C"
