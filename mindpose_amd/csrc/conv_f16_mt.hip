// Persistent multi-tile form of the fp16-MFMA convolution: a workgroup loads its weight slice (all input channels x
// taps x CT couts) into LDS ONCE and then walks a run of pixel tiles, double-buffering the input tiles - tile t+1 is
// fetched into registers while tile t runs on the matrix cores, its residuals are fetched before its MFMA loop, its
// results leave as 8-byte stores while the next tile's data lands in LDS; one barrier per tile.
// Compared with conv_f16_kernel (one tile per workgroup) the weight traffic from L2 drops by the run length and the
// load / MFMA / store phases overlap inside a workgroup instead of only across co-resident ones.
#include "conv_f16.h"
#include "conv_f16_dev.h"

namespace mp {

namespace {

template <int KS, int S, int PS, int CS, int WAVES_P, int WAVES_C, int NI, int OCC, int STATS = 0>
__global__ __launch_bounds__(256, OCC) void conv_f16_mt_kernel(const ConvF16Params p) {
    static_assert(WAVES_P * WAVES_C == 4, "4 waves per workgroup");
    constexpr int T = KS * KS;
    constexpr int CT = 16 * CS * WAVES_C;
    extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
    u32x4* __restrict__ lds_w = smem16;             // [PK/4][T][4][CT]
    u32x4* __restrict__ lds_in = smem16 + p.w_buf;  // [nbuf][in_buf]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp_i = wave % WAVES_P, wc_i = wave / WAVES_P;
    const int lq = lane >> 4, lr = lane & 15;

    int b = blockIdx.x;
    {
        const int nb = p.total_blocks, q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, j = b >> 3;
        b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const int ct = b % p.n_ct;
    const int grp = b / p.n_ct;
    const int t_begin = grp * p.tiles_per_wg;
    const int t_end = min(t_begin + p.tiles_per_wg, p.tiles_total);
    const int HW = p.H * p.W;
    const int plane_o = p.out_h * p.out_w;

    // ---- weights: the whole K x CT slice, once
    {
        const __amdgpu_buffer_rsrc_t rs_w =
            make_rsrc(reinterpret_cast<const char*>(p.wp) + (p.phases > 1 ? (size_t)blockIdx.y * p.w_phase_bytes : (size_t)0),
                      (size_t)p.PK * T * p.Cout_pad16 * 16);
        const int w_units = p.PK * T * CT;
        for (int u0 = 0; u0 < w_units; u0 += 256 * 8) {
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int u = u0 + tid + 256 * i;
                const int row = u / CT, c = u - row * CT;
                const bool ok = u < w_units && ct * CT + c < p.Cout_pad16;
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, ok ? (unsigned)(row * p.Cout_pad16 + ct * CT + c) * 16u : kOob, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int u = u0 + tid + 256 * i;
                if (u < w_units) lds_w[u] = v[i];
            }
        }
    }
    {
        const int n16 = p.nbuf * p.in_buf;
        const u32x4 zero = (u32x4){0u, 0u, 0u, 0u};
        for (int i = tid; i < n16; i += 256) lds_in[i] = zero;
    }

    // ---- tile-independent staging tables: source offset relative to the tile's first row of its first image, LDS
    //      destination, and (image, row) inside the tile for the per-tile range check
    int uoff[NI], udst[NI], ugr[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const unsigned u = tid + 256 * i;
        uoff[i] = 0;
        udst[i] = -1;
        ugr[i] = 0;
        if (u < (unsigned)(p.PKs * p.upc)) {
            const unsigned pl = fastdiv(u, p.upc, p.magic_upc);
            const unsigned rem = u - pl * p.upc;
            const unsigned gr = fastdiv(rem, p.ncols, p.magic_ncols);
            const unsigned xu = rem - gr * p.ncols;
            const unsigned g = p.G > 1 ? fastdiv(gr, p.Rin, p.magic_rin) : 0u;
            const unsigned r = gr - g * p.Rin;
            uoff[i] = (int)(((g * p.C8in + pl) * HW + r * p.W + xu) * 16u);
            udst[i] = (int)(pl * p.plane + g * p.img_plane + r * p.Wp + p.pad_l + xu);
            ugr[i] = (int)((g << 16) | r);
        }
    }

    const int off_y = p.phases > 1 ? (int)(blockIdx.y >> 1) : p.off_y, off_x = p.phases > 1 ? (int)(blockIdx.y & 1) : p.off_x;
    int b_off[PS];
    int pix_rel[PS], pix_gy[PS];  // output offset of the lane's pixel relative to the tile origin; (image << 16 | row), -1 = padding lane
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
        const unsigned pl0 = (unsigned)((wp_i * PS + ps) * 16 + lr);
        const unsigned pl = pl0 < (unsigned)(p.G * p.RWo) ? pl0 : 0u;
        const unsigned g = fastdiv(pl, p.RWo, p.magic_rwo);
        const unsigned rem = pl - g * p.RWo;
        const unsigned y = fastdiv(rem, p.Wo, p.magic_wo);
        const unsigned xx = rem - y * p.Wo;
        b_off[ps] = lq * p.plane + g * p.img_plane + (y * S) * p.Wp + xx * S;
        pix_rel[ps] = (int)((g * p.C8out * plane_o + (y * p.out_mul + off_y) * p.out_w + xx * p.out_mul + off_x) * 16u);
        pix_gy[ps] = pl0 < (unsigned)(p.G * p.RWo) ? (int)((g << 16) | y) : -1;
    }
    int a_off[CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) a_off[cs] = lq * CT + wc_i * CS * 16 + f16_a_row<CS>(cs, lr);

    f32x4 sc[CS], sh[CS];
    unsigned co_off[CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        const int co = ct * CT + wc_i * CS * 16 + f16_d_cout<CS>(cs, lq);
        const bool ok = co < p.C8out * 8;
        const int cc = co < p.Cout_pad16 ? co : 0;
        sc[cs] = *reinterpret_cast<const f32x4*>(p.scale + cc);
        sh[cs] = *reinterpret_cast<const f32x4*>(p.shift + cc);
        co_off[cs] = ok ? (unsigned)(co >> 3) * plane_o * 16u + ((co >> 2) & 1) * 8u : kInv;
    }

    const size_t x_bytes = (size_t)p.N * p.C8in * HW * 16, o_bytes = (size_t)p.N * p.C8out * plane_o * 16;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, x_bytes);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, o_bytes);
    const __amdgpu_buffer_rsrc_t rs_r1 = make_rsrc(p.res1 ? p.res1 : p.out, o_bytes);
    const __amdgpu_buffer_rsrc_t rs_r2 = make_rsrc(p.res2 ? p.res2 : p.out, o_bytes);

    u32x4 vin[NI];
    auto stage_load = [&](int t) {
        const int ty = t % p.tiles_y, tn = t / p.tiles_y;
        const int n0 = tn * p.G, y_in0 = ty * p.R * S - p.pad_t;
        const int base = (n0 * p.C8in * HW + y_in0 * p.W) * 16;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i >= p.ni_used) break;  // uniform: slots beyond this shape's unit count issue nothing
            const int yin = y_in0 + (ugr[i] & 0xFFFF), n = n0 + (ugr[i] >> 16);
            const bool ok = udst[i] >= 0 && yin >= 0 && yin < p.H && n < p.N;
            vin[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(base + uoff[i]) : kOob, 0, 0);
        }
    };
    auto stage_store = [&](int buf) {  // rows / images outside the tensor arrive as zeros and overwrite the last tile's data
        u32x4* __restrict__ din = lds_in + buf * p.in_buf;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i >= p.ni_used) break;
            if (udst[i] >= 0) din[udst[i]] = vin[i];
        }
    };

    stage_load(t_begin);
    __syncthreads();  // weights + zero fill complete
    stage_store(0);
    __syncthreads();

    // epilogue statistics (conv_f16_dev.h): the per-lane sums run over ALL tiles of the workgroup - one partial per workgroup
    constexpr int NPz = CS / 2;
    f32x4 st_a[STATS ? CS : 1], st_b[STATS ? CS : 1];
    if constexpr (STATS) {
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            st_a[cs] = st_b[cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const __amdgpu_buffer_rsrc_t rs_z = make_rsrc((STATS && p.st_z) ? p.st_z : p.out, (STATS && p.st_z) ? o_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc((STATS && p.st_y) ? p.st_y : p.out, (STATS && p.st_y) ? o_bytes : 0);

    const int nq = p.PK >> 2;
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = p.nbuf == 2 ? ((t - t_begin) & 1) : 0;
        const bool more = t + 1 < t_end;
        if (more) stage_load(t + 1);

        // output addresses of this tile, residuals fetched now (their latency hides under the MFMA loop)
        const int ty = t % p.tiles_y, tn = t / p.tiles_y;
        const int n0 = tn * p.G, y0 = ty * p.R;
        const int obase = (n0 * p.C8out * plane_o + y0 * p.out_mul * p.out_w) * 16;
        unsigned pix_off[PS];
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const bool ok = pix_gy[ps] >= 0 && n0 + (pix_gy[ps] >> 16) < p.N && y0 + (pix_gy[ps] & 0xFFFF) < p.Ho;
            pix_off[ps] = ok ? (unsigned)(obase + pix_rel[ps]) : kInv;
        }
        // cout tiles 2j, 2j+1 share a 16-byte channel block per pixel (conv_f16_dev.h): 16-byte residual loads and stores
        constexpr int NP = CS / 2, NS = CS - 2 * NP;
        u32x4 r1p[NP ? NP : 1][PS], r2p[NP ? NP : 1][PS];
        u32x2 r1s[PS], r2s[PS];
        if (p.res1) {
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) r1p[j][ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_r1, co_off[2 * j] + pix_off[ps], 0, 0);
            if (NS) {
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) r1s[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs_r1, co_off[CS - 1] + pix_off[ps], 0, 0);
            }
        }
        if (p.res2) {
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) r2p[j][ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_r2, co_off[2 * j] + pix_off[ps], 0, 0);
            if (NS) {
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) r2s[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs_r2, co_off[CS - 1] + pix_off[ps], 0, 0);
            }
        }

        u32x4 zp[(STATS && NPz) ? NPz : 1][PS], yp[(STATS && NPz) ? NPz : 1][PS];
        u32x2 zs[PS], ys[PS];
        if constexpr (STATS) {
            if constexpr (STATS == 2) {  // the BatchNorm's z (and y) at this tile's output positions, fetched like the residuals
#pragma unroll
                for (int j = 0; j < NPz; ++j)
#pragma unroll
                    for (int ps = 0; ps < PS; ++ps) {
                        zp[j][ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, co_off[2 * j] + pix_off[ps], 0, 0);
                        yp[j][ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, co_off[2 * j] + pix_off[ps], 0, 0);
                    }
                if (CS & 1) {
#pragma unroll
                    for (int ps = 0; ps < PS; ++ps) {
                        zs[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs_z, co_off[CS - 1] + pix_off[ps], 0, 0);
                        ys[ps] = __builtin_amdgcn_raw_buffer_load_b64(rs_y, co_off[CS - 1] + pix_off[ps], 0, 0);
                    }
                }
            }
        }

        f32x4 acc[PS][CS];
#pragma unroll
        for (int ps = 0; ps < PS; ++ps)
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) acc[ps][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const u32x4* __restrict__ lin = lds_in + buf * p.in_buf;
            const u32x4* __restrict__ lw = lds_w;
            u32x4 bv[PS], av[CS];
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) bv[ps] = lin[b_off[ps]];
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) av[cs] = lw[a_off[cs]];
            for (int q = 0; q < nq; ++q) {
                const int in_q = q * 4 * p.plane;
                const int w_q = q * T * 4 * CT;
                const int qn = min(q + 1, nq - 1);
#pragma unroll
                for (int tp = 0; tp < T; ++tp) {
                    const int tn2 = (tp + 1 < T) ? tp + 1 : 0;
                    const int in_off = ((tp + 1 < T) ? in_q : qn * 4 * p.plane) + (tn2 / KS) * p.Wp + (tn2 % KS);
                    const int w_off = ((tp + 1 < T) ? w_q : qn * T * 4 * CT) + tn2 * 4 * CT;
                    u32x4 bn[PS], an[CS];
#pragma unroll
                    for (int cs = 0; cs < CS; ++cs) an[cs] = lw[a_off[cs] + w_off];  // weights first: the next step's first MFMA needs them
#pragma unroll
                    for (int ps = 0; ps < PS; ++ps) bn[ps] = lin[b_off[ps] + in_off];
#pragma unroll
                    for (int ps = 0; ps < PS; ++ps)
#pragma unroll
                        for (int cs = 0; cs < CS; ++cs)
                            acc[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[cs]),
                                                                                  __builtin_bit_cast(f16x8, bv[ps]), acc[ps][cs], 0, 0, 0);
                    {
                        constexpr int NR = PS + CS, NM = PS * CS, NPAIR = NR < NM ? NR : NM;
#pragma unroll
                        for (int i = 0; i < NPAIR; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        if (NM > NPAIR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NPAIR, 0);
                        if (NR > NPAIR) __builtin_amdgcn_sched_group_barrier(0x100, NR - NPAIR, 0);
                    }
#pragma unroll
                    for (int ps = 0; ps < PS; ++ps) bv[ps] = bn[ps];
#pragma unroll
                    for (int cs = 0; cs < CS; ++cs) av[cs] = an[cs];
                }
            }
        }

        const bool has1 = p.res1 != nullptr, has2 = p.res2 != nullptr;
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) {
                const u32x4 a1 = has1 ? r1p[j][ps] : (u32x4){0u, 0u, 0u, 0u}, a2 = has2 ? r2p[j][ps] : (u32x4){0u, 0u, 0u, 0u};
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
                u32x2 o = f16_pack4(f16_epi4(acc[ps][CS - 1], sc[CS - 1], sh[CS - 1], has1, has1 ? r1s[ps] : (u32x2){0u, 0u}, has2,
                                             has2 ? r2s[ps] : (u32x2){0u, 0u}, p.relu));
                if constexpr (STATS) {
                    const bool valid = pix_off[ps] != kInv;
                    if constexpr (STATS == 2) f16_stats_acc<2>(o, valid, st_a[CS - 1], st_b[CS - 1], zs[ps], ys[ps], p.st_relu);
                    else f16_stats_acc<1>(o, valid, st_a[CS - 1], st_b[CS - 1], o, o, 0);
                }
                __builtin_amdgcn_raw_buffer_store_b64(o, rs_o, co_off[CS - 1] + pix_off[ps], 0, 0);
            }
        }

        if (more) {
            if (p.nbuf == 1) __syncthreads();  // single input buffer (big tiles): every wave is done reading it
            stage_store(p.nbuf == 2 ? (buf ^ 1) : 0);
            __syncthreads();
        }
    }
    if constexpr (STATS)  // the scratch lies over the weight slice (the flush's first barrier: every wave is past its last MFMA loop)
        f16_stats_flush<CS, WAVES_P, WAVES_C>(st_a, st_b, reinterpret_cast<float*>(smem16), p.st_part, p.st_nparts, (p.phases > 1 ? (int)blockIdx.y * p.n_groups : 0) + grp, ct * CT, p.C8out,
                                              wp_i, wc_i, lq, lr);
}

constexpr int mt_ni(int occ) { return occ == 1 ? 12 : 8; }

template <int KS, int S, int PS, int CS, int WAVES_P, int WAVES_C, int OCC, int STATS>
int launch_mt_kernel(const ConvF16Params& p, size_t lds_bytes, hipStream_t s) {
    if (g_dry_launch) return MP_OK;  // mp_f16_conv_supported: the dispatch alone
    auto kern = conv_f16_mt_kernel<KS, S, PS, CS, WAVES_P, WAVES_C, mt_ni(OCC), OCC, STATS>;
    static AttrOnce attr_set_once;
    if (attr_set_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(p.total_blocks, p.phases > 1 ? p.phases : 1), dim3(256), lds_bytes, s, p);
    return check_launch();
}

template <int KS, int S, int PS, int CS, int WAVES_P, int WAVES_C, int OCC>
int launch_mt_variant(const ConvF16Params& p, size_t lds_bytes, hipStream_t s) {
    if (p.st_mode != 0) {  // training builds with epilogue statistics: the kernel sizes a BatchNorm follows
        if constexpr (KS == 1 || KS == 3) {
            if (p.st_mode == 1) return launch_mt_kernel<KS, S, PS, CS, WAVES_P, WAVES_C, OCC, 1>(p, lds_bytes, s);
            if constexpr (S == 1) return launch_mt_kernel<KS, S, PS, CS, WAVES_P, WAVES_C, OCC, 2>(p, lds_bytes, s);
        }
        if constexpr (KS == 2 && S == 1) {  // the phase convs of a stride-2 data gradient: backward sums only
            if (p.st_mode == 2) return launch_mt_kernel<KS, S, PS, CS, WAVES_P, WAVES_C, OCC, 2>(p, lds_bytes, s);
        }
        return MP_ERR_UNSUPPORTED;
    }
    return launch_mt_kernel<KS, S, PS, CS, WAVES_P, WAVES_C, OCC, 0>(p, lds_bytes, s);
}

template <int KS, int S, int OCC>
int launch_mt_shape(const ConvF16Params& p, int shape, size_t lds_bytes, hipStream_t s) {
    switch (shape) {
        case F_CT32_PT192: return launch_mt_variant<KS, S, 3, 2, 4, 1, OCC>(p, lds_bytes, s);
        case F_CT64_PT192: return launch_mt_variant<KS, S, 3, 4, 4, 1, OCC>(p, lds_bytes, s);
        case F_CT48_PT192: return launch_mt_variant<KS, S, 3, 3, 4, 1, OCC>(p, lds_bytes, s);
        case F_CT64_PT96: return launch_mt_variant<KS, S, 3, 2, 2, 2, OCC>(p, lds_bytes, s);
        case F_CT32_PT96: return launch_mt_variant<KS, S, 3, 1, 2, 2, OCC>(p, lds_bytes, s);
        case 5: return launch_mt_variant<KS, S, 3, 1, 4, 1, OCC>(p, lds_bytes, s);  // 16 couts x 192 pixels
        default: return MP_ERR_UNSUPPORTED;
    }
}

template <int KS, int S>
int launch_mt_ks(const ConvF16Params& p, int variant, size_t lds_bytes, hipStream_t s) {
    const int shape = variant >= F_CT16_PT192_MT2 ? 5 : (variant - F_MT2_BASE) % 5;
    return f16_variant_mt_occ(variant) == 1 ? launch_mt_shape<KS, S, 1>(p, shape, lds_bytes, s)
                                            : launch_mt_shape<KS, S, 2>(p, shape, lds_bytes, s);
}

}  // namespace

int f16_mt_ni(int occ) { return mt_ni(occ); }

int f16_mt_launch(const ConvF16Launch& L, hipStream_t s) {
    if (L.ks == 3) return L.stride == 1 ? launch_mt_ks<3, 1>(L.p, L.variant, L.lds_bytes, s) : launch_mt_ks<3, 2>(L.p, L.variant, L.lds_bytes, s);
    if (L.ks == 1 && L.stride == 1) return launch_mt_ks<1, 1>(L.p, L.variant, L.lds_bytes, s);
    if (L.ks == 2 && L.stride == 1) return launch_mt_ks<2, 1>(L.p, L.variant, L.lds_bytes, s);
    return MP_ERR_UNSUPPORTED;
}

}  // namespace mp
