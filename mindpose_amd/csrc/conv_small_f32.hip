// fp32 3x3 (stride 1 / 2, pad 1) and 1x1 convolution for SMALL problems - a handful of crops (N in {1, 32}; a top-down pipeline serves
// the people of one frame) - HRNet's branch convs hrnet.py:51-64, 202-241 in eval mode with the BatchNorm folded.
//
// Why another form: at N = 1 the 128 / 256-channel layers are 48 - 192 output pixels against 1.2 - 2.4 MB of weights.  The tile kernels
// (32 - 64 couts x 96 - 192 pixels, the whole k loop in one workgroup) and the Winograd kernel put such a layer on FOUR to EIGHT
// workgroups for 32 - 60 us (profiles/r05_d_timeline_infer_f32_n1.json: 80 of those launches are 3.2 of the step's 4.6 ms of kernel
// time).  Here the layer is cut the other way: a workgroup = 16 couts x 16 consecutive pixels of one image, and its eight waves SPLIT K
// (wave w takes the cin quads q = w, w + 8, ...: the same 16 x 16 tile, an eighth of the 9 Cin / 4 k-steps each), folded through LDS in
// wave order (bit-reproducible).  256 -> 256 @8x6, one crop: 48 workgroups of 72 MFMAs per wave instead of 8 of 3456.
//   * A operand (weights, v_mfma_f32_16x16x4_f32: lane = (cout row l % 16, k l / 16)) straight from the direct kernel's packing
//     [cin quad][tap][4][Cout_pad16] - 16 consecutive floats per k row, one dword per lane and k-step, eight k-steps in flight;
//   * B operand (pixels) from an LDS image of the rows the tile touches (+ halo, zeros outside the map) of ALL input channels,
//     staged once per workgroup: [cin][row][W + 2], plane pitch = 16 mod 32 floats (the four k rows of a step hit four bank groups);
//   * epilogue by wave 0: scale / shift, up to two residuals, ReLU, 64-byte runs of one cout's 16 pixels.
// The tuner times it as fp32 variant 11 next to the direct, Winograd and GEMM forms; it wins where the launch would otherwise
// cover a few CUs, and loses (many workgroups re-staging the same rows) where the other forms fill the chip.
#include "conv_small.h"

namespace mp {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ULL / d) + 1u; }
__device__ __forceinline__ unsigned fdiv(unsigned e, unsigned d, unsigned magic) { return d == 1 ? e : __umulhi(e, magic); }

// weight k-steps in flight per wave = four cin quads x the taps (3x3: 36 dwords per lane, 9 KB per wave - the k loop is a weight
// STREAM: with eight in flight the 256-channel layer took 44 us)

#ifndef MP_SMALL_ABLATE
#define MP_SMALL_ABLATE 0  // diagnostic builds (tools/variant_builds.sh): 1 = no staging, 2 = no MFMA loop, 4 = no weight loads, 8 = no fold; results wrong, timings meaningful
#endif

constexpr int kWaves = 8;  // K is split eight ways: the 256-channel layer's 576 k-steps = 72 per wave = two fills of the weight ring

// PT = 16-pixel tiles per workgroup (consecutive pixels of one image): 1 for a handful of crops (most workgroups); 3 / 4 when the
// batch is large enough that every workgroup re-streaming its 16 couts' weights becomes the cost (N = 8 ... 32: the weights of a
// workgroup then serve 48 / 64 pixels)
template <int KS, int S, int PT>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(4, 8))) void conv_small_f32_kernel(const SmallParams p) {
    constexpr int T = KS * KS, PAD = KS / 2;
    // cin quads of weights in flight per wave: four (36 dwords for 3x3; three with four accumulator tiles) - the budget is 128
    // registers = TWO workgroups per CU (one: 128 -> 128 @16x12, N = 32 took 47 us on 768 workgroups; now 32)
    constexpr int RQ = (PT == 4 && KS == 3) ? 3 : 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;  // MFMA row / column index, k index (inputs) = row group (outputs)

    // block -> (cout tile, image, pixel tile): the cout tiles of one pixel tile are neighbours (they stage the same rows: one L2)
    const unsigned b = blockIdx.x;
    const unsigned pt = fdiv(b, (unsigned)p.n_ct, p.magic_nct);
    const int ct = (int)(b - pt * p.n_ct);
    const unsigned n = fdiv(pt, (unsigned)p.tiles_img, p.magic_tiles);
    const int tile = (int)(pt - n * p.tiles_img);
    const int p0 = tile * 16 * PT;                   // first pixel of the workgroup's tiles (flattened y * W + x)
    const int y_first = (int)fdiv((unsigned)p0, (unsigned)p.W, p.magic_w);
    const int row0 = S * y_first - PAD;              // input row of staged row 0

    // ---- this lane's pixels (B operand column lr of every tile) and weight row (A operand row lr)
    int px[PT];
    unsigned b_base[PT];  // window origin (tap 0 = input row S py - PAD, column S pxx - PAD) in the staged plane (column 0 = input column -PAD)
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        px[i] = p0 + 16 * i + lr;
        const int pv = px[i] < p.HW ? px[i] : p0;
        const int py = (int)fdiv((unsigned)pv, (unsigned)p.W, p.magic_w);
        const int pxx = pv - py * p.W;
        b_base[i] = (unsigned)(lk * p.plane + (S * py - PAD - row0) * p.Wp + S * pxx);
    }
    const float* __restrict__ wrow = p.wp + (size_t)lk * p.Cout_pad16 + ct * 16 + lr;  // + ((q * T + t) * 4) * Cout_pad16 per k-step
    const unsigned w_step = 4u * (unsigned)p.Cout_pad16;

    // (requested BEFORE the staging loop: the first fill of the weight ring flies under it)
    // this wave's k-steps: cin quads q = wave, wave + 8, ... ; per quad the nine taps.  Four quads (36 k-steps) of weights are in
    // flight: slot (j, t) holds tap t of the wave's quad qi0 + j and is refilled for quad qi0 + j + 4 right behind its MFMA
    // (every index below is a compile-time constant: the ring stays in registers)
    const int nq = (p.kq - wave + kWaves - 1) / kWaves;  // quads of this wave
    float a_reg[RQ * T];
    auto w_at = [&](int qi, int t) __attribute__((always_inline)) {
        if (MP_SMALL_ABLATE & 4) return (float)(qi + t);
        return wrow[(size_t)((wave + kWaves * qi) * T + t) * w_step];
    };
#pragma unroll
    for (int j = 0; j < RQ; ++j)
#pragma unroll
        for (int t = 0; t < T; ++t) a_reg[j * T + t] = j < nq ? w_at(j, t) : 0.f;
    // ---- stage: rows row0 .. row0 + rows - 1, columns -PAD .. (zeros outside the map).  Wave w reads ONLY the channels of its own cin
    //      quads (q = w, w + 8, ...), so every wave stages exactly those planes itself - no workgroup barrier in front of the k loop.
    //      Every plane has the same geometry: a lane decodes its slot (element lane + 64 s of a plane -> source offset, in / out of the
    //      map) ONCE and then walks the wave's planes with one add per element, sixteen loads in flight (decoding every element - two
    //      divisions, the bounds, a 64-bit address - made the staging a ~800-instruction VALU phase per lane: as long as the MFMA phase)
    {
        const float* __restrict__ xin = p.x + (size_t)n * p.Cin * p.HWin;
        const int per_plane = p.rows * p.Wp;
        const int n_pl = nq * 4;  // planes of this wave (the last quad may run past Cin: zeros)
        constexpr int U = 16;  // (32 in flight - two slots x 16 planes, or 32 planes - measured slower: 256 -> 256 @8x6, N = 32: 31.8 -> 34.3 us)
        for (int e = lane; e < ((MP_SMALL_ABLATE & 1) ? 0 : per_plane); e += 64) {
            const unsigned r = fdiv((unsigned)e, (unsigned)p.Wp, p.magic_wp);
            const int c = (int)((unsigned)e - r * p.Wp) - PAD, y = row0 + (int)r;
            const bool in_map = y >= 0 && y < p.Hin && c >= 0 && c < p.Win;
            const int src = in_map ? y * p.Win + c : 0;
            for (int k0 = 0; k0 < n_pl; k0 += U) {
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + u;  // plane k of the wave = channel (wave + 8 (k >> 2)) 4 + (k & 3)
                    const int ci = (wave + kWaves * (k >> 2)) * 4 + (k & 3);
                    v[u] = (in_map && k < n_pl && ci < p.Cin) ? xin[(size_t)ci * p.HWin + src] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + u;
                    const int ci = (wave + kWaves * (k >> 2)) * 4 + (k & 3);
                    if (k < n_pl) smem[ci * p.plane + e] = v[u];
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // this wave's own LDS stores above are read back by its other lanes
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    f32x4 acc[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int qi0 = 0; qi0 < ((MP_SMALL_ABLATE & 2) ? 0 : nq); qi0 += RQ) {
#pragma unroll
        for (int j = 0; j < RQ; ++j) {
            const int qi = qi0 + j;
            if (qi < nq) {  // wave-uniform
                const unsigned q_off = (unsigned)((wave + kWaves * qi) * 4 * p.plane);
                // pixel operands one tap ROW at a time (the wide forms: all nine taps of four tiles would be 36 registers)
                constexpr int TR = PT >= 4 ? KS : 1;       // tap rows per operand batch
                constexpr int TB = T / TR;                 // taps per batch
#pragma unroll
                for (int tr = 0; tr < TR; ++tr) {
                    float bv[PT][TB];
#pragma unroll
                    for (int i = 0; i < PT; ++i)
#pragma unroll
                        for (int tb = 0; tb < TB; ++tb) {
                            const int t = tr * TB + tb;
                            bv[i][tb] = smem[b_base[i] + q_off + (unsigned)((t / KS) * p.Wp + (t % KS))];
                        }
#pragma unroll
                    for (int tb = 0; tb < TB; ++tb) {
                        const int t = tr * TB + tb;
                        const float av = a_reg[j * T + t];
                        if (qi + RQ < nq) a_reg[j * T + t] = w_at(qi + RQ, t);
#pragma unroll
                        for (int i = 0; i < PT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[i][tb], acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- fold the K parts through LDS in wave order (the staged image is dead: every wave has passed its last read)
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < PT; ++i) red[((wave - 1) * PT + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (wave != 0) return;
    // (a ROLLED loop: unrolled, the compiler issues all 7 PT partial-tile reads at once and sinks the adds into the epilogue - 112
    //  registers for the wide forms, which then spill or drop to one workgroup per CU)
#pragma unroll 1
    for (int k = 0; k < ((MP_SMALL_ABLATE & 8) ? 0 : kWaves - 1); ++k)  // wave 1, 2, ... in order
#pragma unroll
        for (int i = 0; i < PT; ++i) acc[i] = acc[i] + red[(k * PT + i) * 64 + lane];
    // ---- epilogue: lane = (pixel column lr, cout rows 4 lk .. 4 lk + 3)
    const int co0 = ct * 16 + lk * 4;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        if (px[i] >= p.HW) continue;
        const size_t o_img = (size_t)n * p.Cout * p.HW + px[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + j;
            if (co >= p.Cout) break;
            float v = acc[i][j] * p.scale[co] + p.shift[co];
            const size_t o = o_img + (size_t)co * p.HW;
            if (p.res1) v += p.res1[o];
            if (p.res2) v += p.res2[o];
            if (p.relu) v = fmaxf(v, 0.f);
            p.out[o] = v;
        }
        if (PT >= 3) __builtin_amdgcn_sched_barrier(0);  // (one tile's residual loads at a time)
    }
}

}  // namespace

int small_configure(const mp_conv_desc* d, SmallLaunch& L, int wide) {
    if (!d) return MP_ERR_NULL;
    if (d->n <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return MP_ERR_SHAPE;
    // built forms: 3x3 pad 1 stride 1 / 2 (branch convs; transition / exchange-unit down-paths), 1x1 stride 1 (exchange-unit up-paths, head)
    const bool k3 = d->kh == 3 && d->kw == 3 && (d->stride == 1 || d->stride == 2) && d->pad_top == 1 && d->pad_left == 1;
    const bool k1 = d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_top == 0 && d->pad_left == 0;
    if (!k3 && !k1) return MP_ERR_UNSUPPORTED;
    const int ho = (d->h + 2 * d->pad_top - d->kh) / d->stride + 1, wo = (d->w + 2 * d->pad_left - d->kw) / d->stride + 1;
    if (d->conv_h != ho || d->conv_w != wo || d->out_h != ho || d->out_w != wo) return MP_ERR_UNSUPPORTED;
    if (d->out_mul != 1 || d->out_rep != 1 || d->out_off_y != 0 || d->out_off_x != 0 || d->flags) return MP_ERR_UNSUPPORTED;
    SmallParams& p = L.p;
    p.N = d->n; p.Cin = d->cin; p.Cin_pad4 = (d->cin + 3) / 4 * 4; p.Cout = d->cout; p.Cout_pad16 = (d->cout + 15) / 16 * 16;
    p.H = ho; p.W = wo; p.HW = ho * wo;
    p.Hin = d->h; p.Win = d->w; p.HWin = d->h * d->w;
    p.ks = d->kh; p.stride = d->stride; p.pad = d->pad_top;
    p.Wp = d->stride * (wo - 1) + d->kh;
    // pixel tiles per workgroup: one; the wide form: a whole 48-pixel map (three) or 64 pixels (four)
    p.pt = !wide ? 1 : (ho * wo <= 48 ? 3 : 4);
    if (wide && ho * wo <= 16) return MP_ERR_UNSUPPORTED;
    int span = (16 * p.pt + wo - 2) / wo + 1;  // output rows the workgroup's consecutive pixels can touch
    if (span > ho) span = ho;
    p.rows = d->stride * (span - 1) + d->kh;
    const int raw = p.rows * p.Wp;
    int plane = (raw + 15) / 32 * 32 + 16;  // smallest value >= raw that is 16 (mod 32)
    if (plane - 32 >= raw) plane -= 32;
    p.plane = plane;
    p.tiles_img = (p.HW + 16 * p.pt - 1) / (16 * p.pt);
    p.n_ct = p.Cout_pad16 / 16;
    p.kq = p.Cin_pad4 / 4;
    p.relu = d->relu;
    p.magic_w = magic_of((unsigned)p.W);
    p.magic_wp = magic_of((unsigned)p.Wp);
    p.magic_tiles = magic_of((unsigned)p.tiles_img);
    p.magic_nct = magic_of((unsigned)p.n_ct);
    p.magic_per_plane = magic_of((unsigned)(p.rows * p.Wp));
    const long long blocks = (long long)p.N * p.tiles_img * p.n_ct;
    // what the form is for: launches the tile kernels would put on a few CUs.  Beyond ~1 k workgroups the other forms fill the
    // chip and stage each input row once instead of once per cout tile - and the tuner, which times a launch ALONE, would pick this
    // form for layers where it only wins alone (same-box A/B at N = 32: 5.20 ms without it, 5.43 ms with the bound at 2 k or 16 k)
    if (blocks > (wide ? 2048 : 1024)) return MP_ERR_UNSUPPORTED;
    if ((long long)p.N * p.Cin * p.HWin >= (1LL << 31) || (long long)p.N * p.Cout * p.HW >= (1LL << 31)) return MP_ERR_UNSUPPORTED;
    L.grid = (int)blocks;
    L.lds_bytes = (size_t)p.Cin_pad4 * p.plane * 4;
    if (L.lds_bytes < (size_t)7 * p.pt * 64 * 16) L.lds_bytes = (size_t)7 * p.pt * 64 * 16;  // the fold's seven accumulator sets
    if (L.lds_bytes > 150 * 1024) return MP_ERR_UNSUPPORTED;
    return MP_OK;
}

template <int KS, int S, int PT>
static int small_launch_ks(const SmallLaunch& L, hipStream_t s) {
    static AttrOnce attr_set_once;
    if (attr_set_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_small_f32_kernel<KS, S, PT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL((conv_small_f32_kernel<KS, S, PT>), dim3(L.grid), dim3(64 * kWaves), L.lds_bytes, s, L.p);
    return check_launch();
}

template <int KS, int S>
static int small_launch_pt(const SmallLaunch& L, hipStream_t s) {
    switch (L.p.pt) {
        case 1: return small_launch_ks<KS, S, 1>(L, s);
        case 3: return small_launch_ks<KS, S, 3>(L, s);
        case 4: return small_launch_ks<KS, S, 4>(L, s);
        default: return MP_ERR_UNSUPPORTED;
    }
}

int small_launch(const SmallLaunch& L, hipStream_t s) {
    if (L.p.ks == 3) return L.p.stride == 1 ? small_launch_pt<3, 1>(L, s) : small_launch_pt<3, 2>(L, s);
    return small_launch_pt<1, 1>(L, s);
}

}  // namespace mp
