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
#include "common.h"

namespace mp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
    int CK;            // cin per chunk (multiple of 4)
    int n_chunks;
    int n_ct;          // cout tiles
    int tiles_y;       // row bands per image (1 when G > 1)
    int tiles_n;       // image groups
    int ncols;         // input columns copied per row
    int lds_w_off;     // float offset of the weight tile inside dynamic LDS
    int out_h, out_w, out_mul, out_rep, off_y, off_x;
    int relu;
    unsigned magic_ncols, magic_perc, magic_rwo, magic_wo;  // fast-division multipliers
    int RWo;           // R * Wo
    int total_blocks;
};

__device__ __forceinline__ unsigned fastdiv(unsigned e, unsigned d, unsigned magic) {
    // magic = floor(2^32 / d) + 1; exact while e * d < 2^32 (all uses here: e, d < 2^16)
    return d == 1 ? e : __umulhi(e, magic);
}

template <int KS, int S, int PS, int CS, int WAVES_P, int WAVES_C>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvKParams p) {
    static_assert(WAVES_P * WAVES_C == 4, "4 waves per workgroup");
    constexpr int T = KS * KS;
    constexpr int CT = 16 * CS * WAVES_C;
    constexpr bool SWZ = (CT % 32) == 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ lds_in = smem;
    float* __restrict__ lds_w = smem + p.lds_w_off;

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

    // zero the input tile once: halo positions are never overwritten by the chunk copies
    {
        const int n4 = (p.CK * p.cin_plane) >> 2;
        float4* z = reinterpret_cast<float4*>(lds_in);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid; i < n4; i += 256) z[i] = zero;
    }

    // lane-constant LDS offsets
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

    const int perc = p.Rin * p.ncols;  // staged elements per input channel (rows outside the image are skipped)

    for (int ch = 0; ch < p.n_chunks; ++ch) {
        const int c0 = ch * p.CK;
        const int ckv = min(p.CK, p.Cin - c0);           // real input channels in this chunk
        const int nq = min(p.CK, p.Cin_pad4 - c0) >> 2;  // cin quads (weights are zero-padded)
        __syncthreads();  // previous chunk fully consumed (also orders the zero fill)
        // ---- stage the input chunk: coalesced global rows -> LDS rows with halo offset
        for (int g = 0; g < p.G; ++g) {
            if (n0 + g >= p.N) break;
            const float* __restrict__ src = p.x + ((size_t)(n0 + g) * p.Cin + c0) * HW;
            float* __restrict__ dst = lds_in + g * p.img_plane + p.pad_l;
            const int total = ckv * perc;
#pragma unroll 4
            for (int e = tid; e < total; e += 256) {
                const unsigned c = fastdiv(e, perc, p.magic_perc);
                const unsigned rem = e - c * perc;
                const unsigned r = fastdiv(rem, p.ncols, p.magic_ncols);
                const unsigned xx = rem - r * p.ncols;
                const int yin = y_in0 + (int)r;
                if (yin >= 0 && yin < p.H) dst[c * p.cin_plane + r * p.Wp + xx] = src[(size_t)c * HW + yin * p.W + xx];
            }
        }
        // ---- stage the weight chunk: rows of CT floats out of [quad][tap][kq][Cout_pad16]
        {
            constexpr int C4 = CT / 4;
            const int rows = nq * T * 4;
            const float* __restrict__ wsrc = p.wp + (size_t)(c0 >> 2) * T * 4 * p.Cout_pad16 + ct * CT;
            const int total4 = rows * C4;
#pragma unroll 2
            for (int i = tid; i < total4; i += 256) {
                const int row = i / C4, c4 = (i - row * C4) << 2;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ct * CT + c4 < p.Cout_pad16)
                    v = *reinterpret_cast<const float4*>(wsrc + (size_t)row * p.Cout_pad16 + c4);
                int dc = c4;
                if (SWZ) dc ^= (row & 1) << 4;
                *reinterpret_cast<float4*>(lds_w + row * CT + dc) = v;
            }
        }
        __syncthreads();
        // ---- MFMA over this chunk
        for (int q = 0; q < nq; ++q) {
            const int in_q = q * 4 * p.cin_plane;
            const int w_q = q * T * 4 * CT;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int dy = t / KS, dx = t % KS;
                const int in_off = in_q + dy * p.Wp + dx;
                const int w_off = w_q + t * 4 * CT;
                float av[PS], bv[CS];
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) av[ps] = lds_in[a_off[ps] + in_off];
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) bv[cs] = lds_w[b_off[cs] + w_off];
#pragma unroll
                for (int ps = 0; ps < PS; ++ps)
#pragma unroll
                    for (int cs = 0; cs < CS; ++cs)
                        acc[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ps], bv[cs], acc[ps][cs], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: BN scale/shift (+res1) (+res2) (+ReLU) -> NCHW, with the output mapping
    const int plane_o = p.out_h * p.out_w;
    const bool vec_ok = ((p.Wo & 3) == 0);
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
                    // nearest up-sample by s: 4 pixels -> 4s contiguous outputs per row, s rows
                    const int s = p.out_rep;
                    for (int a = 0; a < s; ++a) {
                        const size_t o = base + (size_t)(yy * s + a + p.off_y) * p.out_w + (size_t)xx * s + p.off_x;
                        for (int j = 0; j < s; ++j) {
                            float4 r4;
                            if (s == 2) r4 = (j == 0) ? make_float4(v[0], v[0], v[1], v[1]) : make_float4(v[2], v[2], v[3], v[3]);
                            else if (s == 4) { const float e = (j == 0) ? v[0] : (j == 1) ? v[1] : (j == 2) ? v[2] : v[3]; r4 = make_float4(e, e, e, e); }
                            else { const int jj = j >> 1; const float e = (jj == 0) ? v[0] : (jj == 1) ? v[1] : (jj == 2) ? v[2] : v[3]; r4 = make_float4(e, e, e, e); }
                            const size_t oo = o + 4 * j;
                            if (p.res1) { const float4 t4 = *reinterpret_cast<const float4*>(p.res1 + oo); r4.x += t4.x; r4.y += t4.y; r4.z += t4.z; r4.w += t4.w; }
                            if (p.res2) { const float4 t4 = *reinterpret_cast<const float4*>(p.res2 + oo); r4.x += t4.x; r4.y += t4.y; r4.z += t4.z; r4.w += t4.w; }
                            if (p.relu) { r4.x = fmaxf(r4.x, 0.f); r4.y = fmaxf(r4.y, 0.f); r4.z = fmaxf(r4.z, 0.f); r4.w = fmaxf(r4.w, 0.f); }
                            *reinterpret_cast<float4*>(p.out + oo) = r4;
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

// tile variants: index -> (PS, CS, WAVES_P, WAVES_C); CT = 16*CS*WAVES_C, PT = 16*PS*WAVES_P
enum ConvVariant { V_CT32_PT192 = 0, V_CT64_PT192 = 1, V_CT48_PT192 = 2, V_CT64_PT96 = 3, V_CT32_PT96 = 4, V_COUNT = 5 };

inline void variant_dims(int v, int& ct, int& pt) {
    static const int cts[V_COUNT] = {32, 64, 48, 64, 32};
    static const int pts[V_COUNT] = {192, 192, 192, 96, 96};
    ct = cts[v];
    pt = pts[v];
}

template <int KS, int S, int PS, int CS, int WAVES_P, int WAVES_C>
int launch_variant(const ConvKParams& p, size_t lds_bytes, hipStream_t s) {
    auto kern = conv_mfma_kernel<KS, S, PS, CS, WAVES_P, WAVES_C>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.total_blocks), dim3(256), lds_bytes, s, p);
    return check_launch();
}

template <int KS, int S>
int launch_ks(const ConvKParams& p, int variant, size_t lds_bytes, hipStream_t s) {
    switch (variant) {
        case V_CT32_PT192: return launch_variant<KS, S, 3, 2, 4, 1>(p, lds_bytes, s);
        case V_CT64_PT192: return launch_variant<KS, S, 3, 4, 4, 1>(p, lds_bytes, s);
        case V_CT48_PT192: return launch_variant<KS, S, 3, 3, 4, 1>(p, lds_bytes, s);
        case V_CT64_PT96: return launch_variant<KS, S, 3, 2, 2, 2>(p, lds_bytes, s);
        case V_CT32_PT96: return launch_variant<KS, S, 3, 1, 2, 2>(p, lds_bytes, s);
        default: return MP_ERR_UNSUPPORTED;
    }
}

// one translation unit per kernel size (parallel compilation)
int launch_conv_k1(const ConvKParams& p, int stride, int variant, size_t lds_bytes, hipStream_t s);
int launch_conv_k2(const ConvKParams& p, int stride, int variant, size_t lds_bytes, hipStream_t s);
int launch_conv_k3(const ConvKParams& p, int stride, int variant, size_t lds_bytes, hipStream_t s);
int launch_conv_k7(const ConvKParams& p, int stride, int variant, size_t lds_bytes, hipStream_t s);

}  // namespace mp
