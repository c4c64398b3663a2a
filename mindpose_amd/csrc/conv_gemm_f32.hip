// fp32 1x1 convolution as a blocked GEMM on the matrix cores - the compute-bound pointwise layers of the ResNet bottlenecks
// (resnet.py:74-138 conv1 / conv3 / down_sample, 128 ... 2048 channels) and of the HRNet exchange units
// (hrnet.py:258-316): Out[co][j] = sum_ci W[co][ci] X[ci][j] over the N * Ho * Wo pixel columns j of the whole batch.
//
//   workgroup   256 threads = 2 x 2 waves, a 128 (cout) x 128 (pixel column) tile of the output, all of Cin in chunks of 16;
//               columns run across images (column j -> image j / HWo, pixel j % HWo), so the small maps (8x6, 16x12) fill the
//               tile with several images
//   MFMA        v_mfma_f32_32x32x2_f32, A = weights (rows = cout), B = input (columns = pixels): a lane's accumulators are
//               16 couts of ONE pixel column and the 32 lanes of a half-wave are 32 consecutive pixels - every residual load
//               and store instruction moves two full 128-byte lines
//   operands    both chunks [16 k][128] in LDS, k-major with a pitch of 160 floats: the two k rows of an operand fetch fall
//               on disjoint bank halves (160 = 32 mod 64), one ds_read_b32 per 32x32 operand.  The weight chunk comes straight
//               from the direct kernel's packing ([Cin][Cout_pad16] for a 1x1 weight: cout contiguous = already k-major)
//   pipeline    chunk c + 1: global -> registers while chunk c runs on the matrix cores, -> LDS behind it, one barrier per
//               chunk; branch-free body (the loads past the last chunk are out of range / never consumed)
//   gather       1x1 stride 2 (the bottleneck down_sample), the four 2x2 sub-pixel phases of the transposed convolution
//               (simple_baseline_head.py:80-88) and - round 4 - the 3x3 stride-2 convolutions of the HRNet transitions and
//               exchange units (hrnet.py:280-313, 440-496) use the same tiles with the input columns fetched one by one - four
//               4-byte loads per staging unit instead of one 16-byte load - and a k loop over (cin chunk, tap) pairs; taps outside
//               the image read zeros through the buffer range check (a 64-bit mask of (tap, column) bits per thread)
//   epilogue    scale / shift per cout from LDS, (+res1)(+ReLU); the residual tile is requested in one batch behind the k loop
#include <stdlib.h>

#include "conv_gemm.h"
#include "conv_mfma.h"

namespace mp {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef MP_GEMM_ABLATE
#define MP_GEMM_ABLATE 0  // diagnostic builds (tools/variant_builds.sh): 1 = no stores, 2 = no MFMA; results wrong, timings meaningful
#endif

constexpr int kTM = 128;     // couts per workgroup
constexpr int kTN = 128;     // pixel columns per workgroup
constexpr int kKC = 16;      // input channels per chunk
constexpr int kPitch = 160;  // floats per k row in LDS
constexpr int kBuf = kKC * kPitch;

__device__ __forceinline__ void gemm_barrier() {
    // LDS traffic of this wave done, then the workgroup barrier; global loads stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// GATHER = false: 1x1 stride 1, the four columns of a staging unit are one 16-byte load.  GATHER = true: every column is fetched by
//          itself at (y * stride - pad_top + ky, x * stride - pad_left + kx) of its image, out-of-image taps read zeros (buffer range
//          check), and the k loop walks (cin chunk, tap) pairs: 1x1 stride 2 (one tap) and the 2x2 sub-pixel phases of the
//          transposed convolution (four taps, simple_baseline_head.py:80-88), whose output lands on every out_mul-th pixel
// NI = 32-column blocks per wave: 2 = 128-column tiles, 1 = 64-column tiles (launches that would otherwise leave CUs without work)
// MI = 32-cout blocks per wave likewise: 2 = 128-cout tiles, 1 = 64-cout tiles (with NI = 1 only: huge K, few outputs)
template <bool GATHER, int NI, int MI>
__global__ __launch_bounds__(256, 2) void conv1x1_f32_gemm_kernel(const GemmParams p) {
    constexpr int TN = 64 * NI, TM = 64 * MI;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* __restrict__ lds_a = smem;             // [2][16][kPitch] weights
    float* __restrict__ lds_b = smem + 2 * kBuf;  // [2][16][kPitch] input
    float* __restrict__ lds_ss = smem + 4 * kBuf; // scale[128] | shift[128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;

    int wg = blockIdx.x;
    {   // XCD-aware workgroup id (blocks b, b+8, ... share an XCD): the cout tiles of one column tile share an L2
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = wg & 7, j = wg >> 3;
        wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const int ct = wg % p.n_ct, co0 = ct * TM, col0 = (wg / p.n_ct) * TN;
    // p.phases == 4: the four sub-pixel phases of a transposed convolution in ONE launch, phase (py, px) = blockIdx.y - its own
    // packed weights (four equal slices of one buffer), padding 1 - p on the top / left, output pixels (2 y + py, 2 x + px)
    const int phase = GATHER && p.phases > 1 ? (int)blockIdx.y : 0;
    const int pad_top = GATHER && p.phases > 1 ? 1 - (phase >> 1) : p.pad_top, pad_left = GATHER && p.phases > 1 ? 1 - (phase & 1) : p.pad_left;
    const int off_y = GATHER && p.phases > 1 ? phase >> 1 : p.off_y, off_x = GATHER && p.phases > 1 ? phase & 1 : p.off_x;
    const float* __restrict__ wp = p.wp + (size_t)phase * p.wp_phase_floats;

    if (tid < TM) {
        const int co = co0 + tid;
        lds_ss[tid] = co < p.Cout ? p.scale[co] : 0.f;
        lds_ss[TM + tid] = co < p.Cout ? p.shift[co] : 0.f;
    }

    // ---- staging: a chunk of the weights = 16 rows x 32 float4 units; thread -> rows (tid >> 5) and (tid >> 5) + 8, unit tid & 31;
    // the input chunk likewise (NI = 2) or 16 rows x 16 units, one per thread (NI = 1).
    // Packed weights [Cin_pad4 / 4][T][4][Cout_pad16]: row of (ci, tap t) = ((ci >> 2) T + t) 4 + (ci & 3)
    const int srow = MI == 2 ? tid >> 5 : tid >> 4, c4 = MI == 2 ? (tid & 31) * 4 : (tid & 15) * 4;
    const int srow_b = NI == 2 ? tid >> 5 : tid >> 4, c4_b = NI == 2 ? (tid & 31) * 4 : (tid & 15) * 4;
    const int T = GATHER ? p.T : 1;
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(wp, (size_t)p.Cin_pad4 * T * p.Cout_pad16 * 4);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, (size_t)p.N * p.Cin * p.HWi * 4);
    const unsigned a_src = co0 + c4 < p.Cout_pad16 ? (unsigned)((((srow >> 2) * T) * 4 + (srow & 3)) * p.Cout_pad16 + co0 + c4) * 4u : kOob;
    const unsigned a_row8 = (unsigned)(8 * T * p.Cout_pad16) * 4u, a_chunk = (unsigned)(kKC * T * p.Cout_pad16) * 4u;
    const unsigned a_tap = (unsigned)(4 * p.Cout_pad16) * 4u;
    unsigned b_src[GATHER ? 4 : 1];  // byte offset of the unit's column(s) in channel srow_b of its image, tap 0
    unsigned long long b_inv = 0;    // GATHER: bit 4 t + e set = tap t of column e is outside the image (or the column past the end)
#pragma unroll
    for (int e = 0; e < (GATHER ? 4 : 1); ++e) {
        const int j = col0 + c4_b + e;  // HWo % 4 == 0: the four columns of a unit are one image's consecutive pixels
        const int n = j / p.HWo, pp = j - n * p.HWo;
        if constexpr (GATHER) {
            const int oy = pp / p.Wo, ox = pp - oy * p.Wo;
            const int y0 = oy * p.stride - pad_top, x0 = ox * p.stride - pad_left;
            b_src[e] = (unsigned)((n * p.Cin + srow_b) * p.HWi + y0 * p.Wi + x0) * 4u;  // may wrap below zero: masked then
            for (int t = 0, ky = 0, kx = 0; t < T; ++t) {
                const int yy = y0 + ky, xx = x0 + kx;
                if (j >= p.cols || yy < 0 || yy >= p.Hi || xx < 0 || xx >= p.Wi) b_inv |= 1ull << (4 * t + e);
                if (++kx == p.kw) { kx = 0; ++ky; }
            }
        } else {
            b_src[e] = j < p.cols ? (unsigned)((n * p.Cin + srow_b) * p.HWi + pp) * 4u : kOob;
        }
    }
    const unsigned b_row8 = (unsigned)(8 * p.HWi) * 4u, b_chunk = (unsigned)(kKC * p.HWi) * 4u;
    const int s_dst = srow * kPitch + c4, s_dst_b = srow_b * kPitch + c4_b;

    f32x4 va[MI], vb[NI];
    // the k loop walks (cin chunk, tap) pairs in order: stage_load is called for iteration 0, 1, 2, ... and steps its own counters
    // (T is 1, 4 or 9: no shift decodes it)
    int s_ch = 0, s_t = 0, s_ky = 0, s_kx = 0;
    auto stage_load = [&](int it) __attribute__((always_inline)) {  // kOob + offset stays out of range (every tensor here spans < 2 GiB)
        const int ch = GATHER ? s_ch : it, t = GATHER ? s_t : 0;
        const unsigned ao = a_src + ch * a_chunk + t * a_tap;
#pragma unroll
        for (int i = 0; i < MI; ++i) va[i] = buf_load4(rs_w, ao + i * a_row8);
        if constexpr (GATHER) {
            const unsigned bo = ch * b_chunk + (unsigned)(s_ky * p.Wi + s_kx) * 4u;
            const unsigned inv4 = (unsigned)(b_inv >> (4 * t)) & 15u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // outside the image: top bit set = out of range = zero.  The mask goes on LAST: a column left of / above the image
                // has a wrapped ("negative") base that a later addition would carry back into range
                const unsigned inv = (inv4 >> e) << 31;
#pragma unroll
                for (int i = 0; i < NI; ++i) vb[i][e] = buf_load1(rs_x, (b_src[e] + bo + i * b_row8) | inv);
            }
            // next (chunk, tap): plain selects (an if / else-if over the counters made hipcc index them in scratch memory)
            ++s_t;
            ++s_kx;
            const bool row_end = s_kx == p.kw, chunk_end = s_t == T;
            s_kx = row_end ? 0 : s_kx;
            s_ky = chunk_end ? 0 : s_ky + (row_end ? 1 : 0);
            s_t = chunk_end ? 0 : s_t;
            s_ch += chunk_end ? 1 : 0;
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) vb[i] = buf_load4(rs_x, b_src[0] + ch * b_chunk + i * b_row8);
        }
    };
    auto stage_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i) *reinterpret_cast<f32x4*>(lds_a + buf * kBuf + s_dst + i * 8 * kPitch) = va[i];
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(lds_b + buf * kBuf + s_dst_b + i * 8 * kPitch) = vb[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    stage_load(0);
    stage_store(0);
    gemm_barrier();

    const int a_off = lh * kPitch + wm * 32 * MI + l31, b_off = lh * kPitch + wn * 32 * NI + l31;
    for (int it = 0; it < p.n_iters; ++it) {
        stage_load(it + 1);
        const float* __restrict__ as = lds_a + (it & 1) * kBuf + a_off;
        const float* __restrict__ bs = lds_b + (it & 1) * kBuf + b_off;
#pragma unroll
        for (int ks = 0; ks < kKC / 2; ++ks) {
            float a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = as[ks * 2 * kPitch + mi * 32];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = bs[ks * 2 * kPitch + ni * 32];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (MP_GEMM_ABLATE & 2) acc[mi][ni][0] += a[mi] * b[ni];
                    else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                }
        }
        stage_store((it + 1) & 1);  // the other buffer: its last readers finished before the previous barrier
        gemm_barrier();
    }

    // ---- epilogue.  Accumulator r of a 32x32 tile: row (cout) 8 (r / 4) + 4 lh + r % 4, column (pixel) l31
    const size_t o_bytes = (size_t)p.N * p.Cout * p.OHW * 4;
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, o_bytes);
    unsigned o_col[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int j = col0 + wn * 32 * NI + ni * 32 + l31;
        const int n = j / p.HWo, pp = j - n * p.HWo;
        const int oy = pp / p.Wo, ox = pp - oy * p.Wo;
        o_col[ni] = j < p.cols ? (unsigned)(n * p.Cout * p.OHW + (oy * p.out_mul + off_y) * p.OW + ox * p.out_mul + off_x) * 4u : kOob;
    }
    const unsigned plane = (unsigned)p.OHW * 4u;
    // scale / shift first, then the residual tensors one after the other through ONE staging set (two sets beside the accumulators
    // would not fit two workgroups per CU), then ReLU and the stores
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = wm * 32 * MI + mi * 32 + g * 8 + lh * 4;  // four consecutive couts of the tile
            const f32x4 sc = *reinterpret_cast<const f32x4*>(lds_ss + row), sh = *reinterpret_cast<const f32x4*>(lds_ss + TM + row);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mi][ni][g * 4 + e] = acc[mi][ni][g * 4 + e] * sc[e] + sh[e];
        }
    for (int which = 0; which < 2; ++which) {
        const float* res = which == 0 ? p.res1 : p.res2;
        if (!res) continue;  // workgroup-uniform
        const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(res, o_bytes);
        f32x16 rr[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm * 32 * MI + mi * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
                    rr[mi][ni][r] = buf_load1(rs_r, (o_col[ni] + co * plane) | (co < p.Cout ? 0u : kOob));  // kOob + offset stays out of range
                }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] += rr[mi][ni][r];
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = wm * 32 * MI + mi * 32 + g * 8 + lh * 4;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = co0 + row + e;
                    float v = acc[mi][ni][g * 4 + e];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if ((MP_GEMM_ABLATE & 1) && v != 12345.678f) continue;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o,
                                                          (o_col[ni] + co * plane) | (co < p.Cout ? 0u : kOob), 0, 0);
                }
        }
}

}  // namespace

int gemm_configure(const mp_conv_desc* d, GemmLaunch& L) {
    if (!d) return MP_ERR_NULL;
    if (d->n <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return MP_ERR_SHAPE;
    if (d->flags & ~MP_CONV_SHARES_CUS) return MP_ERR_UNSUPPORTED;
    // the three forms: 1x1 stride 1 / stride 2 (no padding, dense output) and the 2x2 stride-1 sub-pixel phase of the transposed
    // convolution (padding 0 / 1 per side, output on every out_mul-th pixel)
    const bool pointwise = d->kh == 1 && d->kw == 1 && (d->stride == 1 || d->stride == 2) && d->pad_top == 0 && d->pad_left == 0 &&
                           d->out_mul == 1 && d->out_off_y == 0 && d->out_off_x == 0 && d->out_h == d->conv_h && d->out_w == d->conv_w &&
                           d->conv_h == (d->h - 1) / d->stride + 1 && d->conv_w == (d->w - 1) / d->stride + 1;
    const bool phase = d->kh == 2 && d->kw == 2 && d->stride == 1 && d->pad_top >= 0 && d->pad_top <= 1 && d->pad_left >= 0 &&
                       d->pad_left <= 1 && d->conv_h == d->h && d->conv_w == d->w && d->out_mul >= 1 && d->out_off_y >= 0 &&
                       d->out_off_x >= 0 && (d->conv_h - 1) * d->out_mul + d->out_off_y < d->out_h &&
                       (d->conv_w - 1) * d->out_mul + d->out_off_x < d->out_w;
    // round 4: the 3x3 stride-2 pad-1 convolutions of the HRNet transitions / exchange units, dense output, as a nine-tap gather
    const bool conv3s2 = d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad_top == 1 && d->pad_left == 1 && d->out_mul == 1 &&
                         d->out_off_y == 0 && d->out_off_x == 0 && d->out_h == d->conv_h && d->out_w == d->conv_w &&
                         d->conv_h == (d->h - 1) / 2 + 1 && d->conv_w == (d->w - 1) / 2 + 1;
    if ((!pointwise && !phase && !conv3s2) || d->out_rep != 1) return MP_ERR_UNSUPPORTED;
    const int hwo = d->conv_h * d->conv_w, hwi = d->h * d->w, ohw = d->out_h * d->out_w;
    L.gather = phase || d->stride == 2;
    // whole chunks of 16 input channels; float4 staging units and columns inside one image; at least most of one cout tile (3x3
    // stride 2: of one 64-channel tile - the 64-cout layers of the exchange units take 64 x 64 tiles)
    if ((d->cin % kKC) || (hwo & 3) || (!L.gather && (hwi & 3)) || d->cout < (conv3s2 ? 48 : 96)) return MP_ERR_UNSUPPORTED;
    if ((long long)d->n * d->cin * hwi * 4 >= 0x7FFFFFF0LL || (long long)d->n * d->cout * ohw * 4 >= 0x7FFFFFF0LL) return MP_ERR_UNSUPPORTED;
    GemmParams& p = L.p;
    p.N = d->n; p.Cin = d->cin; p.Cin_pad4 = (d->cin + 3) / 4 * 4; p.Cout = d->cout; p.Cout_pad16 = (d->cout + 15) / 16 * 16;
    p.HWi = hwi; p.Hi = d->h; p.Wi = d->w; p.HWo = hwo; p.Wo = d->conv_w;
    p.OHW = ohw; p.OW = d->out_w; p.out_mul = d->out_mul; p.off_y = d->out_off_y; p.off_x = d->out_off_x;
    p.stride = d->stride; p.pad_top = d->pad_top; p.pad_left = d->pad_left;
    p.T = d->kh * d->kw; p.kw = d->kw;
    p.phases = 1; p.wp_phase_floats = 0;
    L.phases = 1;
    p.cols = d->n * hwo;
    p.n_iters = d->cin / kKC * p.T;
    p.relu = d->relu;
    L.stride = d->stride;
    L.taps = p.T;
    // 128 x 128 tiles unless that leaves fewer than two workgroups per CU: then 128 x 64, then 64 x 64
    const long long t128 = (long long)((p.Cout_pad16 + kTM - 1) / kTM) * ((p.cols + kTN - 1) / kTN);
    L.ni = t128 >= 512 ? 2 : 1;
    L.mi = t128 >= 192 ? 2 : 1;
    if (p.Cout_pad16 <= 64) { L.mi = 1; L.ni = 1; }  // a 128-cout tile would be half empty
    if (const char* e = knob("MP_GEMM_NI")) {  // experiments / tests: 1, 2 = column blocks per wave; 11 = 64 x 64 tiles
        if (atoi(e) == 1 || atoi(e) == 2) { L.ni = atoi(e); L.mi = 2; }
        if (atoi(e) == 11) { L.ni = 1; L.mi = 1; }
    }
    p.n_ct = (p.Cout_pad16 + 64 * L.mi - 1) / (64 * L.mi);
    L.grid = p.n_ct * ((p.cols + 64 * L.ni - 1) / (64 * L.ni));
    L.lds_bytes = (size_t)(4 * kBuf + 2 * kTM) * 4;
    return MP_OK;
}

int gemm_configure_deconv(const mp_conv_desc* d, GemmLaunch& L) {
    // `d` = the phase (0, 0) launch of Conv2dTranspose(k=4, s=2, p=1): 2x2, stride 1, padding 1 / 1, output offset 0 / 0, every second
    // pixel of a 2h x 2w plane (layers.py Plan.deconv4x4s2); the other three phases differ in padding and offset only
    if (!d) return MP_ERR_NULL;
    if (d->kh != 2 || d->kw != 2 || d->stride != 1 || d->pad_top != 1 || d->pad_left != 1 || d->out_mul != 2 || d->out_rep != 1 ||
        d->out_off_y != 0 || d->out_off_x != 0 || d->out_h != 2 * d->h || d->out_w != 2 * d->w || d->conv_h != d->h || d->conv_w != d->w)
        return MP_ERR_UNSUPPORTED;
    int rc = gemm_configure(d, L);
    if (rc != MP_OK) return rc;
    GemmParams& p = L.p;
    p.phases = 4;
    p.wp_phase_floats = (unsigned)((size_t)p.Cin_pad4 * 4 * p.Cout_pad16);
    L.phases = 4;
    // tile shape by the workgroup count of all four phases together
    const long long t128 = 4LL * ((p.Cout_pad16 + kTM - 1) / kTM) * ((p.cols + kTN - 1) / kTN);
    L.ni = t128 >= 512 ? 2 : 1;
    L.mi = t128 >= 192 ? 2 : 1;
    if (const char* e = knob("MP_GEMM_NI")) {
        if (atoi(e) == 1 || atoi(e) == 2) { L.ni = atoi(e); L.mi = 2; }
        if (atoi(e) == 11) { L.ni = 1; L.mi = 1; }
    }
    p.n_ct = (p.Cout_pad16 + 64 * L.mi - 1) / (64 * L.mi);
    L.grid = p.n_ct * ((p.cols + 64 * L.ni - 1) / (64 * L.ni));
    return MP_OK;
}

int gemm_launch(const GemmLaunch& L, hipStream_t s) {
    if (L.mi == 1) {
        if (L.gather) hipLaunchKernelGGL((conv1x1_f32_gemm_kernel<true, 1, 1>), dim3(L.grid, L.phases), dim3(256), L.lds_bytes, s, L.p);
        else hipLaunchKernelGGL((conv1x1_f32_gemm_kernel<false, 1, 1>), dim3(L.grid, L.phases), dim3(256), L.lds_bytes, s, L.p);
    } else if (L.gather && L.ni == 2) hipLaunchKernelGGL((conv1x1_f32_gemm_kernel<true, 2, 2>), dim3(L.grid, L.phases), dim3(256), L.lds_bytes, s, L.p);
    else if (L.gather) hipLaunchKernelGGL((conv1x1_f32_gemm_kernel<true, 1, 2>), dim3(L.grid, L.phases), dim3(256), L.lds_bytes, s, L.p);
    else if (L.ni == 2) hipLaunchKernelGGL((conv1x1_f32_gemm_kernel<false, 2, 2>), dim3(L.grid, L.phases), dim3(256), L.lds_bytes, s, L.p);
    else hipLaunchKernelGGL((conv1x1_f32_gemm_kernel<false, 1, 2>), dim3(L.grid, L.phases), dim3(256), L.lds_bytes, s, L.p);
    return check_launch();
}

}  // namespace mp
