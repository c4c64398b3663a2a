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
//   stride 2    (the bottleneck down_sample / HRNet-free ResNet shortcut): the input columns are gathered (pixel (2 oy, 2 ox)
//               of the input plane) with four 4-byte loads per staging unit instead of one 16-byte load
//   epilogue    scale / shift per cout from LDS, (+res1)(+ReLU); the residual tile is requested in one batch behind the k loop
#include <stdlib.h>

#include "conv_gemm.h"
#include "conv_mfma.h"

namespace mp {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// NI = 32-column blocks per wave: 2 = 128-column tiles, 1 = 64-column tiles (launches that would otherwise leave CUs without work)
template <bool S2, int NI>
__global__ __launch_bounds__(256, 2) void conv1x1_f32_gemm_kernel(const GemmParams p) {
    constexpr int TN = 64 * NI;
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
    const int ct = wg % p.n_ct, co0 = ct * kTM, col0 = (wg / p.n_ct) * TN;

    if (tid < kTM) {
        const int co = co0 + tid;
        lds_ss[tid] = co < p.Cout ? p.scale[co] : 0.f;
        lds_ss[kTM + tid] = co < p.Cout ? p.shift[co] : 0.f;
    }

    // ---- staging: a chunk of the weights = 16 rows x 32 float4 units; thread -> rows (tid >> 5) and (tid >> 5) + 8, unit tid & 31;
    // the input chunk likewise (NI = 2) or 16 rows x 16 units, one per thread (NI = 1)
    const int srow = tid >> 5, c4 = (tid & 31) * 4;
    const int srow_b = NI == 2 ? srow : tid >> 4, c4_b = NI == 2 ? c4 : (tid & 15) * 4;
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.wp, (size_t)p.Cin_pad4 * p.Cout_pad16 * 4);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, (size_t)p.N * p.Cin * p.HWi * 4);
    const unsigned a_src = co0 + c4 < p.Cout_pad16 ? (unsigned)(srow * p.Cout_pad16 + co0 + c4) * 4u : kOob;
    const unsigned a_row8 = (unsigned)(8 * p.Cout_pad16) * 4u, a_chunk = (unsigned)(kKC * p.Cout_pad16) * 4u;
    unsigned b_src[S2 ? 4 : 1];  // byte offset of the unit's column(s) in channel 0 of its image (kOob: past the last column)
#pragma unroll
    for (int e = 0; e < (S2 ? 4 : 1); ++e) {
        const int j = col0 + c4_b + e;  // HWo % 4 == 0: the four columns of a unit are one image's consecutive pixels
        const int n = j / p.HWo, pp = j - n * p.HWo;
        int pix = pp;
        if constexpr (S2) {
            const int oy = pp / p.Wo, ox = pp - oy * p.Wo;
            pix = 2 * oy * p.Wi + 2 * ox;
        }
        b_src[e] = j < p.cols ? (unsigned)((n * p.Cin + srow_b) * p.HWi + pix) * 4u : kOob;
    }
    const unsigned b_row8 = (unsigned)(8 * p.HWi) * 4u, b_chunk = (unsigned)(kKC * p.HWi) * 4u;
    const int s_dst = srow * kPitch + c4, s_dst_b = srow_b * kPitch + c4_b;

    f32x4 va[2], vb[NI];
    auto stage_load = [&](int ch) {  // kOob + offset stays out of range (every tensor here spans < 2 GiB)
#pragma unroll
        for (int i = 0; i < 2; ++i) va[i] = buf_load4(rs_w, a_src + ch * a_chunk + i * a_row8);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if constexpr (S2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) vb[i][e] = buf_load1(rs_x, b_src[e] + ch * b_chunk + i * b_row8);
            } else {
                vb[i] = buf_load4(rs_x, b_src[0] + ch * b_chunk + i * b_row8);
            }
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(lds_a + buf * kBuf + s_dst + i * 8 * kPitch) = va[i];
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(lds_b + buf * kBuf + s_dst_b + i * 8 * kPitch) = vb[i];
    };

    f32x16 acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    stage_load(0);
    stage_store(0);
    gemm_barrier();

    const int a_off = lh * kPitch + wm * 64 + l31, b_off = lh * kPitch + wn * 32 * NI + l31;
    for (int ch = 0; ch < p.n_chunks; ++ch) {
        stage_load(ch + 1);
        const float* __restrict__ as = lds_a + (ch & 1) * kBuf + a_off;
        const float* __restrict__ bs = lds_b + (ch & 1) * kBuf + b_off;
#pragma unroll
        for (int ks = 0; ks < kKC / 2; ++ks) {
            float a[2], b[NI];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = as[ks * 2 * kPitch + mi * 32];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = bs[ks * 2 * kPitch + ni * 32];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        stage_store((ch + 1) & 1);  // the other buffer: its last readers finished before the previous barrier
        gemm_barrier();
    }

    // ---- epilogue.  Accumulator r of a 32x32 tile: row (cout) 8 (r / 4) + 4 lh + r % 4, column (pixel) l31
    const size_t o_bytes = (size_t)p.N * p.Cout * p.HWo * 4;
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, o_bytes);
    const __amdgpu_buffer_rsrc_t rs_r1 = make_rsrc(p.res1 ? p.res1 : p.out, p.res1 ? o_bytes : 0);
    unsigned o_col[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int j = col0 + wn * 32 * NI + ni * 32 + l31;
        const int n = j / p.HWo, pp = j - n * p.HWo;
        o_col[ni] = j < p.cols ? (unsigned)(n * p.Cout * p.HWo + pp) * 4u : kOob;
    }
    const unsigned plane = (unsigned)p.HWo * 4u;
    f32x16 r1[2][NI];
    if (p.res1) {  // workgroup-uniform
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm * 64 + mi * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
                    r1[mi][ni][r] = buf_load1(rs_r1, (o_col[ni] + co * plane) | (co < p.Cout ? 0u : kOob));  // kOob + offset stays out of range
                }
    } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) r1[mi][ni][r] = 0.f;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = wm * 64 + mi * 32 + g * 8 + lh * 4;  // four consecutive couts of the tile
            const f32x4 sc = *reinterpret_cast<const f32x4*>(lds_ss + row), sh = *reinterpret_cast<const f32x4*>(lds_ss + kTM + row);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = co0 + row + e;
                    float v = acc[mi][ni][g * 4 + e] * sc[e] + sh[e] + r1[mi][ni][g * 4 + e];
                    if (p.relu) v = fmaxf(v, 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o,
                                                          (o_col[ni] + co * plane) | (co < p.Cout ? 0u : kOob), 0, 0);
                }
        }
}

}  // namespace

int gemm_configure(const mp_conv_desc* d, GemmLaunch& L) {
    if (!d) return MP_ERR_NULL;
    if (d->n <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0) return MP_ERR_SHAPE;
    if (d->kh != 1 || d->kw != 1 || (d->stride != 1 && d->stride != 2) || d->pad_top != 0 || d->pad_left != 0) return MP_ERR_UNSUPPORTED;
    if (d->conv_h != (d->h - 1) / d->stride + 1 || d->conv_w != (d->w - 1) / d->stride + 1) return MP_ERR_UNSUPPORTED;
    if (d->out_h != d->conv_h || d->out_w != d->conv_w) return MP_ERR_UNSUPPORTED;
    if (d->out_mul != 1 || d->out_rep != 1 || d->out_off_y != 0 || d->out_off_x != 0) return MP_ERR_UNSUPPORTED;
    if (d->flags & ~MP_CONV_SHARES_CUS) return MP_ERR_UNSUPPORTED;
    const int hwo = d->conv_h * d->conv_w, hwi = d->h * d->w;
    // whole chunks of 16 input channels; float4 staging units and columns inside one image; at least most of one cout tile
    if ((d->cin % kKC) || (hwo & 3) || (d->stride == 1 && (hwi & 3)) || d->cout < 96) return MP_ERR_UNSUPPORTED;
    if ((long long)d->n * d->cin * hwi * 4 >= 0x7FFFFFF0LL || (long long)d->n * d->cout * hwo * 4 >= 0x7FFFFFF0LL) return MP_ERR_UNSUPPORTED;
    GemmParams& p = L.p;
    p.N = d->n; p.Cin = d->cin; p.Cin_pad4 = (d->cin + 3) / 4 * 4; p.Cout = d->cout; p.Cout_pad16 = (d->cout + 15) / 16 * 16;
    p.HWi = hwi; p.Wi = d->w; p.HWo = hwo; p.Wo = d->conv_w;
    p.cols = d->n * hwo;
    p.n_ct = (p.Cout_pad16 + kTM - 1) / kTM;
    p.n_chunks = d->cin / kKC;
    p.relu = d->relu;
    p.magic_hwo = 0; p.magic_wo = 0;
    L.stride = d->stride;
    // 128-column tiles unless that leaves fewer than two workgroups per CU
    L.ni = (long long)p.n_ct * ((p.cols + kTN - 1) / kTN) >= 512 ? 2 : 1;
    if (const char* e = getenv("MP_GEMM_NI")) {  // experiments
        if (atoi(e) == 1 || atoi(e) == 2) L.ni = atoi(e);
    }
    L.grid = p.n_ct * ((p.cols + 64 * L.ni - 1) / (64 * L.ni));
    L.lds_bytes = (size_t)(4 * kBuf + 2 * kTM) * 4;
    return MP_OK;
}

int gemm_launch(const GemmLaunch& L, hipStream_t s) {
    if (L.stride == 2 && L.ni == 2) hipLaunchKernelGGL((conv1x1_f32_gemm_kernel<true, 2>), dim3(L.grid), dim3(256), L.lds_bytes, s, L.p);
    else if (L.stride == 2) hipLaunchKernelGGL((conv1x1_f32_gemm_kernel<true, 1>), dim3(L.grid), dim3(256), L.lds_bytes, s, L.p);
    else if (L.ni == 2) hipLaunchKernelGGL((conv1x1_f32_gemm_kernel<false, 2>), dim3(L.grid), dim3(256), L.lds_bytes, s, L.p);
    else hipLaunchKernelGGL((conv1x1_f32_gemm_kernel<false, 1>), dim3(L.grid), dim3(256), L.lds_bytes, s, L.p);
    return check_launch();
}

}  // namespace mp
