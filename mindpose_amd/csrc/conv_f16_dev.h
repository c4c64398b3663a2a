// Device-side helpers shared by the fp16 convolution kernels (conv_f16.hip, conv_f16_mt.hip).
#pragma once
#include "common.h"

namespace mp {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kOob = 0x80000000u;
// "masked" marker for the two ADDENDS of an output offset (cout part + pixel part): marker + marker, marker + any valid part
// and a valid sum are all told apart by the buffer range check alone as long as the tensor is smaller than the marker, so the
// epilogue adds the parts without a select per store (tensors >= 1.5 GB are refused at plan time)
constexpr unsigned kInv = 0x60000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes < 0x7FFFFFF0u ? bytes : 0x7FFFFFF0u), 0x00020000);
}
__device__ __forceinline__ unsigned fastdiv(unsigned e, unsigned d, unsigned magic) { return d == 1 ? e : __umulhi(e, magic); }

// Cout-tile PAIRING: the 16x16 accumulator of v_mfma_f32_16x16x32_f16 gives a lane 4 consecutive MFMA rows of one pixel.  If rows
// map to couts in order, that is 4 couts = HALF a 16-byte channel block -> 8-byte stores (0.54-0.70 of the 16-byte rate, twice the
// store instructions).  Tiles 2j and 2j+1 of a wave therefore carry the couts of their 32-cout group INTERLEAVED in fours: row r
// of the even tile = cout 8*(r/4) + r%4, of the odd tile the same + 4.  The lane (lq, pixel) then holds couts 8*lq .. 8*lq+3 in
// the even tile and 8*lq+4 .. +7 in the odd one: one whole channel block, one 16-byte store (and one 16-byte residual load).
// Only WHICH weight rows a lane fetches changes (the packed weight layout does not); values are bit-identical.  An odd last tile
// keeps the plain order and 8-byte stores.
template <int CS>
__device__ __forceinline__ int f16_a_row(int cs, int lr) {  // cout (within the wave's CS*16) whose weights row lr of tile cs carries
    return cs < (CS & ~1) ? (cs & ~1) * 16 + 8 * (lr >> 2) + (lr & 3) + 4 * (cs & 1) : cs * 16 + lr;
}
template <int CS>
__device__ __forceinline__ int f16_d_cout(int cs, int lq) {  // first of the 4 couts (within the wave's CS*16) in the lane's accumulator
    return cs < (CS & ~1) ? (cs & ~1) * 16 + 8 * lq + 4 * (cs & 1) : cs * 16 + 4 * lq;
}

// epilogue of one pixel of one cout tile: scale/shift, residual halves, ReLU
__device__ __forceinline__ f32x4 f16_epi4(f32x4 acc, f32x4 sc, f32x4 sh, bool has1, u32x2 r1, bool has2, u32x2 r2, int relu) {
    f32x4 v = acc * sc + sh;
    if (has1) {
        const f16x4 h = __builtin_bit_cast(f16x4, r1);
        v += (f32x4){(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    }
    if (has2) {
        const f16x4 h = __builtin_bit_cast(f16x4, r2);
        v += (f32x4){(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
}
__device__ __forceinline__ f32x4 f16_relu4(f32x4 v, int relu) {  // the ReLU of f16_epi4 alone (kernels that cut the epilogue in two)
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
}
__device__ __forceinline__ u32x2 f16_pack4(f32x4 v) {
    const f16x4 o = (f16x4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    return __builtin_bit_cast(u32x2, o);
}

// ---- BatchNorm statistics from the conv epilogue (training, amp O2) ------------------------------------------------------------------
// The conv that feeds a BatchNorm also produces that BatchNorm's per-channel sums, so the separate reduction pass over the tensor
// disappears (hrnet.py:51-64 conv -> bn; tools/train.py:170-181 amp O2).  Two modes (ConvF16Params::st_mode):
//   1  forward:  the launch computes z;  partial sums of  o  and  o*o  with o = the fp16-ROUNDED output (what the BatchNorm reads)
//   2  backward: the launch computes the gradient dy that reaches a BatchNorm's output (a data-gradient conv, residual gradient
//                already added in its epilogue).  With the BatchNorm's input z and (ReLU layers) its output y read at the same
//                offsets:  g = dy * [y > 0];  partial sums of  g  and  g * z;  the tensor stored is g (pre-masked), so the
//                BatchNorm's backward pass needs neither the mask nor the reduction.
// A lane of the 16x16 accumulator holds 4 consecutive couts of one pixel: the per-lane sums run over the lane's pixels (PS tiles,
// every tile of a persistent workgroup), then over the 16 pixel lanes of a row (DPP, fixed order), then over the waves that split
// the pixels (LDS, fixed order).  One fp32 partial per (workgroup, channel): [C8out][n_part][8 channels][2] - a channel block's
// partials are contiguous for the consumer.  No atomics, no cross-workgroup traffic: bit-reproducible.
__device__ __forceinline__ float row16_sum(float v) {
    // sum over the 16 lanes of a DPP row (= the 16 pixel columns of one accumulator row group); every lane ends with the total
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
    return v;
}

// accumulate one stored value group (4 couts of one pixel): `o` = the packed fp16 output of f16_pack4.
//   MODE 1: sums of o and o * o.
//   MODE 2: zq / yq = the 8-byte groups of the BatchNorm's z / y tensors at the same position; g = o * [y > 0] (relu) or o;
//           sums of g and g * z (RAW z: the consumer turns them into sum g * xhat = invstd * (sum g z - mean * sum g) in fp64, so
//           the conv kernel needs no per-channel constants); `o` is rewritten to g.
template <int MODE>
__device__ __forceinline__ void f16_stats_acc(u32x2& o, bool valid, f32x4& sa, f32x4& sb, u32x2 zq, u32x2 yq, int relu) {
    const f16x4 h = __builtin_bit_cast(f16x4, o);
    const float v[4] = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = valid ? v[i] : 0.f;
            sa[i] += x;
            sb[i] = __builtin_fmaf(x, x, sb[i]);
        }
    } else {
        const f16x4 zh = __builtin_bit_cast(f16x4, zq), yh = __builtin_bit_cast(f16x4, yq);
        const float z[4] = {(float)zh.x, (float)zh.y, (float)zh.z, (float)zh.w};
        const float y[4] = {(float)yh.x, (float)yh.y, (float)yh.z, (float)yh.w};
        f16x4 g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool open = valid && (relu == 0 || y[i] > 0.f);
            const float x = open ? v[i] : 0.f;
            sa[i] += x;
            sb[i] = __builtin_fmaf(x, open ? z[i] : 0.f, sb[i]);
            g[i] = (_Float16)x;  // x is v[i] (already an fp16 value) or 0: exact
        }
        o = __builtin_bit_cast(u32x2, g);
    }
}

// End of the workgroup: row sums by DPP, pixel-splitting waves combined through LDS in wave order, one 32-byte store per lane group.
// `scratch`: >= 4 * CS * 16 * 2 floats of LDS that no wave reads any more once the leading barrier has passed.
template <int CS, int WAVES_P, int WAVES_C>
__device__ __forceinline__ void f16_stats_flush(f32x4 (&sa)[CS], f32x4 (&sb)[CS], float* scratch, float* __restrict__ part, int n_part,
                                                int part_idx, int co_wg, int c8out, int wp_i, int wc_i, int lq, int lr) {
#pragma unroll
    for (int cs = 0; cs < CS; ++cs)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sa[cs][i] = row16_sum(sa[cs][i]);
            sb[cs][i] = row16_sum(sb[cs][i]);
        }
    constexpr int CW = CS * 16;  // couts per wave
    if constexpr (WAVES_P == 1) {
        if (lr == 0) {
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                const int co = co_wg + wc_i * CW + f16_d_cout<CS>(cs, lq);
                if (co < c8out * 8) {
                    float* dst = part + (((size_t)(co >> 3) * n_part + part_idx) * 8 + (co & 7)) * 2;
                    *reinterpret_cast<f32x4*>(dst) = (f32x4){sa[cs][0], sb[cs][0], sa[cs][1], sb[cs][1]};
                    *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){sa[cs][2], sb[cs][2], sa[cs][3], sb[cs][3]};
                }
            }
        }
    } else {
        __syncthreads();  // every wave is past its last LDS operand read: the tile memory may be reused
        if (lr == 0) {
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                float* dst = scratch + ((wp_i * WAVES_C + wc_i) * CW + f16_d_cout<CS>(cs, lq)) * 2;
                *reinterpret_cast<f32x4*>(dst) = (f32x4){sa[cs][0], sb[cs][0], sa[cs][1], sb[cs][1]};
                *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){sa[cs][2], sb[cs][2], sa[cs][3], sb[cs][3]};
            }
        }
        __syncthreads();
        const int t = threadIdx.x;  // one thread per (cout of the workgroup, sum | sum of squares)
        if (t < WAVES_C * CW * 2) {
            const int c = t >> 1, wc = c / CW, cw = c - wc * CW;
            float v = 0.f;
#pragma unroll
            for (int wp = 0; wp < WAVES_P; ++wp) v += scratch[((wp * WAVES_C + wc) * CW + cw) * 2 + (t & 1)];
            const int co = co_wg + c;
            if (co < c8out * 8) part[(((size_t)(co >> 3) * n_part + part_idx) * 8 + (co & 7)) * 2 + (t & 1)] = v;
        }
    }
}

inline unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ULL / d) + 1u; }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }


}  // namespace
}  // namespace mp
