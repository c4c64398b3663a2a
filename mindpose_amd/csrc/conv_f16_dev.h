// Device-side helpers shared by the fp16 convolution kernels (conv_f16.hip, conv_f16_mt.hip).
#pragma once
#include "common.h"

namespace mp {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
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
__device__ __forceinline__ u32x2 f16_pack4(f32x4 v) {
    const f16x4 o = (f16x4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    return __builtin_bit_cast(u32x2, o);
}

inline unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ULL / d) + 1u; }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }


}  // namespace
}  // namespace mp
