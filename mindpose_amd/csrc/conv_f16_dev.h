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

inline unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ULL / d) + 1u; }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }


}  // namespace
}  // namespace mp
