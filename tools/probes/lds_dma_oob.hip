// Probe (not product code): what does `buffer_load_dwordx4 ... lds` (LDS-DMA) write for lanes whose offset fails the buffer
// range check?  The LDS is pre-filled with a pattern; afterwards out-of-range lanes read back either zeros (the DMA wrote the
// range-check result 0) or the pattern (the DMA skipped the lane).  conv_f16_wreg relies on the former for its halo.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_oob.hip -o /tmp/lds_dma_oob && /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const void* x, u32x4* out, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) u32x4 sm[];
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = (u32x4){0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, nbytes, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int s0 = wave * 64; s0 < 1024; s0 += 256) {
        unsigned off = (s0 + lane) * 16u;
        if (((s0 + lane) & 7) == 3) off = 0x80000000u;  // every 8th slot: out of range
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sm + s0), 16, off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = sm[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 0x10000u + i;
    unsigned *dx, *dout;
    hipMalloc(&dx, 16384); hipMalloc(&dout, 16384);
    hipMemcpy(dx, h.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 16384, 0, dx, reinterpret_cast<u32x4*>(dout), 16384);
    std::vector<unsigned> o(4096);
    hipMemcpy(o.data(), dout, 16384, hipMemcpyDeviceToHost);
    int ok_in = 0, zero_oob = 0, pattern_oob = 0, other = 0;
    for (int s = 0; s < 1024; ++s) {
        const bool oob = (s & 7) == 3;
        bool same = true, zero = true, pat = true;
        for (int j = 0; j < 4; ++j) { same &= o[4*s+j] == h[4*s+j]; zero &= o[4*s+j] == 0; pat &= o[4*s+j] == 0xDEADBEEFu; }
        if (!oob) ok_in += same; else if (zero) ++zero_oob; else if (pat) ++pattern_oob; else ++other;
    }
    printf("in-range slots copied: %d/896; out-of-range slots: zero %d, untouched %d, other %d (of 128)\n", ok_in, zero_oob, pattern_oob, other);
    printf("LDS_DMA_OOB_WRITES_ZERO=%d\n", zero_oob == 128 && ok_in == 896);
    return 0;
}
