// What fits in the shadow of an MFMA issued by the SAME wave?  (gfx950, v_mfma_f32_16x16x32_f16: 4 passes = 16 cycles on the pipe)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shadow tools/probes/mfma_shadow.hip && /tmp/mfma_shadow
// Every variant runs REP x 32 MFMAs on 8 independent accumulators with K other instructions (inline asm, independent registers)
// behind each MFMA, one wave per SIMD (256 threads, one workgroup) and two waves per SIMD (512 threads); prints s_memtime cycles
// per MFMA.  16.0 = the filler is free, 16 + 4 K = nothing overlaps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b))
#define MFMAV(acc) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int KIND, int K, bool VACC, int UN = 1>
__global__ void probe(unsigned long long* out, float* sink, int rep) {
    extern __shared__ u32x4 lds[];
    u32x4 a = {threadIdx.x, 1u, 2u, 3u}, b = {5u, 6u, 7u, threadIdx.x};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, y0 = 0.5f, y1 = 0.25f;
    unsigned la = (threadIdx.x & 63) * 16;
    u32x4 l0 = {0, 0, 0, 0};
    lds[threadIdx.x] = a;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int i = 0; i < 32 * UN; ++i) {
            if (VACC) MFMAV(acc[i & 7]); else MFMA(acc[i & 7]);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) {  // v_fma_f32 chain-free: four independent registers in rotation
                    if ((k & 3) == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(y0), "v"(y1));
                    if ((k & 3) == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "v"(y0), "v"(y1));
                    if ((k & 3) == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x2) : "v"(y0), "v"(y1));
                    if ((k & 3) == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x3) : "v"(y0), "v"(y1));
                } else if (KIND == 1) {  // ds_read_b128 (never waited for inside the loop)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(l0) : "v"(la) : "memory");
                } else if (KIND == 2) {  // v_cvt_f32_f16 + v_max_f32 pairs
                    if (k & 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x1) : "v"(y0));
                    else asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x2) : "v"(y1));
                } else if (KIND == 3) {  // s_nop 0 (pure issue slot)
                    asm volatile("s_nop 0");
                } else if (KIND == 4) {  // SALU
                    asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
                } else if (KIND == 5) {  // v_accvgpr_read of an accumulator that is not being written right now
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x3) : "a"(a.x));
                } else if (KIND == 6) {  // v_pk_fma_f32
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    static_assert(sizeof(f32x2) == 8, "");
                    asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(*reinterpret_cast<f32x2*>(&acc[0]) /*unused when !VACC*/) : "v"((f32x2){y0, y1}));
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = x0 + x1 + x2 + x3 + __builtin_bit_cast(float, l0.x);
    for (int i = 0; i < 8; ++i) s += acc[i].x;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int K, bool VACC, int UN = 1>
void run(const char* name, unsigned long long* d_out, float* d_sink) {
    const int rep = 256 / UN;
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL((probe<KIND, K, VACC, UN>), dim3(1), dim3(threads), 16384, 0, d_out, d_sink, rep);
        hipLaunchKernelGGL((probe<KIND, K, VACC, UN>), dim3(1), dim3(threads), 16384, 0, d_out, d_sink, rep);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-28s K=%d acc=%s straight-line MFMAs=%d waves/SIMD=%d: %.2f cycles per MFMA (wave 0)\n", name, K, VACC ? "vgpr" : "agpr", 32 * UN,
               threads / 256, (double)h[0] / (rep * 32.0 * UN));
    }
}

int main() {
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, 64 * 8); hipMalloc(&d_sink, 64);
    run<0, 0, false>("mfma only", d_out, d_sink);
    run<0, 0, true>("mfma only", d_out, d_sink);
    run<0, 1, false>("v_fma_f32", d_out, d_sink);
    run<0, 2, false>("v_fma_f32", d_out, d_sink);
    run<0, 3, false>("v_fma_f32", d_out, d_sink);
    run<0, 4, false>("v_fma_f32", d_out, d_sink);
    run<0, 6, false>("v_fma_f32", d_out, d_sink);
    run<0, 8, false>("v_fma_f32", d_out, d_sink);
    run<0, 3, true>("v_fma_f32", d_out, d_sink);
    run<2, 2, false>("cvt / max", d_out, d_sink);
    run<2, 4, false>("cvt / max", d_out, d_sink);
    run<1, 1, false>("ds_read_b128", d_out, d_sink);
    run<1, 2, false>("ds_read_b128", d_out, d_sink);
    run<3, 2, false>("s_nop 0", d_out, d_sink);
    run<3, 6, false>("s_nop 0", d_out, d_sink);
    run<4, 2, false>("s_add_u32", d_out, d_sink);
    run<4, 6, false>("s_add_u32", d_out, d_sink);
    run<5, 2, false>("v_accvgpr_read", d_out, d_sink);
    run<5, 4, false>("v_accvgpr_read", d_out, d_sink);
    // the same streams as longer and longer straight-line code (instruction fetch: does a wave's sequential fetch keep up?)
    run<0, 0, false, 8>("mfma only", d_out, d_sink);
    run<0, 0, false, 32>("mfma only", d_out, d_sink);
    run<0, 2, false, 8>("v_fma_f32", d_out, d_sink);
    run<0, 2, false, 32>("v_fma_f32", d_out, d_sink);
    run<0, 2, false, 64>("v_fma_f32", d_out, d_sink);
    run<0, 4, false, 8>("v_fma_f32", d_out, d_sink);
    run<0, 4, false, 32>("v_fma_f32", d_out, d_sink);
    run<1, 1, false, 32>("ds_read_b128", d_out, d_sink);
    return 0;
}
