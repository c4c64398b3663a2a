// What does it cost a wave that is alone on its SIMD to stage 1 KB pieces (64 lanes x 16 B) of a streamed tensor into LDS while it
// keeps the matrix pipe busy?  (gfx950; the question behind conv_f16_ws.hip's ring: LDS-DMA or registers?)
//   hipcc --offload-arch=gfx950 -O3 -o build/probes/stage_cost tools/probes/stage_cost.hip && build/probes/stage_cost
// 256 workgroups x 256 threads, one workgroup per CU (100 KB of LDS), every wave runs ITERS x (64 MFMAs + P pieces), each piece a
// different KB of a 1 GB buffer (HBM traffic like the real kernel's: P = 1 ~ 2.4 TB/s over the chip, 2 ~ 4.9, 4 ~ 9.8 = beyond HBM).
//   MODE 0  MFMAs only                      MODE 1  buffer_load_dwordx4 ... lds (LDS-DMA), P per iteration, evenly spaced
//   MODE 2  buffer_load_dwordx4 -> VGPRs, written to LDS with ds_write_b128 one iteration later (in-order vmcnt wait per piece)
//   MODE 3  as 2 plus 12 VALU instructions per piece between the wait and the write (8 v_fma_mix + 4 v_pk_max: y = relu(z * s + b))
// Prints s_memtime cycles per iteration (1024 = the MFMAs alone) and the achieved read rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > 0x7FFFFFFFull ? 0x7FFFFFFF : bytes), 0x00020000);
}

template <int MODE, int P, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void probe(const char* src, unsigned long long* out, float* sink, int iters, int skew) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 a = {threadIdx.x, 1u, 2u, 3u}, b = {5u, 6u, 7u, threadIdx.x};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const size_t wave_id = (size_t)blockIdx.x * (THREADS / 64) + wave, n_waves = (size_t)gridDim.x * (THREADS / 64);
    u32x4 r[P > 0 ? P : 1];
    for (int j = 0; j < P; ++j) r[j] = (u32x4){0u, 0u, 0u, 0u};
    float sc = 1.5f, sh = 0.25f;
    u32x4* mine = lds + wave * 512;  // 8 KB of ring per wave
    unsigned long long t0, t1;
    __syncthreads();
    for (int i = 0; i < (wave & 3) * skew; ++i) asm volatile("s_nop 15" ::: "memory");  // skew the CU's four SIMDs against each other (16 cycles a step)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        // every wave of the chip walks its own KB-sized pieces: piece (it, j) of wave w at ((it * P + j) * n_waves + w) KB
        const char* base = src + ((size_t)it * P * n_waves + wave_id) * 1024;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            MFMA(acc[i & 7]);
            if constexpr (P > 0) {
                if (i % (64 / P) == 0) {
                    const int j = i / (64 / P);
                    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base + (size_t)j * n_waves * 1024, 1024);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (MODE == 1) {
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(mine + ((it & 1) * P + j) * 64), 16,
                                                                 lane * 16, 0, 0, 0);
                    } else if constexpr (MODE >= 2) {
                        // the oldest of the P loads in flight is this slot's piece of the previous iteration
                        if (it > 0) {
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P - 1) : "memory");
                            u32x4 v = r[j];
                            if constexpr (MODE == 3) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    unsigned lo, hi = 0;
                                    asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(v[q]), "v"(sc), "v"(sh));
                                    asm volatile("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(v[q]), "v"(sc), "v"(sh));
                                    asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(v[q]) : "v"(lo), "v"(hi));
                                }
                            }
                            mine[((it & 1) * P + j) * 64 + lane] = v;
                        }
                        r[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, 0));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int j = 0; j < P; ++j) s += (float)r[j].x;
    s += (float)mine[lane].x;
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (lane == 0) out[wave_id] = t1 - t0;
}

static double g_ticks_mfma_only = 0;

template <int MODE, int P, int THREADS = 256>
static void run(const char* name, const char* src, unsigned long long* d_out, float* d_sink, int iters_total, int skew = 0) {
    const int iters = iters_total * 256 / THREADS;  // the same work per SIMD
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, P, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, P, THREADS>), dim3(256), dim3(THREADS), 100 * 1024, 0, src, d_out, d_sink, iters, skew);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, P, THREADS>), dim3(256), dim3(THREADS), 100 * 1024, 0, src, d_out, d_sink, iters, skew);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int nw = 256 * THREADS / 64;
    std::vector<unsigned long long> h(nw);
    (void)hipMemcpy(h.data(), d_out, nw * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    // wave life per iteration OF THE SIMD (two waves: each runs half the iterations side by side), in units of the MFMA-only
    // one-wave run = 1024 cycles per iteration
    const double per_iter = sum / nw / iters_total;
    if (MODE == 0 && THREADS == 256) g_ticks_mfma_only = per_iter;
    const double cyc = g_ticks_mfma_only > 0 ? per_iter / g_ticks_mfma_only * 1024.0 : 0.0;
    const double gb = (double)P * 1024.0 * nw * iters / 1e9;
    printf("%-44s %d wave/SIMD P=%d  %7.1f cycles / iteration (+%6.1f, %5.1f per piece)  %6.2f TB/s read  %.3f ms\n", name, THREADS / 256, P,
           cyc, cyc - 1024.0, P ? (cyc - 1024.0) / P : 0.0, gb / ms, ms);
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    char* src; unsigned long long* d_out; float* d_sink;
    (void)hipMalloc(&src, bytes); (void)hipMemset(src, 1, bytes);
    (void)hipMalloc(&d_out, 2048 * 8); (void)hipMalloc(&d_sink, 4096);
    const int iters = 200;  // 200 x 4 pieces x 1024 waves x 1 KB = 0.8 GB at P = 4
    run<0, 0>("MFMA only", src, d_out, d_sink, iters);
    run<1, 1>("LDS-DMA", src, d_out, d_sink, iters);
    run<1, 2>("LDS-DMA", src, d_out, d_sink, iters);
    run<1, 4>("LDS-DMA", src, d_out, d_sink, iters);
    run<2, 1>("registers + ds_write_b128", src, d_out, d_sink, iters);
    run<2, 2>("registers + ds_write_b128", src, d_out, d_sink, iters);
    run<2, 4>("registers + ds_write_b128", src, d_out, d_sink, iters);
    run<3, 1>("registers + scale/shift/ReLU + ds_write_b128", src, d_out, d_sink, iters);
    run<3, 2>("registers + scale/shift/ReLU + ds_write_b128", src, d_out, d_sink, iters);
    run<3, 4>("registers + scale/shift/ReLU + ds_write_b128", src, d_out, d_sink, iters);
    for (int skew : {2, 4, 8}) {  // 32 / 64 / 128 cycles between the waves of a CU
        printf("skew %d x 16 cycles per wave:\n", skew);
        run<1, 1>("LDS-DMA", src, d_out, d_sink, iters, skew);
        run<1, 4>("LDS-DMA", src, d_out, d_sink, iters, skew);
        run<2, 4>("registers + ds_write_b128", src, d_out, d_sink, iters, skew);
    }
    run<0, 0, 512>("MFMA only", src, d_out, d_sink, iters);
    run<1, 1, 512>("LDS-DMA", src, d_out, d_sink, iters);
    run<1, 2, 512>("LDS-DMA", src, d_out, d_sink, iters);
    run<1, 4, 512>("LDS-DMA", src, d_out, d_sink, iters);
    run<2, 4, 512>("registers + ds_write_b128", src, d_out, d_sink, iters);
    run<3, 4, 512>("registers + scale/shift/ReLU + ds_write_b128", src, d_out, d_sink, iters);
    return 0;
}
