// Two chained fp16 1x1 convolutions of HRNet's stage 1 in ONE launch (round 4): the EXPAND conv of Bottleneck i
// (hrnet.py:107-123, 126-146: y = relu(bn3(conv3 m) + identity), 64 -> 256 channels) and the REDUCE conv of Bottleneck i + 1
// (z = relu(bn1(conv1 y)), 256 -> 64).
//
// Why: at N = 128 the two launches are HBM-bound and run ALONE on the chip (stage 1 has no branches: 0.8 of the 3.4 ms amp-O2 step,
// tools/timeline.py) - the expand conv moves 50 + 201 (identity) + 201 (y) MB in 84 us, the reduce conv reads those 201 MB of y
// again and writes 50 MB in 48 us (5.3 TB/s both).  A 1x1 conv has no halo: a workgroup that holds a pixel tile of y in LDS can
// run the next conv on it at once.  Here a workgroup takes 64 consecutive pixels of one image:
//   * the 8 channel planes of m (64 x 16 B each = ONE LDS-DMA piece per plane) -> LDS;
//   * GEMM 1 (K = 64: two k-steps): wave w owns output channels 64 w .. 64 w + 63 for all four pixel tiles of 16 (its weight
//     fragments - 8 x 16 B per lane - come straight from L2); epilogue: scale / shift, + identity, ReLU, ONE rounding - y leaves
//     for HBM in 16-byte channel-block elements AND is written to LDS in the operand layout of the next GEMM ([plane][pixel]);
//   * GEMM 2 (K = 256: eight k-steps, 64 output channels): a wave owns two pixel tiles x one pair of cout tiles (16 weight fragments
//     through a ring of two k-steps); epilogue: scale / shift, ReLU, 16-byte stores of z.
// 40 KB of LDS and <= 168 registers: three workgroups per CU hide each other's memory latency.  HBM traffic 50 + 201 + 201 +
// 50 MB instead of 703.  Same operand mapping, k order and epilogue arithmetic as the stand-alone kernels: y and z are bit-identical
// to two mp_f16_conv2d_fwd launches (tests/test_gpu_f16.py::test_expand_reduce_chain_equals_two_convs).
#include "conv_f16.h"
#include "conv_f16_dev.h"

namespace mp {

namespace {

constexpr int kPT = 64;  // pixels per workgroup

__device__ __forceinline__ void pw_barrier() {
    // this wave's LDS traffic done, then the workgroup barrier; global loads stay in flight (a __syncthreads() would drain them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// CM / CE / CR: channels of m, y, z (multiples of 32 / 64 / 64).  DUAL: the second conv reads m as well (z = act1(conv1x1(m; w1))):
// the down-sample conv and the reduce conv of the FIRST Bottleneck (hrnet.py:74-81, 107-123), both on the block's 64-channel input -
// one read of it, one launch; no identity, y is not staged in LDS.
// DS: the identity of the expand conv is the block's down-sample conv (hrnet.py:74-81: conv1x1(x0; wd) * scale_d + shift_d, no
// ReLU) of the block's input x0, computed HERE as a GEMM of the same shape in front of GEMM 1 and rounded to fp16 exactly as the
// stand-alone conv stores it - its 256-channel output is neither written nor read back (402 of the first block's 854 MB at N = 128).
template <int CM, int CE, int CR, bool DUAL = false, bool DS = false>
__global__ __launch_bounds__(256, DS ? 2 : 3) void expand_reduce_f16_kernel(const PwChainParams p) {
    static_assert(CE == 256 && CR == 64 && CM % 32 == 0, "four waves x 64 expanded channels; 64 reduced channels");
    static_assert(!(DUAL && DS), "one form at a time");
    constexpr int KQ1 = CM / 32, KQ2 = DUAL ? CM / 32 : CE / 32, CM8 = CM / 8, CE8 = CE / 8, CR8 = CR / 8, PS = kPT / 16, CS = 4;
    extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
    u32x4* __restrict__ lds_m = smem16;             // [CM8][64]
    u32x4* __restrict__ lds_y = smem16 + CM8 * kPT;  // [CE8][64]
    [[maybe_unused]] u32x4* __restrict__ lds_x0 = lds_y + CE8 * kPT;  // DS: [CM8][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;

    int b = blockIdx.x;
    {
        const int nb = p.total_blocks, q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, j = b >> 3;
        b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const unsigned n = (unsigned)b / (unsigned)p.tiles_per_img;
    const unsigned p0 = ((unsigned)b - n * (unsigned)p.tiles_per_img) * kPT;
    const unsigned HW = (unsigned)p.HW;

    // ---- m tile: plane pl of this image, pixels p0 .. p0 + 63 = one DMA piece; wave w stages planes w, w + 4, ...
    {
        const __amdgpu_buffer_rsrc_t rs_m = make_rsrc(p.mid, (size_t)p.N * CM8 * HW * 16);
#pragma unroll
        for (int j = 0; j < (CM8 + 3) / 4; ++j) {
            const int pl = wave + 4 * j;
            if (pl < CM8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m, (__attribute__((address_space(3))) void*)(lds_m + pl * kPT), 16,
                                                         ((n * CM8 + pl) * HW + p0 + lane) * 16u, 0, 0, 0);
        }
        if constexpr (DS) {
            const __amdgpu_buffer_rsrc_t rs_x0 = make_rsrc(p.x0, (size_t)p.N * CM8 * HW * 16);
#pragma unroll
            for (int j = 0; j < (CM8 + 3) / 4; ++j) {
                const int pl = wave + 4 * j;
                if (pl < CM8)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, (__attribute__((address_space(3))) void*)(lds_x0 + pl * kPT), 16,
                                                             ((n * CM8 + pl) * HW + p0 + lane) * 16u, 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- GEMM 1 operands: this wave's 64 expanded channels (packed 1x1 weights [k-step][4][CE] x 16 B), the identity, scale / shift
    const __amdgpu_buffer_rsrc_t rs_w3 = make_rsrc(p.w3, (size_t)KQ1 * 4 * CE * 16);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_wd = make_rsrc(DS ? p.wd : p.w3, (size_t)KQ1 * 4 * CE * 16);
    u32x4 A1[KQ1][CS];  // DS: the down-sample weights first, the expand weights behind GEMM 0
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        const unsigned off = (unsigned)(lq * CE + 64 * wave + f16_a_row<CS>(cs, lr)) * 16u;
#pragma unroll
        for (int q = 0; q < KQ1; ++q) A1[q][cs] = __builtin_amdgcn_raw_buffer_load_b128(DS ? rs_wd : rs_w3, off + (unsigned)q * (4u * CE * 16u), 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // a PAIR of cout tiles (2 j, 2 j + 1) gives a lane the 8 channels of ONE channel block of one pixel: plane 8 w + 4 j + lq
    const size_t e_bytes = (size_t)p.N * CE8 * HW * 16;
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.res, e_bytes), rs_y = make_rsrc(p.y, e_bytes);
    const unsigned off_e0 = ((n * CE8 + 8 * wave + lq) * HW + p0 + lr) * 16u;  // (ps, j): + 256 ps + 64 HW j (registers are short here)
    u32x4 r[PS][2];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            r[ps][j] = (DUAL || DS) ? (u32x4){0u, 0u, 0u, 0u} : __builtin_amdgcn_raw_buffer_load_b128(rs_r, off_e0 + 256u * ps + 64u * HW * j, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // the DMA pieces and the weight fragments are OLDER than the 2 PS identity loads: those stay in flight across the barrier
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DUAL || DS) ? 0 : 2 * PS) : "memory");
    pw_barrier();

    if constexpr (DS) {
        // ---- GEMM 0: the identity = the down-sample conv of x0 (same shape, k order and epilogue arithmetic as the stand-alone conv;
        //      ONE rounding to fp16 - what that conv would store), left in the identity registers
        f32x4 acc0[PS][CS];
#pragma unroll
        for (int ps = 0; ps < PS; ++ps)
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) acc0[ps][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KQ1; ++q)
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) {
                const u32x4 bv = lds_x0[(4 * q + lq) * kPT + ps * 16 + lr];
#pragma unroll
                for (int cs = 0; cs < CS; ++cs)
                    acc0[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A1[q][cs]), __builtin_bit_cast(f16x8, bv),
                                                                          acc0[ps][cs], 0, 0, 0);
            }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = 64 * wave + 32 * j + 8 * lq;
            const f32x4 sc_lo = *reinterpret_cast<const f32x4*>(p.scale_d + co), sc_hi = *reinterpret_cast<const f32x4*>(p.scale_d + co + 4);
            const f32x4 sh_lo = *reinterpret_cast<const f32x4*>(p.shift_d + co), sh_hi = *reinterpret_cast<const f32x4*>(p.shift_d + co + 4);
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) {
                const u32x2 lo = f16_pack4(f16_epi4(acc0[ps][2 * j], sc_lo, sh_lo, false, (u32x2){0u, 0u}, false, (u32x2){0u, 0u}, 0));
                const u32x2 hi = f16_pack4(f16_epi4(acc0[ps][2 * j + 1], sc_hi, sh_hi, false, (u32x2){0u, 0u}, false, (u32x2){0u, 0u}, 0));
                r[ps][j] = (u32x4){lo.x, lo.y, hi.x, hi.y};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the expand weights replace the down-sample weights (requested behind the epilogue: with the accumulators of GEMM 0 still
        // live the kernel does not fit the 168 registers of three workgroups per CU - those hide this round trip)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            const unsigned off = (unsigned)(lq * CE + 64 * wave + f16_a_row<CS>(cs, lr)) * 16u;
#pragma unroll
            for (int q = 0; q < KQ1; ++q) A1[q][cs] = __builtin_amdgcn_raw_buffer_load_b128(rs_w3, off + (unsigned)q * (4u * CE * 16u), 0, 0);
        }
    }

    f32x4 acc1[PS][CS];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) acc1[ps][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KQ1; ++q)
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const u32x4 bv = lds_m[(4 * q + lq) * kPT + ps * 16 + lr];
#pragma unroll
            for (int cs = 0; cs < CS; ++cs)
                acc1[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A1[q][cs]), __builtin_bit_cast(f16x8, bv),
                                                                      acc1[ps][cs], 0, 0, 0);
        }
    // ---- epilogue 1: y = relu(acc * scale + shift + identity), one rounding; to HBM and to the LDS operand tile of GEMM 2
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        // scale / shift of the pair's 8 channels: fetched here (2 KB that every workgroup reads: cache hits), not held across GEMM 1
        const int co = 64 * wave + 32 * j + 8 * lq;  // = f16_d_cout<CS>(2 j, lq); the odd tile of the pair: + 4
        const f32x4 sc_lo = *reinterpret_cast<const f32x4*>(p.scale3 + co), sc_hi = *reinterpret_cast<const f32x4*>(p.scale3 + co + 4);
        const f32x4 sh_lo = *reinterpret_cast<const f32x4*>(p.shift3 + co), sh_hi = *reinterpret_cast<const f32x4*>(p.shift3 + co + 4);
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const u32x4 a1 = r[ps][j];
            const u32x2 lo = f16_pack4(f16_epi4(acc1[ps][2 * j], sc_lo, sh_lo, !DUAL, (u32x2){a1.x, a1.y}, false, (u32x2){0u, 0u}, p.relu3));
            const u32x2 hi = f16_pack4(f16_epi4(acc1[ps][2 * j + 1], sc_hi, sh_hi, !DUAL, (u32x2){a1.z, a1.w}, false,
                                                (u32x2){0u, 0u}, p.relu3));
            const u32x4 v = (u32x4){lo.x, lo.y, hi.x, hi.y};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_y, off_e0 + 256u * ps + 64u * HW * j, 0, 0);
            if constexpr (!DUAL) lds_y[(8 * wave + 4 * j + lq) * kPT + ps * 16 + lr] = v;
        }
    }
    // ---- GEMM 2 (K = CE): wave = (pixel half wp, cout half wc): pixel tiles 2 wp, 2 wp + 1 x ONE pair of cout tiles (32 channels) -
    //      a quarter of the weight stream per wave of the all-couts split, 16-byte stores.  The fragments walk through a ring of two
    //      k-steps (k-step q + 2 replaces k-step q right behind its MFMAs)
    constexpr int CS2 = 2, PS2 = 2;
    const int wp = wave & 1, wc = wave >> 1;
    const __amdgpu_buffer_rsrc_t rs_w1 = make_rsrc(p.w1, (size_t)KQ2 * 4 * CR * 16);
    unsigned a2_off[CS2];
#pragma unroll
    for (int cs = 0; cs < CS2; ++cs) a2_off[cs] = (unsigned)(lq * CR + 32 * wc + f16_a_row<CS2>(cs, lr)) * 16u;
    u32x4 A2[2][CS2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int cs = 0; cs < CS2; ++cs) A2[q][cs] = __builtin_amdgcn_raw_buffer_load_b128(rs_w1, a2_off[cs] + (unsigned)q * (4u * CR * 16u), 0, 0);
    const int co2 = 32 * wc + 8 * lq;  // the pair's channel block of this lane
    const f32x4 sc1_lo = *reinterpret_cast<const f32x4*>(p.scale1 + co2), sc1_hi = *reinterpret_cast<const f32x4*>(p.scale1 + co2 + 4);
    const f32x4 sh1_lo = *reinterpret_cast<const f32x4*>(p.shift1 + co2), sh1_hi = *reinterpret_cast<const f32x4*>(p.shift1 + co2 + 4);
    pw_barrier();  // every wave's part of the y tile is in LDS

    f32x4 acc2[PS2][CS2];
#pragma unroll
    for (int ps = 0; ps < PS2; ++ps)
#pragma unroll
        for (int cs = 0; cs < CS2; ++cs) acc2[ps][cs] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KQ2; ++q) {
#pragma unroll
        for (int ps = 0; ps < PS2; ++ps) {
            const u32x4 bv = (DUAL ? lds_m : lds_y)[(4 * q + lq) * kPT + (2 * wp + ps) * 16 + lr];
#pragma unroll
            for (int cs = 0; cs < CS2; ++cs)
                acc2[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A2[q & 1][cs]), __builtin_bit_cast(f16x8, bv),
                                                                      acc2[ps][cs], 0, 0, 0);
        }
        if (q + 2 < KQ2) {
#pragma unroll
            for (int cs = 0; cs < CS2; ++cs)
                A2[q & 1][cs] = __builtin_amdgcn_raw_buffer_load_b128(rs_w1, a2_off[cs] + (unsigned)(q + 2) * (4u * CR * 16u), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const __amdgpu_buffer_rsrc_t rs_z = make_rsrc(p.z, (size_t)p.N * CR8 * HW * 16);
#pragma unroll
    for (int ps = 0; ps < PS2; ++ps) {
        const u32x2 lo = f16_pack4(f16_epi4(acc2[ps][0], sc1_lo, sh1_lo, false, (u32x2){0u, 0u}, false, (u32x2){0u, 0u}, p.relu1));
        const u32x2 hi = f16_pack4(f16_epi4(acc2[ps][1], sc1_hi, sh1_hi, false, (u32x2){0u, 0u}, false, (u32x2){0u, 0u}, p.relu1));
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){lo.x, lo.y, hi.x, hi.y}, rs_z,
                                               ((n * CR8 + 4 * wc + lq) * HW + p0 + (2 * wp + ps) * 16 + lr) * 16u, 0, 0);
    }
}

}  // namespace

int pwchain_build(const void* mid, const void* res, const void* w3, const float* scale3, const float* shift3, int relu3, const void* w1,
                  const float* scale1, const float* shift1, int relu1, void* y, void* z, int n, int cm, int ce, int cr, int h, int w,
                  PwChainLaunch& L, const void* x0, const void* wd, const float* scale_d, const float* shift_d) {
    const bool ds = x0 != nullptr;
    if (!mid || (!res && !ds) || !w3 || !scale3 || !shift3 || !w1 || !scale1 || !shift1 || !y || !z) return MP_ERR_NULL;
    if (ds && (res || !wd || !scale_d || !shift_d)) return MP_ERR_NULL;  // exactly one source of the identity
    if (n <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    if (cm != 64 || ce != 256 || cr != 64) return MP_ERR_UNSUPPORTED;  // HRNet's stage 1 (hrnet.py:377-385: Bottleneck, 64 channels, x 4)
    L.dual = !ds && res == mid;  // the caller's mark for the two-convs-on-one-input form (mp_f16_dual_pw_fwd passes the input twice)
    L.ds = ds;
    const long long hw = (long long)h * w;
    if (hw % kPT != 0) return MP_ERR_UNSUPPORTED;  // a pixel tile never straddles images
    if ((long long)n * (ce / 8) * hw * 16 >= 0x7FFFFFF0LL) return MP_ERR_UNSUPPORTED;  // 32-bit buffer offsets
    PwChainParams& p = L.p;
    p.mid = mid; p.res = res; p.w3 = w3; p.scale3 = scale3; p.shift3 = shift3; p.w1 = w1; p.scale1 = scale1; p.shift1 = shift1;
    p.y = y; p.z = z; p.x0 = x0; p.wd = wd; p.scale_d = scale_d; p.shift_d = shift_d;
    p.N = n; p.HW = (int)hw; p.relu3 = relu3 ? 1 : 0; p.relu1 = relu1 ? 1 : 0;
    p.tiles_per_img = (int)(hw / kPT);
    p.total_blocks = n * p.tiles_per_img;
    L.cm = cm; L.ce = ce; L.cr = cr; L.h = h; L.w = w;
    L.lds_bytes = (size_t)(cm / 8 + (L.dual ? 0 : ce / 8) + (ds ? cm / 8 : 0)) * kPT * 16;
    return MP_OK;
}

int pwchain_launch(const PwChainLaunch& L, hipStream_t s) {
    if (g_dry_launch) return MP_OK;
    if (L.ds) hipLaunchKernelGGL((expand_reduce_f16_kernel<64, 256, 64, false, true>), dim3(L.p.total_blocks), dim3(256), L.lds_bytes, s, L.p);
    else if (L.dual) hipLaunchKernelGGL((expand_reduce_f16_kernel<64, 256, 64, true>), dim3(L.p.total_blocks), dim3(256), L.lds_bytes, s, L.p);
    else hipLaunchKernelGGL((expand_reduce_f16_kernel<64, 256, 64, false>), dim3(L.p.total_blocks), dim3(256), L.lds_bytes, s, L.p);
    return check_launch();
}

}  // namespace mp

using namespace mp;

extern "C" int mp_f16_expand_reduce_fwd(const void* mid, const void* res, const void* packed_w3, const float* scale3, const float* shift3,
                                        int relu3, const void* packed_w1, const float* scale1, const float* shift1, int relu1, void* y,
                                        void* z, int n, int cm, int ce, int cr, int h, int w, mp_stream_t stream) {
    PwChainLaunch L{};
    const int rc = pwchain_build(mid, res, packed_w3, scale3, shift3, relu3, packed_w1, scale1, shift1, relu1, y, z, n, cm, ce, cr, h, w, L);
    if (rc != MP_OK) return rc;
    return pwchain_launch(L, as_stream(stream));
}

extern "C" int mp_f16_dual_pw_fwd(const void* x, const void* packed_wa, const float* scale_a, const float* shift_a, int relu_a,
                                  const void* packed_wb, const float* scale_b, const float* shift_b, int relu_b, void* ya, void* zb, int n,
                                  int cm, int ce, int cr, int h, int w, mp_stream_t stream) {
    PwChainLaunch L{};
    const int rc = pwchain_build(x, x, packed_wa, scale_a, shift_a, relu_a, packed_wb, scale_b, shift_b, relu_b, ya, zb, n, cm, ce, cr, h, w, L);
    if (rc != MP_OK) return rc;
    return pwchain_launch(L, as_stream(stream));
}

extern "C" int mp_f16_ds_expand_reduce_fwd(const void* mid, const void* x0, const void* packed_wd, const float* scale_d, const float* shift_d,
                                           const void* packed_w3, const float* scale3, const float* shift3, int relu3, const void* packed_w1,
                                           const float* scale1, const float* shift1, int relu1, void* y, void* z, int n, int cm, int ce,
                                           int cr, int h, int w, mp_stream_t stream) {
    if (!x0) return MP_ERR_NULL;
    PwChainLaunch L{};
    const int rc = pwchain_build(mid, nullptr, packed_w3, scale3, shift3, relu3, packed_w1, scale1, shift1, relu1, y, z, n, cm, ce, cr, h, w, L,
                                 x0, packed_wd, scale_d, shift_d);
    if (rc != MP_OK) return rc;
    return pwchain_launch(L, as_stream(stream));
}
