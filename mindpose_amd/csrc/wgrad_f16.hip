// Weight gradient of a convolution on the fp16 matrix cores (amp O2 training), channel-blocked fp16 operands.
//
//   dW[co][ci][t] = sum over pixels p of dz[p][co] * x[p*S + tap t][ci]        (fp32 accumulation, fp32 result)
//
// Per tap this is a GEMM with the PIXEL axis as K.  Both operands live in HBM channel-blocked (8 channels of one pixel =
// 16 B), i.e. K-strided for the MFMA, so the tiles are staged pixel-major in LDS - row = one position, 32 channels per row -
// and read with the gfx950 transposing LDS read ds_read_b64_tr_b16: a 16-lane group fetches 4 positions x 16 channels and
// every lane receives ITS channel at the 4 positions, exactly the k-contiguous fragment v_mfma_f32_16x16x32_f16 wants.
//
// Position trick: the dz tile is stored with the SAME row pitch P as the (halo-padded) x tile, its padding columns zero,
// so that for stride S the x position of tap (ty, tx) is  S*k + ty*P + tx  - linear in the dz position k = y*P + x
// ((S*y + ty)*P + S*x + tx).  The padding columns cost MFMA work on zeros (a few per cent at stride 1, half of the
// k-steps at stride 2, which only the rare down-sampling convs use) and remove all index arithmetic from the loop.
//
// Workgroup = 32 couts x 32 cins x all taps (wave = one 16x16 sub-tile, T accumulators); the pixel axis is split over
// gridDim.y slabs that a fixed-order reduce sums (deterministic, no atomics).  Tiles are double-buffered in LDS with the
// next tile's 16-byte range-checked buffer loads in flight during the MFMA loop.
#include "common.h"
#include "conv_f16_dev.h"

namespace mp {

namespace {

typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
struct frag8 {
    s16x4 lo, hi;
};

constexpr int kRowHalfs = 40;  // LDS row = 32 channels (64 B) + 16 B pad: rows stay 16-byte aligned, banks spread

struct Wgrad16Params {
    const void* x;
    const void* dz;
    float* slabs;
    int N, Cin, C8in, H, W, Cout, C8out, Ho, Wo, pad;
    int R, P, Px, Rin, K, xrows;
    int tiles_y, tiles, splits, tiles_per_split, ci_tiles;
    int nbuf;  // 2 = double-buffered tiles; 1 = one LDS buffer (wide stride-2 layers whose tiles do not fit twice)
    unsigned magic_wo, magic_w;
    // LDS-DMA form (conv_wgrad_f16_dma_kernel): channel-block-major LDS images, element counts per 8-channel block
    int xslots, zslots, pieces;  // == 4 (mod 16) each: the two blocks a 16-lane group reads land on disjoint banks
    int x_pieces, z_base;        // DMA pieces of the input image; first element of the gradient image (a multiple of 64)
    int wide;                    // LDS-DMA form, tile shape: 0 = 32 x 32 (cout x cin), 1 = 64 x 64 (wave = one 32 x 32 quarter), 2 = 64 x 16 (narrow)
    int co_tiles;
    int plane_slots;             // LDS-DMA form, stride 2: > 0 = the input rows of a tile lie in two row-parity planes of this many slots
    unsigned magic_px;
    unsigned magic_xs, magic_zs, magic_p;
    // grouped launch (mp_f16_conv_wgrad_grouped): blockIdx.z = job, up to kWgradJobs layers of ONE shape; the operand pointers
    // travel by value in the kernel arguments (nothing to upload, hipGraph-capturable); n_jobs == 0: the single-layer launch
    int n_jobs;
    const void* jx[8];
    const void* jdz[8];
    float* jdw[8];
};
constexpr int kWgradJobs = 8;

// The operand pointers of this workgroup's job (blockIdx.z).  Written as selects over CONSTANT indices: a dynamic index into the
// by-value argument struct made the compiler copy the whole struct to scratch memory, and every later read of a launch parameter
// became a scratch load - a vector-memory operation, whose s_waitcnt vmcnt(0) inside the tile loop also waited for the NEXT tile's
// LDS-DMA, i.e. serialised the copy the loop exists to overlap.
__device__ __forceinline__ void job_tensors(const Wgrad16Params& p, const void*& x, const void*& dz) {
    if (p.n_jobs == 0) return;
    const int job = blockIdx.z;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (job == j) { x = p.jx[j]; dz = p.jdz[j]; }
}

__device__ __forceinline__ s16x4 tr_read(const _Float16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

template <int KS, int S, int NZ, int NX>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_kernel(const Wgrad16Params p) {
    constexpr int T = KS * KS;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    const int zbuf = p.K * kRowHalfs, xbuf = p.xrows * kRowHalfs;  // halfs per buffer
    _Float16* lds_z = smem_h;                  // [nbuf][zbuf]
    _Float16* lds_x = smem_h + p.nbuf * zbuf;  // [nbuf][xbuf]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co_sub = wave & 1, ci_sub = wave >> 1;
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int co_tile = blockIdx.x / p.ci_tiles, ci_tile = blockIdx.x % p.ci_tiles;
    const int t_begin = blockIdx.y * p.tiles_per_split;
    const int t_end = min(t_begin + p.tiles_per_split, p.tiles);

    {
        u32x4* z = reinterpret_cast<u32x4*>(smem_h);
        const int n16 = (p.nbuf * (zbuf + xbuf)) >> 3;
        const u32x4 zero = (u32x4){0u, 0u, 0u, 0u};
        for (int i = tid; i < n16; i += 256) z[i] = zero;
    }

    // staging tables (tile independent): 16-byte units = (position, 8-channel block)
    int zoff[NZ], zdst[NZ], zrow[NZ];
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        const unsigned u = tid + 256 * i;
        zdst[i] = -1; zoff[i] = 0; zrow[i] = 0;
        if (u < (unsigned)(p.R * p.Wo * 4)) {
            const unsigned blk = u & 3, pos = u >> 2;
            const unsigned r = fastdiv(pos, p.Wo, p.magic_wo), xx = pos - r * p.Wo;
            const int cb = co_tile * 4 + (int)blk;
            if (cb < p.C8out) {
                zoff[i] = (int)(((unsigned)cb * p.Ho * p.Wo + r * p.Wo + xx) * 16u);
                zdst[i] = (int)((r * p.P + xx) * kRowHalfs + blk * 8);
                zrow[i] = (int)r;
            }
        }
    }
    int xoff[NX], xdst[NX], xrow[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const unsigned u = tid + 256 * i;
        xdst[i] = -1; xoff[i] = 0; xrow[i] = 0;
        if (u < (unsigned)(p.Rin * p.W * 4)) {
            const unsigned blk = u & 3, pos = u >> 2;
            const unsigned r = fastdiv(pos, p.W, p.magic_w), xx = pos - r * p.W;
            const int cb = ci_tile * 4 + (int)blk;
            if (cb < p.C8in) {
                xoff[i] = (int)(((unsigned)cb * p.H * p.W + r * p.W + xx) * 16u);
                xdst[i] = (int)((r * p.Px + xx + p.pad) * kRowHalfs + blk * 8);
                xrow[i] = (int)r;
            }
        }
    }

    const void* x_ptr = p.x;
    const void* dz_ptr = p.dz;
    job_tensors(p, x_ptr, dz_ptr);
    const __amdgpu_buffer_rsrc_t rs_z = make_rsrc(dz_ptr, (size_t)p.N * p.C8out * p.Ho * p.Wo * 16);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x_ptr, (size_t)p.N * p.C8in * p.H * p.W * 16);
    u32x4 vz[NZ], vx[NX];
    auto stage_load = [&](int t) {
        const int n = t / p.tiles_y, ty = t - n * p.tiles_y;
        const int y0 = ty * p.R, yin0 = y0 * S - p.pad;
        const int zb = (n * p.C8out * p.Ho * p.Wo + y0 * p.Wo) * 16;
        const int xb = (n * p.C8in * p.H * p.W + yin0 * p.W) * 16;
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const bool ok = zdst[i] >= 0 && y0 + zrow[i] < p.Ho;
            vz[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, ok ? (unsigned)(zb + zoff[i]) : kOob, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int yin = yin0 + xrow[i];
            const bool ok = xdst[i] >= 0 && yin >= 0 && yin < p.H;
            vx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(xb + xoff[i]) : kOob, 0, 0);
        }
    };
    auto stage_store = [&](int buf) {  // rows outside the tensor arrive as zeros and overwrite the previous tile's data
        _Float16* dzp = lds_z + buf * zbuf;
        _Float16* dxp = lds_x + buf * xbuf;
#pragma unroll
        for (int i = 0; i < NZ; ++i)
            if (zdst[i] >= 0) *reinterpret_cast<u32x4*>(dzp + zdst[i]) = vz[i];
#pragma unroll
        for (int i = 0; i < NX; ++i)
            if (xdst[i] >= 0) *reinterpret_cast<u32x4*>(dxp + xdst[i]) = vx[i];
    };

    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // lane-constant operand offsets (halfs): row (8g + q) of a k-step, this wave's 16 channels, 4 halfs per lane
    const int a_base = (8 * g + q) * kRowHalfs + co_sub * 16 + 4 * pp;
    const int b_base = S * (8 * g + q) * kRowHalfs + ci_sub * 16 + 4 * pp;

    if (t_begin < t_end) {
        stage_load(t_begin);
        __syncthreads();  // zero fill complete
        stage_store(0);
        __syncthreads();
    }
    const int ksteps = p.K >> 5;
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = p.nbuf == 2 ? ((t - t_begin) & 1) : 0;
        const bool more = t + 1 < t_end;
        if (more) stage_load(t + 1);
        const _Float16* zt = lds_z + buf * zbuf + a_base;
        const _Float16* xt = lds_x + buf * xbuf + b_base;
        for (int ks = 0; ks < ksteps; ++ks) {
            const _Float16* za = zt + ks * 32 * kRowHalfs;
            const _Float16* xb = xt + ks * 32 * S * kRowHalfs;
            frag8 a;
            a.lo = tr_read(za);
            a.hi = tr_read(za + 4 * kRowHalfs);
            const f16x8 af = __builtin_bit_cast(f16x8, a);
#pragma unroll
            for (int tp = 0; tp < T; ++tp) {
                const int off = ((tp / KS) * p.Px + (tp % KS)) * kRowHalfs;
                frag8 b;
                b.lo = tr_read(xb + off);
                b.hi = tr_read(xb + off + 4 * S * kRowHalfs);
                acc[tp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, b), acc[tp], 0, 0, 0);
            }
        }
        if (more) {
            if (p.nbuf == 1) __syncthreads();  // every wave is done reading the only buffer
            stage_store(p.nbuf == 2 ? (buf ^ 1) : 0);
            __syncthreads();
        }
    }

    // D: lane holds couts 4g .. 4g+3 (rows) of cin (lane & 15) (column) -> slab [Cout][Cin][T]
    float* slab = p.slabs + ((size_t)blockIdx.z * p.splits + blockIdx.y) * p.Cout * p.Cin * T;
    const int ci = ci_tile * 32 + ci_sub * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co_tile * 32 + co_sub * 16 + 4 * g + r;
        if (co < p.Cout && ci < p.Cin) {
#pragma unroll
            for (int tp = 0; tp < T; ++tp) slab[((size_t)co * p.Cin + ci) * T + tp] = acc[tp][r];
        }
    }
}

// one tile's LDS-DMA: the wave's pieces (64 consecutive elements each) of the input image, then of the gradient image.  A macro,
// expanded at its two call sites: as a lambda (or a function taking the descriptors) it made the HOST compilation pass drop the
// kernel's stub without a diagnostic - __amdgpu_buffer_rsrc_t cannot be captured / passed there.
#define WGRAD_DMA_TILE(DST, TILE)                                                                                               \
    do {                                                                                                                         \
        const int t_ = (TILE);                                                                                                   \
        u32x4* dst_ = (DST);                                                                                                     \
        const int n_ = t_ / p.tiles_y, ty_ = t_ - n_ * p.tiles_y;                                                                \
        const int y0_ = ty_ * p.R, yin0_ = y0_ * S - p.pad;                                                                      \
        const unsigned zb_ = (unsigned)((n_ * p.C8out * p.Ho * p.Wo + y0_ * p.Wo) * 16);                                         \
        const unsigned xb_ = (unsigned)((n_ * p.C8in * p.H * p.W + yin0_ * p.W) * 16);                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) {                                                                      \
            const int piece_ = wave + 4 * i_;                                                                                    \
            if (piece_ >= p.pieces) break; /* wave-uniform */                                                                    \
            const int r_ = piece_row[i_];                                                                                        \
            if (piece_ < p.x_pieces) {                                                                                           \
                const bool ok_ = piece_rel[i_] != kOob && yin0_ + r_ >= 0 && yin0_ + r_ < p.H;                                   \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(dst_ + piece_ * 64),   \
                                                         16, ok_ ? xb_ + piece_rel[i_] : kOob, 0, 0, 0);                         \
            } else {                                                                                                             \
                const bool ok_ = piece_rel[i_] != kOob && y0_ + r_ < p.Ho;                                                       \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_z, (__attribute__((address_space(3))) void*)(dst_ + piece_ * 64),   \
                                                         16, ok_ ? zb_ + piece_rel[i_] : kOob, 0, 0, 0);                         \
            }                                                                                                                    \
        }                                                                                                                        \
    } while (0)

// ---- LDS-DMA form -----------------------------------------------------------------------------------------------------------
// The kernel above stages both tiles global -> VGPR -> ds_write into a PIXEL-major LDS image (row = one position x 32 channels),
// zero-fills its LDS first and pays a load / wait / write / barrier chain per tile; the layers of the deep branches give a
// workgroup 1 - 4 tiles, so nothing hides that chain (28 - 40 us per layer for 7.25 GFLOP: the matrix pipe ~10 % busy).
// Here the LDS images keep the HBM layout - [8-channel block][position] x 16 B - so a tile is a plain copy: LDS-DMA
// (buffer_load ... lds), no staging registers, no ds_write, no zero fill (padding columns, rows outside the image and the K tail
// arrive as zeros through the buffer range check).  ds_read_b64_tr_b16 takes PER-LANE addresses, so the transposing read works on
// this image too: lane 4q+p of a 16-lane group points at position q, channels 4p .. 4p+3 = block (p >> 1), byte (p & 1) * 8.
// Same position trick, same MFMA loop order, same slab layout as above: results are bit-identical.
// (A __device__ body behind concrete __global__ wrappers: as a __global__ TEMPLATE containing the LDS-DMA builtin the host
// compilation pass of this file emitted no stub for it - undefined symbol at load time, no diagnostic.)
// LDS operand reads of the LDS-DMA kernel, as inline assembly.  Written with the builtin, every ds_read of the tile loop got a
// compiler-inserted s_waitcnt vmcnt(0) in front of it: the waitcnt pass cannot tell the stage being read from the stage the
// in-flight DMA (buffer_load ... lds) writes, so it waited for the NEXT tile's copy before the first MFMA of THIS tile - the two
// stages ran strictly one after the other (round-3 counters: waves 53 % waiting, matrix pipe 20 % busy).  An asm read is invisible
// to that pass; the hazards are handled by hand instead: vmcnt(0) + barrier at the top of a tile (the stage has landed, the other
// one is no longer read), and counted lgkmcnt waits - tied to the registers they guard - before each MFMA.
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr(unsigned addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void lds_landed(frag8& f) {  // at most N LDS reads issued after f's may still be in flight
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.lo), "+v"(f.hi) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_landed(frag8& f, frag8& g) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.lo), "+v"(f.hi), "+v"(g.lo), "+v"(g.hi) : "n"(N));
}

// one k-step's taps, unrolled by template recursion (the read offsets and wait counts must be immediates): the operand of tap
// TP + D is requested before tap TP's MFMA, so D fragments (2 D reads) are in flight behind the one being consumed
template <int KS, int S, int T, int D, int TP>
struct WgradTaps {
    template <int I>
    static __device__ __forceinline__ void request(frag8 (&b)[T], const unsigned (&xr)[KS]) {
        b[I].lo = lds_tr<(I % KS) * 16>(xr[I / KS]);
        b[I].hi = lds_tr<(I % KS) * 16 + 4 * S * 16>(xr[I / KS]);
    }
    static __device__ __forceinline__ void run(frag8& a, frag8 (&b)[T], f32x4 (&acc)[T], const unsigned (&xr)[KS]) {
        if constexpr (TP == 0) {
            constexpr int first = D < T ? D : T;
            prologue<0, first>(b, xr);
        }
        if constexpr (TP + D < T) request<TP + D>(b, xr);
        constexpr int behind = (T - 1 - TP) < D ? (T - 1 - TP) : D;
        if constexpr (TP == 0) lds_landed<2 * behind>(a, b[0]);
        else lds_landed<2 * behind>(b[TP]);
        acc[TP] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b[TP]), acc[TP], 0, 0, 0);
        if constexpr (TP + 1 < T) WgradTaps<KS, S, T, D, TP + 1>::run(a, b, acc, xr);
    }
    template <int I, int N>
    static __device__ __forceinline__ void prologue(frag8 (&b)[T], const unsigned (&xr)[KS]) {
        if constexpr (I < N) {
            request<I>(b, xr);
            prologue<I + 1, N>(b, xr);
        }
    }
};

// ---- wide form ----------------------------------------------------------------------------------------------------------------
// Above, a workgroup owns a 32 x 32 (cout x cin) tile and wave (co_sub, ci_sub) a 16 x 16 quarter of it: per 32 positions the
// workgroup copies 64 channels x 32 positions into LDS for 4 T MFMAs.  Whatever the layer, that is ~130 B of LDS-DMA per MFMA, and
// the copies - not the matrix pipe, not the LDS reads - set the pace: LDS holds two stages per workgroup, so half of it at most
// is in flight, and a stage is consumed in a fraction of the time its copy takes to arrive (every grouped launch of 58 GFLOP took
// 120 - 135 us, 32- to 256-channel layers alike).  Layers with more than 32 channels on both sides get a 64 x 64 tile here:
// wave (co_half, ci_half) computes a 32 x 32 quarter as 2 x 2 MFMA tiles per tap (4 T accumulators), 128 channels x 32 positions
// are copied for 16 T MFMAs - half the bytes per MFMA - and a 64-channel layer reads each tensor once instead of twice.
// NK k-steps are unrolled into one straight-line block (requests of the next k-step fly under the last taps of this one; nothing
// in flight is carried around a loop edge, where a compiler-inserted register copy would read it early).
template <int KS, int S, int T, int NK>
struct WgradTaps2 {
    static constexpr int NV = NK * T;        // virtual taps of the block: V = (k-step j of the block) * T + tap
    static constexpr int D = T >= 2 ? 2 : 1;  // request distance in virtual taps
    static constexpr int reads(int v) { return 4 + ((v % T) == 0 ? 4 : 0); }  // LDS reads request<v> issues
    static constexpr int behind(int v) {  // reads requested after tap v's, i.e. allowed in flight when v is consumed
        int n = 0;
        for (int j = 1; j < D; ++j)
            if (v + j < NV) n += reads(v + j);
        return n;
    }
    template <int V>
    static __device__ __forceinline__ void request(frag8 (&a)[NK][2], frag8 (&b)[NV][2], unsigned za0, unsigned za1,
                                                   const unsigned (&xr0)[KS], const unsigned (&xr1)[KS]) {
        constexpr int J = V / T, I = V % T;
        if constexpr (I == 0) {
            a[J][0].lo = lds_tr<J * 512>(za0);
            a[J][0].hi = lds_tr<J * 512 + 64>(za0);
            a[J][1].lo = lds_tr<J * 512>(za1);
            a[J][1].hi = lds_tr<J * 512 + 64>(za1);
        }
        constexpr int OFF = J * 512 * S + (I % KS) * 16;
        b[V][0].lo = lds_tr<OFF>(xr0[I / KS]);
        b[V][0].hi = lds_tr<OFF + 4 * S * 16>(xr0[I / KS]);
        b[V][1].lo = lds_tr<OFF>(xr1[I / KS]);
        b[V][1].hi = lds_tr<OFF + 4 * S * 16>(xr1[I / KS]);
    }
    template <int V>
    static __device__ __forceinline__ void run(frag8 (&a)[NK][2], frag8 (&b)[NV][2], f32x4 (&acc)[T][2][2], unsigned za0, unsigned za1,
                                               const unsigned (&xr0)[KS], const unsigned (&xr1)[KS]) {
        constexpr int J = V / T, I = V % T;
        if constexpr (V == 0) {
            request<0>(a, b, za0, za1, xr0, xr1);
            if constexpr (D == 2 && NV > 1) request<1>(a, b, za0, za1, xr0, xr1);
        }
        constexpr int N = behind(V);
        if constexpr (I == 0)
            asm volatile("s_waitcnt lgkmcnt(%8)"
                         : "+v"(a[J][0].lo), "+v"(a[J][0].hi), "+v"(a[J][1].lo), "+v"(a[J][1].hi), "+v"(b[V][0].lo), "+v"(b[V][0].hi),
                           "+v"(b[V][1].lo), "+v"(b[V][1].hi)
                         : "n"(N));
        else
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b[V][0].lo), "+v"(b[V][0].hi), "+v"(b[V][1].lo), "+v"(b[V][1].hi) : "n"(N));
        if constexpr (V + D < NV) request<V + D>(a, b, za0, za1, xr0, xr1);
#pragma unroll
        for (int is = 0; is < 2; ++is)
#pragma unroll
            for (int cs = 0; cs < 2; ++cs)
                acc[I][cs][is] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[J][cs]),
                                                                        __builtin_bit_cast(f16x8, b[V][is]), acc[I][cs][is], 0, 0, 0);
        if constexpr (V + 1 < NV) run<V + 1>(a, b, acc, za0, za1, xr0, xr1);
    }
};

// MODE 0: 32 x 32 tile, wave = (16 couts, 16 cins) quarter.  1 (wide): 64 x 64 tile, wave = 32 x 32 quarter (see above).
// 2 (narrow, layers with <= 16 input channels - the 3-channel stem): 64 couts x 16 cins, wave w = couts 16 w .. 16 w + 15: no wave
//   multiplies all-zero channels, the input rows are copied once for all 64 couts and only two channel blocks of them.
template <int KS, int S, int NP, int MODE = 0>
__device__ __forceinline__ void wgrad_dma_body(const Wgrad16Params& p) {
    constexpr int T = KS * KS;
    extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
    // per buffer: [4][xslots] input image, padding up to a whole DMA piece, [4][zslots] gradient image, padding likewise
    const int buf_units = p.pieces * 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool WIDE = MODE == 1;
    const int co_sub = MODE == 2 ? wave : (wave & 1), ci_sub = MODE == 2 ? 0 : (wave >> 1);
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int co_tile = blockIdx.x / p.ci_tiles, ci_tile = blockIdx.x % p.ci_tiles;
    const int t_begin = blockIdx.y * p.tiles_per_split;
    const int t_end = min(t_begin + p.tiles_per_split, p.tiles);

    const void* x_ptr = p.x;
    const void* dz_ptr = p.dz;
    job_tensors(p, x_ptr, dz_ptr);
    const __amdgpu_buffer_rsrc_t rs_z = make_rsrc(dz_ptr, (size_t)p.N * p.C8out * p.Ho * p.Wo * 16);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x_ptr, (size_t)p.N * p.C8in * p.H * p.W * 16);

    // DMA piece descriptors, decoded once: piece = 64 consecutive slots of the buffer; slot -> (image kind, block, row, column)
    unsigned piece_rel[NP];  // byte offset relative to the tile origin of its tensor; kOob = padding slot
    int piece_row[NP];       // row within the tile, bit 30 set = gradient image
    constexpr int TBX = MODE == 1 ? 8 : MODE == 2 ? 2 : 4, TBZ = MODE == 0 ? 4 : 8;  // 8-channel blocks per tile side (input, gradient)
    const int x_units = TBX * p.xslots, z_units = TBZ * p.zslots;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int piece = wave + 4 * i;
        piece_rel[i] = kOob;
        piece_row[i] = 0;
        if (piece >= p.pieces) continue;  // wave-uniform
        if (piece < p.x_pieces) {
            const int s = piece * 64 + lane;
            const unsigned blk = fastdiv((unsigned)s, p.xslots, p.magic_xs);
            unsigned rem = s - blk * p.xslots;
            // stride 2: slots [0, plane_slots) hold the even tile rows, [plane_slots, 2 plane_slots) the odd ones (see geometry_dma)
            const unsigned plane = (S == 2 && p.plane_slots > 0 && rem >= (unsigned)p.plane_slots) ? 1u : 0u;
            rem -= plane * p.plane_slots;
            const unsigned rp = fastdiv(rem, p.Px, p.magic_px);
            const int c = (int)(rem - rp * p.Px) - p.pad;
            const unsigned r = (S == 2 && p.plane_slots > 0) ? 2 * rp + plane : rp;
            const int cb = ci_tile * TBX + (int)blk;
            if (s < x_units && r < (unsigned)p.Rin && c >= 0 && c < p.W && cb < p.C8in)
                piece_rel[i] = ((unsigned)cb * p.H * p.W + r * p.W + c) * 16u;
            piece_row[i] = (int)r;
        } else {
            const int sz = (piece - p.x_pieces) * 64 + lane;
            const unsigned blk = fastdiv((unsigned)sz, p.zslots, p.magic_zs);
            const unsigned rem = sz - blk * p.zslots;
            const unsigned r = fastdiv(rem, p.P, p.magic_p);
            const unsigned c = rem - r * p.P;
            const int cb = co_tile * TBZ + (int)blk;
            if (sz < z_units && r < (unsigned)p.R && c < (unsigned)p.Wo && cb < p.C8out)
                piece_rel[i] = ((unsigned)cb * p.Ho * p.Wo + r * p.Wo + c) * 16u;
            piece_row[i] = (int)r;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem16;  // LDS byte address of the stages
    unsigned row_off[KS];  // byte offset of tap row ky inside a channel block's input image
#pragma unroll
    for (int r = 0; r < KS; ++r)
        row_off[r] = (S == 2 && p.plane_slots > 0) ? (unsigned)(((r & 1) * p.plane_slots + (r >> 1) * p.Px) * 16) : (unsigned)(r * p.Px * 16);
    if constexpr (WIDE) {
        f32x4 acc[T][2][2];  // [tap][16-cout half][16-cin half] of this wave's 32 x 32 quarter
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i >> 1][i & 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // lane-constant operand byte offsets: position (8g + q) of a k-step, channels 4pp .. 4pp+3 of the FIRST 16 of this wave's 32
        // (blocks 4 co_sub .. of the tile); the second 16 channels are two 8-channel blocks further
        const int a_base = ((co_sub * 4 + (pp >> 1)) * p.zslots + (8 * g + q)) * 16 + (pp & 1) * 8 + p.z_base * 16;
        const int b_base = ((ci_sub * 4 + (pp >> 1)) * p.xslots + S * (8 * g + q)) * 16 + (pp & 1) * 8;
        const unsigned a_half = (unsigned)(2 * p.zslots * 16), b_half = (unsigned)(2 * p.xslots * 16);
        if (t_begin < t_end) WGRAD_DMA_TILE(smem16, t_begin);
        const int ksteps = p.K >> 5;
        for (int t = t_begin; t < t_end; ++t) {
            const int buf = (t - t_begin) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile has landed; the other stage is no longer read
            __syncthreads();
            if (t + 1 < t_end) WGRAD_DMA_TILE(smem16 + (buf ^ 1) * buf_units, t + 1);
            const unsigned stage = lds0 + (unsigned)(buf * buf_units * 16);
            int ks = 0;
            for (; ks + 1 < ksteps; ks += 2) {  // two k-steps per straight-line block
                const unsigned za0 = stage + (unsigned)(a_base + ks * 512), xb = stage + (unsigned)(b_base + ks * 512 * S);
                unsigned xr0[KS], xr1[KS];
#pragma unroll
                for (int r = 0; r < KS; ++r) { xr0[r] = xb + row_off[r]; xr1[r] = xr0[r] + b_half; }
                frag8 a[2][2], b[2 * T][2];
                WgradTaps2<KS, S, T, 2>::template run<0>(a, b, acc, za0, za0 + a_half, xr0, xr1);
            }
            if (ks < ksteps) {
                const unsigned za0 = stage + (unsigned)(a_base + ks * 512), xb = stage + (unsigned)(b_base + ks * 512 * S);
                unsigned xr0[KS], xr1[KS];
#pragma unroll
                for (int r = 0; r < KS; ++r) { xr0[r] = xb + row_off[r]; xr1[r] = xr0[r] + b_half; }
                frag8 a[1][2], b[T][2];
                WgradTaps2<KS, S, T, 1>::template run<0>(a, b, acc, za0, za0 + a_half, xr0, xr1);
            }
        }
        // D: lane holds couts 4g .. 4g+3 (rows) of cin (lane & 15) (column) -> slab [Cout][Cin][T]
        float* slab = p.slabs + ((size_t)blockIdx.z * p.splits + blockIdx.y) * p.Cout * p.Cin * T;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cs = i >> 1, is = i & 1;
            const int ci = ci_tile * 64 + ci_sub * 32 + is * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co_tile * 64 + co_sub * 32 + cs * 16 + 4 * g + r;
                if (co < p.Cout && ci < p.Cin) {
#pragma unroll
                    for (int tp = 0; tp < T; ++tp) slab[((size_t)co * p.Cin + ci) * T + tp] = acc[tp][cs][is][r];
                }
            }
        }
        return;
    }
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // lane-constant operand byte offsets: position (8g + q) of a k-step, channels 4pp .. 4pp+3 of this wave's 16
    const int a_base = ((co_sub * 2 + (pp >> 1)) * p.zslots + (8 * g + q)) * 16 + (pp & 1) * 8 + p.z_base * 16;
    const int b_base = ((ci_sub * 2 + (pp >> 1)) * p.xslots + S * (8 * g + q)) * 16 + (pp & 1) * 8;

    if (t_begin < t_end) WGRAD_DMA_TILE(smem16, t_begin);
    const int ksteps = p.K >> 5;
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        // this tile has landed (every wave waits for its own pieces, the barrier publishes them); the other buffer is free
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < t_end)  // flies under this tile's MFMA loop
            WGRAD_DMA_TILE(smem16 + (buf ^ 1) * buf_units, t + 1);
        unsigned za = lds0 + (unsigned)(buf * buf_units * 16 + a_base);
        unsigned xb = lds0 + (unsigned)(buf * buf_units * 16 + b_base);
        for (int ks = 0; ks < ksteps; ++ks, za += 32 * 16, xb += 32 * S * 16) {
            frag8 a, b[T];
            a.lo = lds_tr<0>(za);
            a.hi = lds_tr<4 * 16>(za);
            unsigned xr[KS];
#pragma unroll
            for (int r = 0; r < KS; ++r) xr[r] = xb + row_off[r];
            WgradTaps<KS, S, T, (T < 3 ? T : 3), 0>::run(a, b, acc, xr);
        }
    }

    // D: lane holds couts 4g .. 4g+3 (rows) of cin (lane & 15) (column) -> slab [Cout][Cin][T]
    float* slab = p.slabs + ((size_t)blockIdx.z * p.splits + blockIdx.y) * p.Cout * p.Cin * T;
    const int ci = ci_tile * (MODE == 2 ? 16 : 32) + ci_sub * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co_tile * (MODE == 2 ? 64 : 32) + co_sub * 16 + 4 * g + r;
        if (co < p.Cout && ci < p.Cin) {
#pragma unroll
            for (int tp = 0; tp < T; ++tp) slab[((size_t)co * p.Cin + ci) * T + tp] = acc[tp][r];
        }
    }
}

struct WgradDst {  // destinations of a grouped reduce: blockIdx.y = job
    float* dw[8];
};

__global__ __launch_bounds__(256) void wgrad16_reduce_kernel(const float* __restrict__ slabs, WgradDst dst, size_t count,
                                                             int splits, float scale, int accumulate) {
    float* __restrict__ dw = dst.dw[blockIdx.y];
    slabs += (size_t)blockIdx.y * splits * count;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= splits; k += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) part[j] += slabs[(size_t)(k + j) * count + i];
        }
        for (int j = 0; k < splits; ++k, ++j) part[j] += slabs[(size_t)k * count + i];
        const float v = (((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]))) * scale;
        dw[i] = accumulate ? dw[i] + v : v;
    }
}

// The same reduction for SMALL weights: with |dW| = 9 K floats (the 32-channel layers) the kernel above is 36 blocks of
// threads that each walk 512 slabs serially - latency-bound at ~25 us.  Here a block serves 256 / G elements with G threads
// per element, thread g summing slabs g, g + G, ... (four interleaved partial sums), then a fixed-order combine through LDS
// (deterministic): G times the blocks, 1 / G of the serial depth.
template <int G>
__global__ __launch_bounds__(256) void wgrad16_reduce_grouped_kernel(const float* __restrict__ slabs, WgradDst dst,
                                                                     size_t count, int splits, float scale, int accumulate) {
    float* __restrict__ dw = dst.dw[blockIdx.y];
    slabs += (size_t)blockIdx.y * splits * count;
    constexpr int E = 256 / G;
    __shared__ float sm[G][E];
    const int e = threadIdx.x % E, g = threadIdx.x / E;
    const size_t i = (size_t)blockIdx.x * E + e;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < count) {
        int k = g, j = 0;
        for (; k + 3 * G < splits; k += 4 * G) {
#pragma unroll
            for (int u = 0; u < 4; ++u) part[u] += slabs[(size_t)(k + u * G) * count + i];
        }
        for (; k < splits; k += G, ++j) part[j] += slabs[(size_t)k * count + i];
    }
    sm[g][e] = (part[0] + part[1]) + (part[2] + part[3]);
    __syncthreads();
    if (g == 0 && i < count) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < G; ++q) v += sm[q][e];
        v *= scale;
        dw[i] = accumulate ? dw[i] + v : v;
    }
}

constexpr int kDmaPieces = 28;  // DMA pieces per wave and tile the LDS-DMA kernel is built for
constexpr int kDmaPiecesK = 12;  // ... and its wide form (4 T accumulators per lane: the piece descriptors must stay small)
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_dma_k1s1(const Wgrad16Params p) { wgrad_dma_body<1, 1, kDmaPieces>(p); }
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_dma_k1s2(const Wgrad16Params p) { wgrad_dma_body<1, 2, kDmaPieces>(p); }
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_dma_k3s1(const Wgrad16Params p) { wgrad_dma_body<3, 1, kDmaPieces>(p); }
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_dma_k3s2(const Wgrad16Params p) { wgrad_dma_body<3, 2, kDmaPieces>(p); }
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_dma_k4s2(const Wgrad16Params p) { wgrad_dma_body<4, 2, kDmaPieces>(p); }
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_dmak_k1s1(const Wgrad16Params p) { wgrad_dma_body<1, 1, kDmaPiecesK, 1>(p); }
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_dmak_k1s2(const Wgrad16Params p) { wgrad_dma_body<1, 2, kDmaPiecesK, 1>(p); }
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_dmak_k3s1(const Wgrad16Params p) { wgrad_dma_body<3, 1, kDmaPiecesK, 1>(p); }
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_dmak_k3s2(const Wgrad16Params p) { wgrad_dma_body<3, 2, kDmaPiecesK, 1>(p); }
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_dman_k3s1(const Wgrad16Params p) { wgrad_dma_body<3, 1, kDmaPieces, 2>(p); }
__global__ __launch_bounds__(256, 2) void conv_wgrad_f16_dman_k3s2(const Wgrad16Params p) { wgrad_dma_body<3, 2, kDmaPieces, 2>(p); }

// LDS-DMA form of the same decomposition: two buffers of [4][xslots] + [4][zslots] elements; false = does not fit (the
// register-staged kernel serves the shape).  MP_WGRAD16_DMA=0 switches it off (A/B).
bool geometry_dma(Wgrad16Params& p, int KS, int S, size_t& lds_bytes) {
    if (const char* e = knob("MP_WGRAD16_DMA"))
        if (atoi(e) == 0) return false;
    // wide form (64 x 64 tile per workgroup): 1x1 / 3x3 layers with more than 32 channels on both sides
    bool wide = KS <= 3 && p.Cin > 32 && p.Cout > 32;
    if (const char* e = knob("MP_WGRAD16_WIDE")) wide = wide && atoi(e) != 0;
    // Stride 2: the position trick needs (input position) = 2 x (gradient position) + (tap offset), i.e. the gradient rows on the
    // pitch of the INPUT rows - with one linear input image that is 2 Wo + 2 positions per gradient row of Wo, half of every
    // k-step multiplying zeros.  With the tile's input rows split by parity into two planes (tile row i -> plane i & 1, row i >> 1;
    // tap row ky reads plane ky & 1 from row ky >> 1 on) the input pitch Px belongs to ONE plane and the gradient pitch is Px / 2.
    bool planes = S == 2;
    if (const char* e = knob("MP_WGRAD16_PLANES")) planes = planes && atoi(e) != 0;
    const int P0 = p.P, Px0 = p.Px;
    if (planes) {
        p.P = (p.W + 2 * p.pad + 1) / 2 > p.Wo ? (p.W + 2 * p.pad + 1) / 2 : p.Wo;
        p.Px = 2 * p.P;
    }
    // narrow form (64 couts x 16 cins): 3x3 layers with at most 16 input channels
    bool narrow = KS == 3 && p.Cin <= 16 && p.Cout > 16;
    if (const char* e = knob("MP_WGRAD16_NARROW")) narrow = narrow && atoi(e) != 0;
    for (int w = wide ? 1 : narrow ? 2 : 0; w >= 0; w = w == 2 ? 0 : w - 1) {
        const int tbx = w == 1 ? 8 : w == 2 ? 2 : 4, tbz = w ? 8 : 4;  // 8-channel blocks per tile side (input, gradient)
        const int np = w == 1 ? kDmaPiecesK : kDmaPieces;
        for (int pass = 0; pass < 2; ++pass) {
            const size_t budget = pass == 0 ? 78 * 1024 : 150 * 1024;
            for (int R = p.Ho < 16 ? p.Ho : 16; R >= 1; --R) {
                const int Rin = (R - 1) * S + KS;
                const int K = (R * p.P + 31) / 32 * 32;
                int xneed = S * (K - 1) + (KS - 1) * p.Px + (KS - 1) + 1;
                if (xneed < Rin * p.Px) xneed = Rin * p.Px;
                int plane_slots = 0;
                if (planes) {  // per plane: the furthest position a k-step reads, and every row the plane holds
                    plane_slots = S * (K - 1) + ((KS - 1) >> 1) * p.Px + (KS - 1) + 1;
                    if (plane_slots < ((Rin + 1) >> 1) * p.Px) plane_slots = ((Rin + 1) >> 1) * p.Px;
                    xneed = 2 * plane_slots;
                }
                // == 4 (mod 16): the two channel blocks a 16-lane group reads are 64 B apart modulo the 256-byte bank row
                const int xslots = (xneed + 11) / 16 * 16 + 4, zslots = K + 4 + ((K % 16) ? 16 - K % 16 : 0);
                const int x_pieces = (tbx * xslots + 63) / 64, z_pieces = (tbz * zslots + 63) / 64;
                const int pieces = x_pieces + z_pieces;
                const size_t bytes = (size_t)2 * pieces * 64 * 16;
                if (pieces > 4 * np || bytes > budget) continue;
                p.R = R; p.Rin = Rin; p.K = K; p.xrows = xneed; p.nbuf = 2;
                p.xslots = xslots; p.zslots = zslots; p.pieces = pieces; p.x_pieces = x_pieces; p.z_base = x_pieces * 64;
                p.magic_xs = magic_of(xslots); p.magic_zs = magic_of(zslots); p.magic_p = magic_of(p.P); p.magic_px = magic_of(p.Px);
                p.plane_slots = plane_slots;
                p.wide = w;
                lds_bytes = bytes;
                return true;
            }
        }
    }
    p.P = P0; p.Px = Px0;  // the register-staged kernel keeps the single linear image
    return false;
}

constexpr int kNZ = 4, kNX = 10;

int geometry(const mp_conv_desc* d, Wgrad16Params& p, size_t& lds_bytes, int n_jobs = 1) {
    if (!d) return MP_ERR_NULL;
    if (d->n <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0 || d->conv_h <= 0 || d->conv_w <= 0) return MP_ERR_SHAPE;
    // 1x1 / 3x3 with padding k/2, or the 4x4 stride-2 padding-1 form (the transposed convolution's weight gradient)
    if (d->kh != d->kw || !(d->kh == 1 || d->kh == 3 || d->kh == 4)) return MP_ERR_UNSUPPORTED;
    if (!(d->stride == 1 || d->stride == 2)) return MP_ERR_UNSUPPORTED;
    if (d->pad_top != d->pad_left) return MP_ERR_UNSUPPORTED;
    if (d->kh == 4 ? (d->stride != 2 || d->pad_top != 1) : (d->pad_top != d->kh / 2)) return MP_ERR_UNSUPPORTED;
    const int S = d->stride, KS = d->kh;
    p.N = d->n; p.Cin = d->cin; p.C8in = (d->cin + 7) / 8; p.H = d->h; p.W = d->w;
    p.Cout = d->cout; p.C8out = (d->cout + 7) / 8; p.Ho = d->conv_h; p.Wo = d->conv_w; p.pad = d->pad_top;
    if ((long long)p.N * p.C8in * p.H * p.W * 16 >= 0x7FFFFFF0LL || (long long)p.N * p.C8out * p.Ho * p.Wo * 16 >= 0x7FFFFFF0LL)
        return MP_ERR_UNSUPPORTED;
    // common pitch: a row holds the Wo gradient columns and the W + 2*pad input columns
    p.P = p.W + 2 * p.pad > p.Wo ? p.W + 2 * p.pad : p.Wo;
    p.Px = p.P;
    // pass 0: double-buffered, two workgroups per CU; pass 1: double-buffered, whatever fits; pass 2: single buffer
    p.pieces = 0;
    p.wide = 0;
    p.plane_slots = 0;
    bool found = geometry_dma(p, KS, S, lds_bytes);  // the LDS-DMA form where its tiles fit (p.pieces > 0 marks it)
    for (int pass = 0; pass < 3 && !found; ++pass) {
        p.nbuf = pass < 2 ? 2 : 1;
        const size_t budget = pass == 0 ? 78 * 1024 : 150 * 1024;
        for (int R = p.Ho < 16 ? p.Ho : 16; R >= 1; --R) {
            p.R = R;
            p.Rin = (R - 1) * S + KS;
            p.K = (R * p.P + 31) / 32 * 32;
            p.xrows = S * (p.K - 1) + (KS - 1) * p.Px + (KS - 1) + 1;
            if (p.xrows < p.Rin * p.Px) p.xrows = p.Rin * p.Px;
            lds_bytes = (size_t)p.nbuf * (p.K + p.xrows) * kRowHalfs * 2;
            if (R * p.Wo * 4 <= kNZ * 256 && p.Rin * p.W * 4 <= kNX * 256 && lds_bytes <= budget) { found = true; break; }
        }
    }
    if (!found) return MP_ERR_UNSUPPORTED;
    p.tiles_y = (p.Ho + p.R - 1) / p.R;
    p.tiles = p.N * p.tiles_y;
    const int tw_o = p.wide ? 64 : 32, tw_i = p.wide == 1 ? 64 : p.wide == 2 ? 16 : 32;  // channels per tile side
    p.ci_tiles = (p.Cin + tw_i - 1) / tw_i;
    p.co_tiles = (p.Cout + tw_o - 1) / tw_o;
    const int ct = p.co_tiles * p.ci_tiles;
    int target = 512;  // two workgroups per CU; the slab reduce reads splits x |dW| floats, so no finer than that
    if (const char* e = knob("MP_WGRAD16_WGS")) {  // experiments: total workgroups per launch
        const int v = atoi(e);
        if (v >= 1) target = v;
    }
    // a grouped launch spreads ~768 workgroups over its layers: per layer fewer, longer pixel slabs - the slab traffic (written here,
    // re-read by the reduce) and the per-workgroup prologue / epilogue shrink by the group size
    // (a 64 x 64 tile's slab is four times the bytes: the wide form stays at ~512 workgroups)
    int splits = n_jobs > 1 ? (p.wide == 1 ? target : target + target / 2) / (ct * n_jobs) : target / ct;
    if (splits < 1) splits = 1;
    if (splits > p.tiles) splits = p.tiles;
    p.tiles_per_split = (p.tiles + splits - 1) / splits;
    p.splits = (p.tiles + p.tiles_per_split - 1) / p.tiles_per_split;
    p.magic_wo = magic_of(p.Wo);
    p.magic_w = magic_of(p.W);
    return MP_OK;
}

template <int KS, int S>
int launch_wgrad16_dma(const Wgrad16Params& p, size_t lds, hipStream_t s) {
    auto kern = KS == 4 ? conv_wgrad_f16_dma_k4s2
                        : KS == 3 ? (S == 1 ? conv_wgrad_f16_dma_k3s1 : conv_wgrad_f16_dma_k3s2)
                                  : (S == 1 ? conv_wgrad_f16_dma_k1s1 : conv_wgrad_f16_dma_k1s2);
    auto kern_k = KS == 3 ? (S == 1 ? conv_wgrad_f16_dmak_k3s1 : conv_wgrad_f16_dmak_k3s2)
                          : (S == 1 ? conv_wgrad_f16_dmak_k1s1 : conv_wgrad_f16_dmak_k1s2);
    static AttrOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    if (p.wide == 1 && KS <= 3) kern = kern_k;
    if (p.wide == 2 && KS == 3) {
        kern = S == 1 ? conv_wgrad_f16_dman_k3s1 : conv_wgrad_f16_dman_k3s2;
        static AttrOnce attr_n;
        if (attr_n.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipGetLastError();
        }
    }
    hipLaunchKernelGGL(kern, dim3(p.co_tiles * p.ci_tiles, p.splits, p.n_jobs ? p.n_jobs : 1), dim3(256), lds, s, p);
    return check_launch();
}

template <int KS, int S>
int launch_wgrad16(const Wgrad16Params& p, size_t lds, hipStream_t s) {
    if (p.pieces > 0) return launch_wgrad16_dma<KS, S>(p, lds, s);
    auto kern = conv_wgrad_f16_kernel<KS, S, kNZ, kNX>;
    static AttrOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(p.co_tiles * p.ci_tiles, p.splits, p.n_jobs ? p.n_jobs : 1), dim3(256), lds, s, p);
    return check_launch();
}

// launches + fixed-order slab reduction of 1 .. kWgradJobs layers of one shape
int wgrad16_run(const mp_conv_desc* desc, Wgrad16Params& p, size_t lds, const WgradDst& dst, int jobs, float scale, int accumulate,
                hipStream_t s) {
    int rc;
    if (desc->kh == 4) rc = launch_wgrad16<4, 2>(p, lds, s);
    else if (desc->kh == 3) rc = desc->stride == 1 ? launch_wgrad16<3, 1>(p, lds, s) : launch_wgrad16<3, 2>(p, lds, s);
    else rc = desc->stride == 1 ? launch_wgrad16<1, 1>(p, lds, s) : launch_wgrad16<1, 2>(p, lds, s);
    if (rc != MP_OK) return rc;
    const size_t count = (size_t)p.Cout * p.Cin * desc->kh * desc->kw;
    if (count * 16 <= 147456 * 2 && p.splits >= 64) {  // <= 18 K weights (32-channel 3x3): 16 threads per element
        hipLaunchKernelGGL(wgrad16_reduce_grouped_kernel<16>, dim3((unsigned)((count + 15) / 16), jobs), dim3(256), 0, s, p.slabs, dst, count,
                           p.splits, scale, accumulate ? 1 : 0);
    } else if (count * 4 <= 147456 * 2 && p.splits >= 16) {  // <= 74 K weights (64-channel 3x3, the 1x1 convs): 4 per element
        hipLaunchKernelGGL(wgrad16_reduce_grouped_kernel<4>, dim3((unsigned)((count + 63) / 64), jobs), dim3(256), 0, s, p.slabs, dst, count,
                           p.splits, scale, accumulate ? 1 : 0);
    } else {
        size_t blocks = (count + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(wgrad16_reduce_kernel, dim3((unsigned)blocks, jobs), dim3(256), 0, s, p.slabs, dst, count, p.splits, scale,
                           accumulate ? 1 : 0);
    }
    return check_launch();
}

}  // namespace
}  // namespace mp

using namespace mp;

extern "C" {

size_t mp_f16_conv_wgrad_workspace_bytes(const mp_conv_desc* desc) {
    Wgrad16Params p{};
    size_t lds = 0;
    if (geometry(desc, p, lds) != MP_OK) return 0;
    return (size_t)p.splits * p.Cout * p.Cin * desc->kh * desc->kw * sizeof(float);
}

int mp_f16_conv_wgrad(const mp_conv_desc* desc, const void* x, const void* dz, float* dw, float scale, int accumulate,
                      void* workspace, size_t workspace_bytes, mp_stream_t stream) {
    if (!x || !dz || !dw) return MP_ERR_NULL;
    Wgrad16Params p{};
    size_t lds = 0;
    int rc = geometry(desc, p, lds);
    if (rc != MP_OK) return rc;
    const size_t count = (size_t)p.Cout * p.Cin * desc->kh * desc->kw;
    if (!workspace || workspace_bytes < (size_t)p.splits * count * sizeof(float)) return MP_ERR_WORKSPACE;
    p.x = x; p.dz = dz; p.slabs = reinterpret_cast<float*>(workspace);
    WgradDst dst{};
    dst.dw[0] = dw;
    return wgrad16_run(desc, p, lds, dst, 1, scale, accumulate, as_stream(stream));
}

size_t mp_f16_conv_wgrad_grouped_workspace_bytes(const mp_conv_desc* desc, int n_jobs) {
    if (n_jobs < 1 || n_jobs > kWgradJobs) return 0;
    Wgrad16Params p{};
    size_t lds = 0;
    if (geometry(desc, p, lds, n_jobs) != MP_OK) return 0;
    return (size_t)n_jobs * p.splits * p.Cout * p.Cin * desc->kh * desc->kw * sizeof(float);
}

int mp_f16_conv_wgrad_grouped(const mp_conv_desc* desc, const void* const* x, const void* const* dz, float* const* dw, int n_jobs,
                              float scale, int accumulate, void* workspace, size_t workspace_bytes, mp_stream_t stream) {
    if (!x || !dz || !dw) return MP_ERR_NULL;
    if (n_jobs < 1 || n_jobs > kWgradJobs) return MP_ERR_SHAPE;
    Wgrad16Params p{};
    size_t lds = 0;
    int rc = geometry(desc, p, lds, n_jobs);
    if (rc != MP_OK) return rc;
    const size_t count = (size_t)p.Cout * p.Cin * desc->kh * desc->kw;
    if (!workspace || workspace_bytes < (size_t)n_jobs * p.splits * count * sizeof(float)) return MP_ERR_WORKSPACE;
    WgradDst dst{};
    for (int j = 0; j < n_jobs; ++j) {
        if (!x[j] || !dz[j] || !dw[j]) return MP_ERR_NULL;
        p.jx[j] = x[j]; p.jdz[j] = dz[j]; p.jdw[j] = dw[j];
        dst.dw[j] = dw[j];
    }
    p.n_jobs = n_jobs;
    p.slabs = reinterpret_cast<float*>(workspace);
    return wgrad16_run(desc, p, lds, dst, n_jobs, scale, accumulate, as_stream(stream));
}

}  // extern "C"
