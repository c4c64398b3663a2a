// C-ABI entry points of the convolution family: weight packing, tile-configuration choice,
// launch, and the recorded launch plan (one native call replays a whole network forward).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "conv_f16.h"
#include "conv_mfma.h"
#include "conv_pw.h"
#include "conv_wino.h"
#include "conv_stem.h"
#include "conv_gemm.h"
#include "conv_small.h"
#include "pwchain_f32.h"

namespace mp {

static inline unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ULL / d) + 1u; }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// [Cout,Cin,kh,kw] (or the (py,px) 2x2 phase of a [Cin,Cout,4,4] transposed-conv weight)
//   -> [Cin_pad4/4][T][4][Cout_pad16], zero padded.
// source value of packed element (cout co, cin ci, tap t); callers have checked co < cout && ci < cin
__device__ __forceinline__ float pack_source(const float* __restrict__ w, int cout, int cin, int kh, int kw, int transposed,
                                             int py, int px, int co, int ci, int t) {
    const int ty = t / kw, tx = t % kw;
    if (transposed == 0) return w[(((size_t)co * cin + ci) * kh + ty) * kw + tx];
    if (transposed == 2) {
        // data gradient of a stride-1 conv whose weight is [cin, cout, kh, kw] in forward terms (this packed
        // conv's cout = forward cin): roles swapped, taps mirrored
        return w[(((size_t)ci * cout + co) * kh + (kh - 1 - ty)) * kw + (kw - 1 - tx)];
    }
    if (transposed == 3) {
        // data gradient of a 3x3 stride-2 pad-1 conv, output parity phase (py, px), as a 2x2 conv over dy:
        // dx[2a+p] = sum_t dy[a+t] * w[k(p,t)],  k(0,0)=1, k(0,1)=none, k(1,0)=2, k(1,1)=0
        const int ky = py == 0 ? (ty == 0 ? 1 : -1) : (ty == 0 ? 2 : 0);
        const int kx = px == 0 ? (tx == 0 ? 1 : -1) : (tx == 0 ? 2 : 0);
        return (ky < 0 || kx < 0) ? 0.f : w[(((size_t)ci * cout + co) * 3 + ky) * 3 + kx];
    }
    if (transposed == 4) {
        // data gradient of sub-pixel phase (py, px) of Conv2dTranspose k=4 s=2 p=1: roles swapped (this packed conv's
        // cout = the transposed conv's cin), the 2x2 phase taps mirrored; w = [cout, cin, 4, 4] in packed terms
        const int my = 1 - ty, mx = 1 - tx;
        const int ky = py == 0 ? 3 - 2 * my : 2 - 2 * my;
        const int kx = px == 0 ? 3 - 2 * mx : 2 - 2 * mx;
        return w[(((size_t)co * cin + ci) * 4 + ky) * 4 + kx];
    }
    // Conv2dTranspose k=4 s=2 p=1: out row 2m+py reads in row m-1+py+ty with kernel row
    // ky = 3-2*ty (py=0) or 2-2*ty (py=1); same along x.
    const int ky = py == 0 ? 3 - 2 * ty : 2 - 2 * ty;
    const int kx = px == 0 ? 3 - 2 * tx : 2 - 2 * tx;
    return w[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int cout,
                                                          int cin, int kh, int kw, int cin_pad4, int cout_pad16,
                                                          int transposed, int py, int px) {
    const int T = kh * kw;
    const size_t total = (size_t)cin_pad4 * T * cout_pad16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout_pad16);
        size_t r = i / cout_pad16;
        const int kq = (int)(r & 3);
        r >>= 2;
        const int t = (int)(r % T);
        const int q = (int)(r / T);
        const int ci = q * 4 + kq;
        float v = 0.f;
        if (co < cout && ci < cin) v = pack_source(w, cout, cin, kh, kw, transposed, py, px, co, ci, t);
        out[i] = v;
    }
}

// every fp32 weight packing of a training step in ONE launch (same job table as mp_f16_pack_weight_batch): block b serves the
// job whose block range holds it, a thread writes four consecutive output channels (one 16-byte store)
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const mp_f16_pack_job* __restrict__ jobs,
                                                                const unsigned* __restrict__ first_block, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first_block[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const mp_f16_pack_job jb = jobs[lo];
    const int T = jb.kh * jb.kw, cp = (jb.cout + 15) / 16 * 16, cin_pad4 = (jb.cin + 3) / 4 * 4;
    const unsigned u = (blockIdx.x - first_block[lo]) * 256u + threadIdx.x;
    if (jb.transposed >= 5) {  // Winograd forms (5: forward weight, 6: data gradient): a thread writes the 16 xi of one (cin, cout)
        if (u >= (unsigned)cin_pad4 * cp) return;
        const int co = (int)(u % cp), ci = (int)(u / cp);
        float uu[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) uu[k] = 0.f;
        if (co < jb.cout && ci < jb.cin) wino_transform_weight(jb.w, jb.cout, jb.cin, jb.transposed == 6, co, ci, uu);
        float4* o4 = reinterpret_cast<float4*>(jb.packed) + (size_t)u * 4;
#pragma unroll
        for (int a = 0; a < 4; ++a) o4[a] = make_float4(uu[a * 4], uu[a * 4 + 1], uu[a * 4 + 2], uu[a * 4 + 3]);
        return;
    }
    const unsigned units = (unsigned)cin_pad4 * T * (cp / 4);
    if (u >= units) return;
    const int c4 = (int)(u % (cp / 4)) * 4;
    unsigned r = u / (cp / 4);
    const int kq = (int)(r & 3);
    r >>= 2;
    const int t = (int)(r % T);
    const int ci = (int)(r / T) * 4 + kq;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = c4 + j;
        v[j] = (co < jb.cout && ci < jb.cin) ? pack_source(jb.w, jb.cout, jb.cin, jb.kh, jb.kw, jb.transposed, jb.phase_y, jb.phase_x, co, ci, t)
                                             : 0.f;
    }
    reinterpret_cast<float4*>(jb.packed)[u] = make_float4(v[0], v[1], v[2], v[3]);
}

static unsigned long long* g_stamp_buf = nullptr;
static size_t g_stamp_bytes = 0;

unsigned long long* conv_stamp_buffer(size_t need_bytes) { return (g_stamp_buf && need_bytes <= g_stamp_bytes) ? g_stamp_buf : nullptr; }

struct ConvLaunch {
    ConvKParams p;
    int ks, stride, variant;
    size_t lds_bytes;
    bool pointwise;  // variant kPointwise: the streaming 1x1 kernel (conv_pw_f32.hip), launch record in pw
    PwLaunch pw;
    bool gemm;       // variant kGemm: the blocked-GEMM 1x1 kernel (conv_gemm_f32.hip), launch record in gm
    GemmLaunch gm;
    bool small_k;    // variant kSmall: the K-split 3x3 kernel for small problems (conv_small_f32.hip), launch record in sm
    SmallLaunch sm;
};

constexpr int kGemm = V_COUNT + 2;   // forced-variant index of the blocked-GEMM 1x1 kernel (V_COUNT + 1 is the tuner's index of the Winograd form)
constexpr int kSmall = V_COUNT + 3;  // forced-variant index of the K-split small-problem 3x3 kernel (the tuner's index 11)
constexpr int kPointwise = V_COUNT;  // forced-variant index of the streaming 1x1 kernel (never chosen by the library heuristic)

static const int kLdsMax = 150 * 1024;

// LDS budget per workgroup: 78 KiB = two workgroups per CU (160 KiB LDS).  MP_CONV_LDS_KB overrides it
// (tuning experiments only).
static int lds_budget() {
    static int v = -1;
    if (v < 0) {
        const char* e = knob("MP_CONV_LDS_KB");
        int kb = e ? atoi(e) : 0;
        v = (kb >= 16 && kb <= 150) ? kb * 1024 : 78 * 1024;
    }
    return v;
}

static int plane_pad(int raw) {  // smallest value >= raw that is == 16 (mod 32)
    int v = (raw + 15) / 32 * 32 + 16;
    if (v - 32 >= raw) v -= 32;
    return v;
}

// geometry for a given variant; returns false when it cannot fit LDS
static bool configure(const mp_conv_desc& d, int variant, ConvLaunch& L) {
    int CT, PT;
    variant_dims(variant, CT, PT);
    if (d.kh == 7 && variant != V_CT64_PT192 && variant != V_CT32_PT192_H) return false;  // only these are built
    ConvKParams& p = L.p;
    const int S = d.stride, KS = d.kh;
    p.N = d.n; p.Cin = d.cin; p.H = d.h; p.W = d.w; p.Cout = d.cout;
    p.Cout_pad16 = round_up(d.cout, 16);
    p.Cin_pad4 = round_up(d.cin, 4);
    p.Ho = d.conv_h; p.Wo = d.conv_w; p.pad_t = d.pad_top; p.pad_l = d.pad_left;
    if (p.Wo > PT) return false;
    const int rows_fit = PT / p.Wo;
    if (rows_fit >= p.Ho) {
        p.R = p.Ho;
        p.G = PT / (p.Ho * p.Wo);
        if (p.G > p.N) p.G = p.N;
        if (p.G < 1) p.G = 1;
    } else {
        p.R = rows_fit;
        p.G = 1;
    }
    p.RWo = p.R * p.Wo;
    p.Rin = (p.R - 1) * S + KS;
    p.Wp = (p.Wo - 1) * S + KS;
    if (p.Wp < p.pad_l + p.W && p.pad_l + p.W - p.Wp <= 2) p.Wp = p.pad_l + p.W;  // copy whole rows (16-B units)
    p.img_plane = p.Rin * p.Wp;
    p.cin_plane = plane_pad(p.G * p.img_plane);
    p.ncols = p.W < p.Wp - p.pad_l ? p.W : p.Wp - p.pad_l;
    if (p.ncols < 1) return false;
    const int T = KS * KS;
    // staging units: 4 consecutive columns (one 16-B global load) when rows are 16-B aligned
    p.vec = ((p.W & 3) == 0 && (p.ncols & 3) == 0) ? 1 : 0;
    p.upr = p.vec ? p.ncols / 4 : p.ncols;
    p.upc = p.G * p.Rin * p.upr;
    // cin chunk: the largest multiple of 4 that divides Cin_pad4, fits the per-thread staging registers
    // (kNI / NW units of 16 B) and - double-buffered when there is more than one chunk - the LDS budget
    const bool light = variant_light(variant) && KS <= 3;
    const int nw = stage_nw(KS, light), ni = stage_ni(KS, light, p.vec != 0);
    // three workgroups per CU for the light variants (160 KiB / 3), two otherwise
    const int budget = (light && !knob("MP_CONV_LDS_KB")) ? 52 * 1024 : lds_budget();
    int best_ck = 0;
    for (int ck = 4; ck <= p.Cin_pad4 && ck <= 128; ck += 4) {
        if (p.Cin_pad4 % ck) continue;
        if ((long long)ck * p.upc > (long long)ni * 256) continue;
        if ((long long)ck * T * CT / 4 > (long long)nw * 256) continue;
        const int nbuf = ck < p.Cin_pad4 ? 2 : 1;
        const long long bytes = (long long)nbuf * (ck * p.cin_plane + ck * T * CT) * 4;
        if (bytes > budget && !(ck == 4 && bytes <= kLdsMax)) continue;
        best_ck = ck;
    }
    if (best_ck == 0) return false;
    p.CK = best_ck;
    p.n_chunks = p.Cin_pad4 / p.CK;
    p.nbuf = p.n_chunks > 1 ? 2 : 1;
    p.in_buf = p.CK * p.cin_plane;
    p.w_buf = p.CK * T * CT;
    p.n_ct = (p.Cout_pad16 + CT - 1) / CT;
    p.tiles_y = (p.G > 1 || p.R >= p.Ho) ? 1 : (p.Ho + p.R - 1) / p.R;
    p.tiles_n = (p.N + p.G - 1) / p.G;
    p.out_h = d.out_h; p.out_w = d.out_w; p.out_mul = d.out_mul; p.out_rep = d.out_rep;
    p.off_y = d.out_off_y; p.off_x = d.out_off_x; p.relu = d.relu;
    p.magic_upr = magic_of(p.upr);
    p.magic_upc = magic_of(p.upc);
    p.magic_rin = magic_of(p.Rin);
    p.magic_rwo = magic_of(p.RWo);
    p.magic_wo = magic_of(p.Wo);
    p.total_blocks = p.n_ct * p.tiles_y * p.tiles_n;
    L.ks = KS; L.stride = S; L.variant = variant;
    L.lds_bytes = (size_t)p.nbuf * (p.in_buf + p.w_buf) * 4;
    return L.lds_bytes <= (size_t)kLdsMax;
}

static int choose_variant(const mp_conv_desc& d, ConvLaunch& best, int forced = -1) {
    if (forced == kPointwise) {
        int rc = pw_configure(&d, best.pw);
        if (rc != MP_OK) return rc;
        best.pointwise = true;
        best.ks = 1; best.stride = 1; best.variant = kPointwise;
        best.lds_bytes = best.pw.lds_bytes;
        return MP_OK;
    }
    if (forced == kGemm) {
        int rc = gemm_configure(&d, best.gm);
        if (rc != MP_OK) return rc;
        best.gemm = true;
        best.ks = d.kh; best.stride = best.gm.stride; best.variant = kGemm;
        best.lds_bytes = best.gm.lds_bytes;
        return MP_OK;
    }
    if (forced == kSmall || forced == kSmall + 1) {  // + 1: the wide form (48 / 64 pixels per workgroup)
        int rc = small_configure(&d, best.sm, forced - kSmall);
        if (rc != MP_OK) return rc;
        best.small_k = true;
        best.ks = d.kh; best.stride = d.stride; best.variant = forced;
        best.lds_bytes = best.sm.lds_bytes;
        return MP_OK;
    }
    if (forced >= 0) {
        if (forced >= V_COUNT) return MP_ERR_UNSUPPORTED;
        return configure(d, forced, best) ? MP_OK : MP_ERR_UNSUPPORTED;
    }
    // cout tile by divisibility; pixel tile 192 unless that leaves the chip under-filled
    int order[V_COUNT];
    int n = 0;
    const int c = d.cout;
    if (d.kh == 7) { order[n++] = V_CT64_PT192; order[n++] = V_CT32_PT192_H; }
    else if (c % 64 == 0) { order[n++] = V_CT64_PT192; order[n++] = V_CT64_PT96; order[n++] = V_CT32_PT192; order[n++] = V_CT32_PT96; }
    else if (c % 48 == 0) { order[n++] = V_CT48_PT192; order[n++] = V_CT64_PT192; order[n++] = V_CT64_PT96; }
    else if (c <= 32) { order[n++] = V_CT32_PT192; order[n++] = V_CT32_PT96; }
    else if (c <= 48) { order[n++] = V_CT48_PT192; order[n++] = V_CT64_PT96; }
    else { order[n++] = V_CT64_PT192; order[n++] = V_CT64_PT96; order[n++] = V_CT32_PT192; }
    bool have = false;
    double best_score = 0;
    for (int i = 0; i < n; ++i) {
        ConvLaunch L{};
        if (!configure(d, order[i], L)) continue;
        int CT, PT;
        variant_dims(order[i], CT, PT);
        // useful fraction of the MFMA tile x how well the grid fills 256 CUs x 2 workgroups
        const double pix_eff = (double)(L.p.G * L.p.RWo) / PT;
        const double co_eff = (double)d.cout / (L.p.n_ct * CT);
        const double waves = (double)L.p.total_blocks / 512.0;
        const double fill = waves >= 1.0 ? waves / (double)((long long)waves + ((waves - (long long)waves) > 1e-9 ? 1 : 0)) : waves;
        const double big = (CT >= 48 ? 1.0 : 0.93) * (PT >= 192 ? 1.0 : 0.95);  // bigger tiles re-use LDS operands
        const double score = pix_eff * co_eff * fill * big;
        if (!have || score > best_score * 1.02) {
            best = L;
            best_score = score;
            have = true;
        }
    }
    return have ? MP_OK : MP_ERR_UNSUPPORTED;
}

static int validate_desc(const mp_conv_desc* d) {
    if (!d) return MP_ERR_NULL;
    if (d->n <= 0 || d->cin <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0) return MP_ERR_SHAPE;
    if (d->kh != d->kw) return MP_ERR_UNSUPPORTED;
    if (!(d->kh == 1 || d->kh == 2 || d->kh == 3 || d->kh == 7)) return MP_ERR_UNSUPPORTED;
    if (!(d->stride == 1 || d->stride == 2)) return MP_ERR_UNSUPPORTED;
    if ((d->kh == 2 && d->stride != 1) || (d->kh == 7 && d->stride != 2)) return MP_ERR_UNSUPPORTED;
    if (d->pad_top < 0 || d->pad_left < 0 || d->pad_top >= d->kh + 1 || d->pad_left >= d->kw + 1) return MP_ERR_SHAPE;
    if (d->conv_h <= 0 || d->conv_w <= 0 || d->out_h <= 0 || d->out_w <= 0) return MP_ERR_SHAPE;
    if (d->out_mul < 1 || d->out_rep < 1 || d->out_off_y < 0 || d->out_off_x < 0) return MP_ERR_SHAPE;
    if (d->flags & ~MP_CONV_SHARES_CUS) return MP_ERR_UNSUPPORTED;
    // output mapping must stay inside [out_h, out_w]
    if ((d->conv_h - 1) * d->out_mul + d->out_off_y + d->out_rep > d->out_h) return MP_ERR_SHAPE;
    if ((d->conv_w - 1) * d->out_mul + d->out_off_x + d->out_rep > d->out_w) return MP_ERR_SHAPE;
    if ((long long)d->n * d->cout * d->out_h * d->out_w >= (1LL << 40)) return MP_ERR_UNSUPPORTED;
    return MP_OK;
}

static int launch(const ConvLaunch& L0, hipStream_t s) {
    if (L0.pointwise) return pw_launch(L0.pw, s);
    if (L0.gemm) return gemm_launch(L0.gm, s);
    if (L0.small_k) return small_launch(L0.sm, s);
    ConvLaunch L = L0;
    L.p.dbg = (g_stamp_buf && (size_t)L.p.total_blocks * 64 <= g_stamp_bytes) ? g_stamp_buf : nullptr;
    switch (L.ks) {
        case 1: return launch_conv_k1(L.p, L.stride, L.variant, L.lds_bytes, s);
        case 2: return launch_conv_k2(L.p, L.stride, L.variant, L.lds_bytes, s);
        case 3: return L.stride == 1 ? launch_conv_k3s1(L.p, L.variant, L.lds_bytes, s)
                                     : launch_conv_k3s2(L.p, L.variant, L.lds_bytes, s);
        case 7: return launch_conv_k7(L.p, L.stride, L.variant, L.lds_bytes, s);
        default: return MP_ERR_UNSUPPORTED;
    }
}

static int build_launch(const mp_conv_desc* desc, const float* x, const float* w, const float* scale,
                        const float* shift, const float* res1, const float* res2, float* out, ConvLaunch& L,
                        int forced = -1) {
    int rc = validate_desc(desc);
    if (rc != MP_OK) return rc;
    if (!x || !w || !scale || !shift || !out) return MP_ERR_NULL;
    rc = choose_variant(*desc, L, forced);
    if (rc != MP_OK) return rc;
    if (L.pointwise) {
        if (res2) return MP_ERR_UNSUPPORTED;  // one residual tensor in the streaming kernel
        L.pw.p.x = x; L.pw.p.wp = w; L.pw.p.scale = scale; L.pw.p.shift = shift; L.pw.p.res1 = res1; L.pw.p.out = out;
        return MP_OK;
    }
    if (L.gemm) {
        L.gm.p.x = x; L.gm.p.wp = w; L.gm.p.scale = scale; L.gm.p.shift = shift; L.gm.p.res1 = res1; L.gm.p.res2 = res2; L.gm.p.out = out;
        return MP_OK;
    }
    if (L.small_k) {
        L.sm.p.x = x; L.sm.p.wp = w; L.sm.p.scale = scale; L.sm.p.shift = shift; L.sm.p.res1 = res1; L.sm.p.res2 = res2; L.sm.p.out = out;
        return MP_OK;
    }
    L.p.x = x; L.p.wp = w; L.p.scale = scale; L.p.shift = shift; L.p.res1 = res1; L.p.res2 = res2; L.p.out = out;
    return MP_OK;
}

}  // namespace mp

using namespace mp;

struct mp_plan {
    struct Entry {
        int kind;  // 0 conv, 1 maxpool, 2 fuse-sum; fp16 layout: 3 conv, 4 fuse-sum, 5 NCHW fp32 -> c8, 6 c8 -> NCHW fp32;
                   // 7 = all-lane barrier (no launch), 8 = fused fp16 BasicBlock, 9 = fp32 Winograd conv,
                   // 10 = fp16 expand + reduce 1x1 chain (stage 1), 11 = fp16 first conv from the fp32 image, 12 = fp32 first conv (streaming form),
                   // 13 = fp32 expand + reduce 1x1 chain (stage 1)
        int lane;  // execution lane: 0 = the caller's stream, 1..3 = the plan's own side streams
        ConvLaunch conv;
        ConvF16Launch conv16;
        BlockF16Launch block16;
        PwChainLaunch pwchain;
        PwChainF32Launch pwchain32;
        StemF16Launch stem16;
        StemF32Launch stem32;
        WinoLaunch wino;
        const void* t16[3];
        const void* x16;
        void* out16;
        const float* x;
        float* out;
        int n, c, h, w;
        const float* t[3];
        int s[3];
        int relu;
    };
    std::vector<Entry> entries;
    int cur_lane = 0;
    int max_lane = 0;
    // side streams / events of the multi-lane replay, created on first use (one plan = one device)
    mutable hipStream_t side[3] = {nullptr, nullptr, nullptr};
    mutable hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    mutable hipEvent_t ev_fork = nullptr;
    ~mp_plan() {
        for (auto& st : side)
            if (st) (void)hipStreamDestroy(st);
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
    }
};

static const int kPlanLanes = 4;

static int run_entry(const mp_plan::Entry& e, mp_stream_t stream) {
    switch (e.kind) {
        case 0: return launch(e.conv, as_stream(stream));
        case 1: return mp_maxpool3x3s2_same(e.x, e.out, e.n, e.c, e.h, e.w, stream);
        case 2: return mp_fuse_upsample_sum(e.x, e.t[0], e.s[0], e.t[1], e.s[1], e.t[2], e.s[2], e.out, e.n, e.c, e.h, e.w, e.relu, stream);
        case 3: return f16_launch(e.conv16, as_stream(stream));
        case 4:
            return mp_f16_fuse_upsample_sum(e.x16, e.t16[0], e.s[0], e.t16[1], e.s[1], e.t16[2], e.s[2], e.out16, e.n, e.c, e.h, e.w,
                                            e.relu, stream);
        case 5: return mp_f16_to_c8(e.x, e.out16, e.n, e.c, e.h, e.w, stream);
        case 6: return mp_f16_from_c8(e.x16, e.out, e.n, e.c, e.h, e.w, stream);
        case 7: return MP_OK;
        case 8: return blockf16_launch(e.block16, as_stream(stream));
        case 9: return wino_launch(e.wino, as_stream(stream));
        case 10: return pwchain_launch(e.pwchain, as_stream(stream));
        case 11: return stemf16_launch(e.stem16, as_stream(stream));
        case 12: return stemf32_launch(e.stem32, as_stream(stream));
        case 13: return pwchain32_launch(e.pwchain32, as_stream(stream));
        default: return MP_ERR_UNSUPPORTED;
    }
}

static int hip_rc(hipError_t e) {
    if (e == hipSuccess) return MP_OK;
    g_last_hip_error = (int)e;
    return MP_ERR_HIP;
}

// Multi-lane replay: independent sub-graphs (the HRNet branches, the rows of an exchange unit) are enqueued on different
// HIP streams so that their kernels overlap on the chip; "barrier" entries order every lane after every other one.  Works
// under stream capture too (fork / join through events), so a captured hipGraph keeps the parallel structure.
static int run_multi_lane(const mp_plan* plan, hipStream_t main_stream) {
    for (int i = 0; i < 3; ++i)
        if (!plan->side[i]) {
            int rc = hip_rc(hipStreamCreateWithFlags(&plan->side[i], hipStreamNonBlocking));
            if (rc != MP_OK) return rc;
        }
    for (int i = 0; i < kPlanLanes; ++i)
        if (!plan->ev[i]) {
            int rc = hip_rc(hipEventCreateWithFlags(&plan->ev[i], hipEventDisableTiming));
            if (rc != MP_OK) return rc;
        }
    if (!plan->ev_fork) {
        int rc = hip_rc(hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming));
        if (rc != MP_OK) return rc;
    }
    hipStream_t lanes[kPlanLanes] = {main_stream, plan->side[0], plan->side[1], plan->side[2]};
    const int nl = plan->max_lane + 1;
    int rc = hip_rc(hipEventRecord(plan->ev_fork, main_stream));
    for (int l = 1; l < nl && rc == MP_OK; ++l) rc = hip_rc(hipStreamWaitEvent(lanes[l], plan->ev_fork, 0));
    for (size_t i = 0; i < plan->entries.size() && rc == MP_OK; ++i) {
        const mp_plan::Entry& e = plan->entries[i];
        if (e.kind == 7) {
            for (int l = 0; l < nl && rc == MP_OK; ++l) rc = hip_rc(hipEventRecord(plan->ev[l], lanes[l]));
            for (int l = 0; l < nl && rc == MP_OK; ++l)
                for (int m = 0; m < nl && rc == MP_OK; ++m)
                    if (m != l) rc = hip_rc(hipStreamWaitEvent(lanes[l], plan->ev[m], 0));
        } else {
            rc = run_entry(e, reinterpret_cast<mp_stream_t>(lanes[e.lane]));
        }
    }
    // join: the caller's stream continues only after every side lane has drained
    for (int l = 1; l < nl && rc == MP_OK; ++l) {
        rc = hip_rc(hipEventRecord(plan->ev[l], lanes[l]));
        if (rc == MP_OK) rc = hip_rc(hipStreamWaitEvent(main_stream, plan->ev[l], 0));
    }
    return rc;
}

extern "C" {

size_t mp_conv_packed_weight_bytes(int cout, int cin, int kh, int kw) {
    if (cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return 0;
    return (size_t)round_up(cin, 4) * kh * kw * round_up(cout, 16) * sizeof(float);
}

int mp_conv_pack_weight(const float* w, float* packed, int cout, int cin, int kh, int kw, int transposed, int phase_y,
                        int phase_x, mp_stream_t stream) {
    if (!w || !packed) return MP_ERR_NULL;
    if (cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return MP_ERR_SHAPE;
    if (transposed == 5 || transposed == 6)  // Winograd forms of a 3x3 weight (packed size: mp_conv_winograd_packed_weight_bytes)
        return (kh == 3 && kw == 3) ? wino_pack_launch(w, packed, cout, cin, transposed == 6, as_stream(stream)) : MP_ERR_UNSUPPORTED;
    if (transposed < 0 || transposed > 4) return MP_ERR_UNSUPPORTED;
    if ((transposed == 1 || transposed == 3 || transposed == 4) && (kh != 2 || kw != 2 || phase_y < 0 || phase_y > 1 || phase_x < 0 || phase_x > 1))
        return MP_ERR_UNSUPPORTED;
    const int cin_pad4 = round_up(cin, 4), cout_pad16 = round_up(cout, 16);
    const size_t total = (size_t)cin_pad4 * kh * kw * cout_pad16;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, packed, cout, cin, kh, kw,
                       cin_pad4, cout_pad16, transposed, phase_y, phase_x);
    return check_launch();
}

int mp_conv_pack_weight_batch(const mp_f16_pack_job* jobs_dev, const unsigned* first_block_dev, int n_jobs, unsigned total_blocks,
                              mp_stream_t stream) {
    if (n_jobs == 0) return MP_OK;
    if (!jobs_dev || !first_block_dev) return MP_ERR_NULL;
    if (n_jobs < 0 || total_blocks == 0) return MP_ERR_SHAPE;
    hipLaunchKernelGGL(pack_weight_batch_kernel, dim3(total_blocks), dim3(256), 0, as_stream(stream), jobs_dev, first_block_dev,
                       n_jobs);
    return check_launch();
}

int mp_conv2d_fwd(const mp_conv_desc* desc, const float* x, const float* packed_w, const float* scale,
                  const float* shift, const float* res1, const float* res2, float* out, mp_stream_t stream) {
    ConvLaunch L{};
    int rc = build_launch(desc, x, packed_w, scale, shift, res1, res2, out, L);
    if (rc != MP_OK) return rc;
    return launch(L, as_stream(stream));
}

int mp_debug_set_stamp_buffer(void* dev_ptr, size_t bytes) {
#if MP_CONV_STAMPS
    g_stamp_buf = reinterpret_cast<unsigned long long*>(dev_ptr);
    g_stamp_bytes = dev_ptr ? bytes : 0;
    return MP_OK;
#else
    (void)dev_ptr; (void)bytes;
    return MP_ERR_UNSUPPORTED;
#endif
}

int mp_conv2d_fwd_variant(const mp_conv_desc* desc, int variant, const float* x, const float* packed_w, const float* scale,
                          const float* shift, const float* res1, const float* res2, float* out, mp_stream_t stream) {
    ConvLaunch L{};
    int rc = build_launch(desc, x, packed_w, scale, shift, res1, res2, out, L, variant);
    if (rc != MP_OK) return rc;
    return launch(L, as_stream(stream));
}

int mp_plan_add_conv_variant(mp_plan* plan, const mp_conv_desc* desc, int variant, const float* x, const float* packed_w,
                             const float* scale, const float* shift, const float* res1, const float* res2, float* out) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 0;
    int rc = build_launch(desc, x, packed_w, scale, shift, res1, res2, out, e.conv, variant);
    if (rc != MP_OK) return rc;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

static int build_deconv_gemm(const mp_conv_desc* desc, const float* x, const float* packed4, const float* scale, const float* shift,
                             float* out, ConvLaunch& L) {
    int rc = validate_desc(desc);
    if (rc != MP_OK) return rc;
    if (!x || !packed4 || !scale || !shift || !out) return MP_ERR_NULL;
    rc = gemm_configure_deconv(desc, L.gm);
    if (rc != MP_OK) return rc;
    L.gemm = true;
    L.ks = 2; L.stride = 1; L.variant = kGemm; L.lds_bytes = L.gm.lds_bytes;
    L.gm.p.x = x; L.gm.p.wp = packed4; L.gm.p.scale = scale; L.gm.p.shift = shift; L.gm.p.res1 = nullptr; L.gm.p.res2 = nullptr; L.gm.p.out = out;
    return MP_OK;
}

int mp_deconv4x4s2_gemm_supported(const mp_conv_desc* phase00_desc) {
    GemmLaunch L{};
    int rc = validate_desc(phase00_desc);
    return rc != MP_OK ? rc : gemm_configure_deconv(phase00_desc, L);
}

int mp_deconv4x4s2_gemm_fwd(const mp_conv_desc* phase00_desc, const float* x, const float* packed4, const float* scale,
                            const float* shift, float* out, mp_stream_t stream) {
    ConvLaunch L{};
    int rc = build_deconv_gemm(phase00_desc, x, packed4, scale, shift, out, L);
    if (rc != MP_OK) return rc;
    return launch(L, as_stream(stream));
}

int mp_plan_add_deconv4x4s2_gemm(mp_plan* plan, const mp_conv_desc* phase00_desc, const float* x, const float* packed4,
                                 const float* scale, const float* shift, float* out) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 0;
    int rc = build_deconv_gemm(phase00_desc, x, packed4, scale, shift, out, e.conv);
    if (rc != MP_OK) return rc;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

mp_plan* mp_plan_create(void) { return new (std::nothrow) mp_plan(); }

void mp_plan_destroy(mp_plan* plan) { delete plan; }

int mp_plan_add_conv(mp_plan* plan, const mp_conv_desc* desc, const float* x, const float* packed_w,
                     const float* scale, const float* shift, const float* res1, const float* res2, float* out) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 0;
    int rc = build_launch(desc, x, packed_w, scale, shift, res1, res2, out, e.conv);
    if (rc != MP_OK) return rc;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_maxpool(mp_plan* plan, const float* x, float* out, int n, int c, int h, int w) {
    if (!plan || !x || !out) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    mp_plan::Entry e{};
    e.kind = 1;
    e.x = x; e.out = out; e.n = n; e.c = c; e.h = h; e.w = w;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_fuse_sum(mp_plan* plan, const float* base, const float* t1, int s1, const float* t2, int s2,
                         const float* t3, int s3, float* out, int n, int c, int h, int w, int relu) {
    if (!plan || !base || !t1 || !out) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    mp_plan::Entry e{};
    e.kind = 2;
    e.x = base; e.out = out; e.n = n; e.c = c; e.h = h; e.w = w; e.relu = relu;
    e.t[0] = t1; e.t[1] = t2; e.t[2] = t3; e.s[0] = s1; e.s[1] = s2; e.s[2] = s3;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_conv_f16(mp_plan* plan, const mp_conv_desc* desc, int variant, const void* x, const void* packed_w,
                         const float* scale, const float* shift, const void* res1, const void* res2, void* out) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 3;
    int rc = f16_build_launch(desc, variant, x, packed_w, scale, shift, res1, res2, out, e.conv16);
    if (rc != MP_OK) return rc;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_conv_winograd(mp_plan* plan, const mp_conv_desc* desc, const float* x, const float* packed_u, const float* scale,
                              const float* shift, const float* res1, const float* res2, float* out) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 9;
    int rc = wino_configure(desc, e.wino);
    if (rc != MP_OK) return rc;
    if (!x || !packed_u || !scale || !shift || !out) return MP_ERR_NULL;
    e.wino.p.x = x; e.wino.p.u = packed_u; e.wino.p.scale = scale; e.wino.p.shift = shift; e.wino.p.res1 = res1; e.wino.p.res2 = res2;
    e.wino.p.out = out;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_basicblock_f16(mp_plan* plan, const void* x, const void* packed_w1, const float* scale1, const float* shift1,
                               const void* packed_w2, const float* scale2, const float* shift2, void* out, int n, int c, int h,
                               int w, int rows) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 8;
    int rc = blockf16_build(x, packed_w1, scale1, shift1, packed_w2, scale2, shift2, out, n, c, h, w, rows, e.block16);
    if (rc != MP_OK) return rc;
    e.n = n; e.c = c; e.h = h; e.w = w;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_expand_reduce_f16(mp_plan* plan, const void* mid, const void* res, const void* packed_w3, const float* scale3,
                                  const float* shift3, int relu3, const void* packed_w1, const float* scale1, const float* shift1,
                                  int relu1, void* y, void* z, int n, int cm, int ce, int cr, int h, int w) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 10;
    int rc = pwchain_build(mid, res, packed_w3, scale3, shift3, relu3, packed_w1, scale1, shift1, relu1, y, z, n, cm, ce, cr, h, w, e.pwchain);
    if (rc != MP_OK) return rc;
    e.n = n; e.c = ce; e.h = h; e.w = w;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_expand_reduce(mp_plan* plan, const float* mid, const float* res, const float* x0, const float* packed_wd, const float* scale_d,
                              const float* shift_d, const float* packed_w3, const float* scale3, const float* shift3, const float* packed_w1,
                              const float* scale1, const float* shift1, float* y, float* z, int n, int cm, int ce, int cr, int h, int w) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 13;
    int rc = pwchain32_build(mid, res, x0, packed_wd, scale_d, shift_d, packed_w3, scale3, shift3, packed_w1, scale1, shift1, y, z, n, cm, ce, cr,
                             h, w, e.pwchain32);
    if (rc != MP_OK) return rc;
    e.n = n; e.c = ce; e.h = h; e.w = w;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_ds_expand_reduce_f16(mp_plan* plan, const void* mid, const void* x0, const void* packed_wd, const float* scale_d,
                                     const float* shift_d, const void* packed_w3, const float* scale3, const float* shift3, int relu3,
                                     const void* packed_w1, const float* scale1, const float* shift1, int relu1, void* y, void* z, int n,
                                     int cm, int ce, int cr, int h, int w) {
    if (!plan || !x0) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 10;
    int rc = pwchain_build(mid, nullptr, packed_w3, scale3, shift3, relu3, packed_w1, scale1, shift1, relu1, y, z, n, cm, ce, cr, h, w, e.pwchain,
                           x0, packed_wd, scale_d, shift_d);
    if (rc != MP_OK) return rc;
    e.n = n; e.c = ce; e.h = h; e.w = w;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_dual_pw_f16(mp_plan* plan, const void* x, const void* packed_wa, const float* scale_a, const float* shift_a, int relu_a,
                            const void* packed_wb, const float* scale_b, const float* shift_b, int relu_b, void* ya, void* zb, int n, int cm,
                            int ce, int cr, int h, int w) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 10;
    int rc = pwchain_build(x, x, packed_wa, scale_a, shift_a, relu_a, packed_wb, scale_b, shift_b, relu_b, ya, zb, n, cm, ce, cr, h, w, e.pwchain);
    if (rc != MP_OK) return rc;
    e.n = n; e.c = ce; e.h = h; e.w = w;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_stem_conv(mp_plan* plan, const float* x, const float* weight, const float* scale, const float* shift, int relu, float* out,
                          int n, int h, int w) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 12;
    int rc = stemf32_build(x, weight, scale, shift, relu, out, n, h, w, e.stem32);
    if (rc != MP_OK) return rc;
    e.n = n; e.c = 64; e.h = h; e.w = w;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_stem_conv_f16(mp_plan* plan, const float* x, const float* weight, const float* scale, const float* shift, int relu,
                              void* out, int n, int h, int w) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 11;
    int rc = stemf16_build(x, weight, scale, shift, relu, out, n, h, w, e.stem16);
    if (rc != MP_OK) return rc;
    e.n = n; e.c = 64; e.h = h; e.w = w;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_fuse_sum_f16(mp_plan* plan, const void* base, const void* t1, int s1, const void* t2, int s2, const void* t3,
                             int s3, void* out, int n, int c, int h, int w, int relu) {
    if (!plan || !base || !t1 || !out) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    mp_plan::Entry e{};
    e.kind = 4;
    e.x16 = base; e.out16 = out; e.n = n; e.c = c; e.h = h; e.w = w; e.relu = relu;
    e.t16[0] = t1; e.t16[1] = t2; e.t16[2] = t3; e.s[0] = s1; e.s[1] = s2; e.s[2] = s3;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_add_layout_f16(mp_plan* plan, int to_c8, const void* x, void* out, int n, int c, int h, int w) {
    if (!plan || !x || !out) return MP_ERR_NULL;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return MP_ERR_SHAPE;
    mp_plan::Entry e{};
    e.kind = to_c8 ? 5 : 6;
    if (to_c8) { e.x = reinterpret_cast<const float*>(x); e.out16 = out; }
    else { e.x16 = x; e.out = reinterpret_cast<float*>(out); }
    e.n = n; e.c = c; e.h = h; e.w = w;
    e.lane = plan->cur_lane;
    plan->entries.push_back(e);
    return MP_OK;
}

int mp_plan_size(const mp_plan* plan) { return plan ? (int)plan->entries.size() : MP_ERR_NULL; }

int mp_plan_run_range(const mp_plan* plan, int first, int count, mp_stream_t stream) {
    if (!plan) return MP_ERR_NULL;
    if (first < 0 || count < 0 || (size_t)first + count > plan->entries.size()) return MP_ERR_SHAPE;
    for (int i = first; i < first + count; ++i) {
        int rc = run_entry(plan->entries[i], stream);  // one stream, in order: per-entry timing / profiling
        if (rc != MP_OK) return rc;
    }
    return MP_OK;
}

int mp_plan_entry_info(const mp_plan* plan, int index, int64_t info[12]) {
    if (!plan || !info) return MP_ERR_NULL;
    if (index < 0 || (size_t)index >= plan->entries.size()) return MP_ERR_SHAPE;
    const mp_plan::Entry& e = plan->entries[index];
    for (int i = 0; i < 12; ++i) info[i] = 0;
    info[0] = e.kind;
    if (e.kind == 0 && e.conv.pointwise) {
        info[1] = 1; info[2] = 1; info[3] = kPointwise; info[4] = e.conv.pw.grid; info[5] = (int64_t)e.conv.pw.lds_bytes;
        info[6] = e.conv.pw.p.Cout; info[7] = 64; info[8] = 64; info[9] = e.conv.pw.cbw; info[10] = e.conv.pw.p.tiles_per_wg;
        info[11] = e.conv.pw.kq;
    } else if (e.kind == 0 && e.conv.gemm) {
        info[1] = e.conv.ks; info[2] = e.conv.gm.stride; info[3] = kGemm; info[11] = e.conv.gm.gather ? 1 : 0; info[4] = (int64_t)e.conv.gm.grid * e.conv.gm.phases; info[5] = (int64_t)e.conv.gm.lds_bytes;
        info[6] = 64 * e.conv.gm.mi; info[7] = 64 * e.conv.gm.ni; info[8] = 16; info[9] = e.conv.gm.phases; info[10] = 1;
    } else if (e.kind == 0 && e.conv.small_k) {
        info[1] = e.conv.ks; info[2] = e.conv.stride; info[3] = e.conv.variant; info[4] = e.conv.sm.grid; info[5] = (int64_t)e.conv.sm.lds_bytes;
        info[6] = 16; info[7] = 16 * e.conv.sm.p.pt; info[8] = e.conv.sm.p.Cin_pad4; info[9] = 1; info[10] = e.conv.sm.p.rows;
    } else if (e.kind == 0) {
        int ct, pt;
        variant_dims(e.conv.variant, ct, pt);
        info[1] = e.conv.ks; info[2] = e.conv.stride; info[3] = e.conv.variant; info[4] = e.conv.p.total_blocks;
        info[5] = (int64_t)e.conv.lds_bytes; info[6] = ct; info[7] = pt; info[8] = e.conv.p.CK; info[9] = e.conv.p.G;
        info[10] = e.conv.p.R;
        info[11] = variant_light(e.conv.variant) ? 1 : 0;
    } else if (e.kind == 3) {
        int ct, pt;
        f16_variant_dims(e.conv16.variant, ct, pt);
        info[1] = e.conv16.ks; info[2] = e.conv16.stride; info[3] = e.conv16.variant; info[4] = e.conv16.p.total_blocks;
        info[5] = (int64_t)e.conv16.lds_bytes; info[6] = ct; info[7] = pt; info[8] = e.conv16.p.PK * 8; info[9] = e.conv16.p.G;
        info[10] = e.conv16.p.R;
        info[11] = f16_variant_light(e.conv16.variant) ? 1 : 0;
    } else if (e.kind == 9) {
        info[1] = 3; info[2] = 1; info[3] = 9 /* the tuner's index of the Winograd form */; info[4] = e.wino.p.total_blocks;
        info[5] = (int64_t)e.wino.lds_bytes; info[6] = 32 * e.wino.teams; info[7] = e.wino.p.M * 4; info[8] = 8; info[9] = 1; info[10] = e.wino.p.R;
        info[11] = e.wino.ni;
    } else if (e.kind == 12) {
        info[1] = 3; info[2] = 2; info[3] = 0; info[4] = e.stem32.p.total_blocks;
        info[5] = (int64_t)e.stem32.lds_bytes; info[6] = 64; info[7] = 8 * e.stem32.p.Wo; info[8] = 3; info[9] = 1; info[10] = 8;
    } else if (e.kind == 11) {
        info[1] = 3; info[2] = 2; info[3] = 0; info[4] = e.stem16.p.total_blocks;
        info[5] = (int64_t)e.stem16.lds_bytes; info[6] = 64; info[7] = 8 * e.stem16.p.Wo; info[8] = 3; info[9] = 1; info[10] = 8;
    } else if (e.kind == 13) {
        info[1] = 1; info[2] = 1; info[3] = (e.pwchain32.ds ? 1 : 0) + (e.pwchain32.red ? 0 : 2); info[4] = e.pwchain32.grid;
        info[5] = (int64_t)e.pwchain32.lds_bytes; info[6] = 256; info[7] = e.pwchain32.form == 2 ? 32 : 64; info[8] = 64; info[9] = 1; info[10] = 0;
        info[11] = e.pwchain32.form;  // 4 / 8 waves per workgroup on 64-pixel tiles, 2 = four waves on 32-pixel tiles, two workgroups per CU
    } else if (e.kind == 10) {
        info[1] = 1; info[2] = 1; info[3] = e.pwchain.dual ? 1 : e.pwchain.ds ? 2 : 0; info[4] = e.pwchain.p.total_blocks;
        info[5] = (int64_t)e.pwchain.lds_bytes; info[6] = e.pwchain.ce; info[7] = 64; info[8] = e.pwchain.cm; info[9] = 1; info[10] = 0;
    } else if (e.kind == 8) {
        info[1] = 3; info[2] = 1; info[3] = e.block16.small; info[4] = e.block16.p.total_blocks;
        const int blk_c = e.block16.small == 4 ? 64 : e.block16.small == 5 ? 128 : 32;  // cout tile / cin chunk = the block's width
        info[5] = (int64_t)e.block16.lds_bytes; info[6] = blk_c; info[7] = e.block16.p.M2; info[8] = blk_c; info[9] = 1;
        info[10] = e.block16.p.R;
    }
    return MP_OK;
}

int mp_plan_run(const mp_plan* plan, mp_stream_t stream) {
    if (!plan) return MP_ERR_NULL;
    if (plan->max_lane > 0) return run_multi_lane(plan, as_stream(stream));
    return mp_plan_run_range(plan, 0, (int)plan->entries.size(), stream);
}

int mp_plan_set_lane(mp_plan* plan, int lane) {
    if (!plan) return MP_ERR_NULL;
    if (lane < 0 || lane >= kPlanLanes) return MP_ERR_SHAPE;
    plan->cur_lane = lane;
    if (lane > plan->max_lane) plan->max_lane = lane;
    return MP_OK;
}

int mp_plan_add_barrier(mp_plan* plan) {
    if (!plan) return MP_ERR_NULL;
    mp_plan::Entry e{};
    e.kind = 7;
    e.lane = 0;
    plan->entries.push_back(e);
    return MP_OK;
}

}  // extern "C"
