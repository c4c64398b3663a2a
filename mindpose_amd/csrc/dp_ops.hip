// Data-parallel step tail (SURVEY.md 8a row a17, 8b "mp_allreduce_grads"): the gradient mean over ranks on RCCL, the dynamic
// loss-scale overflow check and the AdamWeightDecay update as streaming passes over the flat fp32 gradient arena.
//
// Reference: tools/train.py:43-49 (data_parallel, gradients_mean=True: MindSpore all-reduces every gradient each step),
// tools/train.py:170-181 (DynamicLossScaleManager: overflow check, skip + halve), optim/optim_factory.py:69-72 (AdamWeightDecay).
//
// RCCL is bound at run time (the copy the process already holds - PyTorch ships its own librccl - found with RTLD_NOLOAD; the
// system library only when none is loaded): the library has no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>

#include "common.h"

namespace mp {
struct Id128 {  // ncclUniqueId: 128 opaque bytes passed BY VALUE to ncclCommInitRank
    char internal[128];
};
namespace {

// ---- overflow check: one read of the arena, flag |= any(!isfinite) ----------------------------------------------------------
__global__ __launch_bounds__(256) void finite_check_kernel(const float4* __restrict__ g4, const float* __restrict__ g, size_t n4,
                                                           size_t n, int* __restrict__ flag) {
    // a value is non-finite iff its exponent field is all ones
    unsigned bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        const unsigned e = 0x7f800000u;
        bad |= (unsigned)((__float_as_uint(v.x) & e) == e) | (unsigned)((__float_as_uint(v.y) & e) == e) |
               (unsigned)((__float_as_uint(v.z) & e) == e) | (unsigned)((__float_as_uint(v.w) & e) == e);
    }
    if (blockIdx.x == 0) {
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) bad |= (unsigned)((__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u);
    }
    if (__ballot(bad != 0) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// ---- AdamWeightDecay with the gradient scale folded in (1 / (loss_scale * world)) and a device-side skip flag ----------------
__global__ __launch_bounds__(256) void adamw_scaled_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                           float wd, float grad_scale, const int* __restrict__ skip) {
    if (skip != nullptr && *skip != 0) return;  // overflow step: parameters and moments untouched (uniform over the grid)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        float upd = mi / (sqrtf(vi) + eps);
        upd += wd * p[i];
        p[i] = p[i] - lr * upd;
    }
}

// ---- RCCL, bound lazily ------------------------------------------------------------------------------------------------------
struct Rccl {
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl t;
        void* h = nullptr;
        if (dlsym(RTLD_DEFAULT, "ncclAllReduce") == nullptr) {
            // PyTorch links its own librccl.so (SONAME librccl.so.1) and Python loads it RTLD_LOCAL, so the symbols are not in the
            // global scope: RTLD_NOLOAD hands out THAT copy by soname.  A second RCCL in the process (two sets of IPC state, two
            // proxy threads per communicator) is never created: only when no copy is loaded at all - a C caller without PyTorch -
            // the system library is loaded.
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
            if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
            if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        }
        void* scope = h ? h : RTLD_DEFAULT;
        t.GetUniqueId = reinterpret_cast<decltype(t.GetUniqueId)>(dlsym(scope, "ncclGetUniqueId"));
        t.CommInitRank = reinterpret_cast<decltype(t.CommInitRank)>(dlsym(scope, "ncclCommInitRank"));
        t.CommDestroy = reinterpret_cast<decltype(t.CommDestroy)>(dlsym(scope, "ncclCommDestroy"));
        t.CommCount = reinterpret_cast<decltype(t.CommCount)>(dlsym(scope, "ncclCommCount"));
        t.AllReduce = reinterpret_cast<decltype(t.AllReduce)>(dlsym(scope, "ncclAllReduce"));
        t.ReduceScatter = reinterpret_cast<decltype(t.ReduceScatter)>(dlsym(scope, "ncclReduceScatter"));
        t.AllGather = reinterpret_cast<decltype(t.AllGather)>(dlsym(scope, "ncclAllGather"));
        t.GroupStart = reinterpret_cast<decltype(t.GroupStart)>(dlsym(scope, "ncclGroupStart"));
        t.GroupEnd = reinterpret_cast<decltype(t.GroupEnd)>(dlsym(scope, "ncclGroupEnd"));
        t.ok = t.GetUniqueId && t.CommInitRank && t.CommDestroy && t.AllReduce && t.ReduceScatter && t.AllGather && t.GroupStart &&
               t.GroupEnd;
        return t;
    }();
    return r;
}
constexpr int kNcclFloat = 7, kNcclSum = 0, kNcclAvg = 4;  // rccl.h: ncclFloat32 = 7, ncclSum = 0, ncclAvg = 4
thread_local int g_last_rccl_error = 0;
}  // namespace
}  // namespace mp

using namespace mp;

extern "C" {

int mp_grad_finite_check(const float* grad, size_t count, int* flag, mp_stream_t stream) {
    if (!grad || !flag) return MP_ERR_NULL;
    if (count == 0) return MP_OK;
    if ((reinterpret_cast<uintptr_t>(grad) & 15) != 0) return MP_ERR_UNSUPPORTED;  // arena slices are 16-byte aligned by the caller
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(finite_check_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(grad), grad, n4, count, flag);
    return check_launch();
}

int mp_adamw_step_scaled(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t count, float lr, float beta1,
                         float beta2, float eps, float weight_decay, float grad_scale, const int* skip_flag, mp_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) return MP_ERR_NULL;
    if (count == 0) return MP_OK;
    size_t blocks = (count + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(adamw_scaled_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq,
                       count, lr, beta1, beta2, eps, weight_decay, grad_scale, skip_flag);
    return check_launch();
}

int mp_comm_available(void) { return rccl().ok ? 1 : 0; }

int mp_comm_last_error(void) { return g_last_rccl_error; }

int mp_comm_get_unique_id(void* id128) {
    if (!id128) return MP_ERR_NULL;
    if (!rccl().ok) return MP_ERR_UNSUPPORTED;
    const int rc = rccl().GetUniqueId(id128);
    if (rc != 0) {
        g_last_rccl_error = rc;
        return MP_ERR_HIP;
    }
    return MP_OK;
}

int mp_comm_init_rank(void** comm_out, int nranks, const void* id128, int rank) {
    if (!comm_out || !id128) return MP_ERR_NULL;
    if (nranks <= 0 || rank < 0 || rank >= nranks) return MP_ERR_SHAPE;
    if (!rccl().ok) return MP_ERR_UNSUPPORTED;
    Id128 id;
    __builtin_memcpy(&id, id128, sizeof(id));
    void* comm = nullptr;
    const int rc = rccl().CommInitRank(&comm, nranks, id, rank);
    if (rc != 0) {
        g_last_rccl_error = rc;
        return MP_ERR_HIP;
    }
    *comm_out = comm;
    return MP_OK;
}

int mp_comm_destroy(void* comm) {
    if (!comm) return MP_ERR_NULL;
    if (!rccl().ok) return MP_ERR_UNSUPPORTED;
    const int rc = rccl().CommDestroy(comm);
    if (rc != 0) {
        g_last_rccl_error = rc;
        return MP_ERR_HIP;
    }
    return MP_OK;
}

int mp_comm_count(void* comm) {
    if (!comm || !rccl().ok || !rccl().CommCount) return -1;
    int n = -1;
    const int rc = rccl().CommCount(comm, &n);
    if (rc != 0) {
        g_last_rccl_error = rc;
        return -1;
    }
    return n;
}

int mp_allreduce_grads(void* comm, float* arena, size_t count, int average, mp_stream_t stream) {
    if (!comm || !arena) return MP_ERR_NULL;
    if (count == 0) return MP_OK;
    if (!rccl().ok) return MP_ERR_UNSUPPORTED;
    const int rc = rccl().AllReduce(arena, arena, count, kNcclFloat, average ? kNcclAvg : kNcclSum, comm, as_stream(stream));
    if (rc != 0) {
        g_last_rccl_error = rc;
        return MP_ERR_HIP;
    }
    return MP_OK;
}

int mp_reduce_scatter_allgather_grads(void* comm, float* arena, size_t count, int nranks, int rank, int average, mp_stream_t stream) {
    if (!comm || !arena) return MP_ERR_NULL;
    if (nranks <= 0 || rank < 0 || rank >= nranks) return MP_ERR_SHAPE;
    if (count == 0) return MP_OK;
    if (count % (size_t)nranks != 0) return MP_ERR_SHAPE;  // the caller pads the arena to a multiple of the world size
    if (!rccl().ok) return MP_ERR_UNSUPPORTED;
    const size_t shard = count / (size_t)nranks;
    hipStream_t s = as_stream(stream);
    int rc = rccl().ReduceScatter(arena, arena + (size_t)rank * shard, shard, kNcclFloat, average ? kNcclAvg : kNcclSum, comm, s);
    if (rc == 0) rc = rccl().AllGather(arena + (size_t)rank * shard, arena, shard, kNcclFloat, comm, s);
    if (rc != 0) {
        g_last_rccl_error = rc;
        return MP_ERR_HIP;
    }
    return MP_OK;
}

}  // extern "C"
