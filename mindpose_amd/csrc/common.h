// Shared helpers of libmindpose_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mindpose_hip.h"

namespace mp {

extern thread_local int g_last_hip_error;
// set around a dispatch by the "would this launch be accepted" queries: the leaf launch functions return MP_OK without launching
extern thread_local bool g_dry_launch;

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return MP_ERR_HIP;
    }
    return MP_OK;
}

inline hipStream_t as_stream(mp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: one of these per kernel instantiation
// (function-local static) remembers on which devices it has been raised, so a process that drives a second device sets it
// there too; fetch_or makes concurrent first launches from several host threads harmless (both set it, once each at worst).
struct AttrOnce {
    unsigned long long mask = 0;
    bool need() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return true;
        const unsigned long long bit = 1ull << (dev & 63);
        return (__atomic_fetch_or(&mask, bit, __ATOMIC_RELAXED) & bit) == 0;
    }
};

// Experiment knobs (MP_* environment variables: tile counts, forced forms - used by tests/ and tools/ to reach code paths small
// problems would not take).  A production process never looks at them: whether they are honoured at all is decided ONCE per
// process by MINDPOSE_EXPERIMENT_KNOBS=1 (tests/conftest.py and the tools set it), so the configure functions - called per conv
// launch by the eager training path - cost no getenv() unless that switch is on.
inline bool knobs_on() {
    static const bool on = [] {
        const char* e = getenv("MINDPOSE_EXPERIMENT_KNOBS");
        return e != nullptr && atoi(e) != 0;
    }();
    return on;
}
inline const char* knob(const char* name) { return knobs_on() ? getenv(name) : nullptr; }

}  // namespace mp
