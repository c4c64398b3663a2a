// Shared helpers of libmindpose_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mindpose_hip.h"

namespace mp {

extern thread_local int g_last_hip_error;

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

}  // namespace mp
