// common.h -- shared helpers of libv2ce_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/v2ce_hip.h"

namespace v2ce {

void set_error(const char *fmt, ...);
void clear_error();

inline hipStream_t as_stream(v2ce_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define V2CE_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            v2ce::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                    \
            return V2CE_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define V2CE_REQUIRE(cond, code, ...)        \
    do {                                     \
        if (!(cond)) {                       \
            v2ce::set_error(__VA_ARGS__);    \
            return (code);                   \
        }                                    \
    } while (0)

constexpr int kWave = 64;  // gfx950 wavefront

// power of two s with  amax * s  in [2^14, 2^15): the hi halves use the top of the fp16 range (max
// 65504) and the lo halves stay normal for every element within 2^-15 of the maximum
__host__ __device__ __forceinline__ float pow2_prescale(float amax) {
    if (!(amax > 0.0f)) return 1.0f;
    int e;
    frexpf(amax, &e);                       // amax = m * 2^e, m in [0.5, 1)
    int k = 15 - e;
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    return ldexpf(1.0f, k);
}


}  // namespace v2ce
