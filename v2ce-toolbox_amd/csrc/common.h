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

}  // namespace v2ce
