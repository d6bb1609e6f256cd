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


// internals shared between translation units (conv3d.hip, conv3d_up.hip, sn.hip)
int v2ce_pack_weights_f16x2_absmax_only(const float *w, int Cout, int Cin, int k3, const float *sigma, void *w_f16x2, v2ce_stream_t stream);
int v2ce_pack_weights_f16x2_pack_only(const float *w, int Cout, int Cin, int k3, const float *sigma, void *w_f16x2, v2ce_stream_t stream);
// the folded region of a v2ce_pack_weights_f16x2_up buffer (conv3d_up.hip): max |folded sums| into tail[0]; the pack itself
int v2ce_up_fold_absmax(const float *w, int Cout, int Cin, int C0, const float *sigma, void *w_up, hipStream_t st);
int v2ce_up_fold_pack(const float *w, int Cout, int Cin, int C0, const float *sigma, void *w_up, hipStream_t st);
// the same for up to eight layers in one launch; pass 0 = the maxima, pass 1 = the packs
int v2ce_up_fold_batch(const float *const *w, const int *Cout, const int *Cin, const int *C0, const float *const *sigma, void *const *w_up,
                       int n, int pass, hipStream_t st);

// the Winograd planes (conv3d_wt.hip) of up to 16 layers in one launch per pass: pass 0 raises tail[0] (zeroed by the caller) to
// max |G|, pass 1 derives the pre-scale from it and writes the planes; sigma[l]: device scalar or null
// cin_total / ci0 (may be null: the whole tensor): layer l packs input channels [ci0[l], ci0[l] + cin[l]) of a [rows][cin_total[l]][27] tensor
int v2ce_wt_pack_batch(const float *const *w, const float *const *sigma, void *const *packed, const int *rows, const int *cin, int n, int pass,
                       hipStream_t st, const int *cin_total = nullptr, const int *ci0 = nullptr);

}  // namespace v2ce
