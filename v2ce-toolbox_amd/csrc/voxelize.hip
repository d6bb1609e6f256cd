// voxelize.hip -- events -> discretised event volume on gfx950 (the inverse of LDATI).
//
// Replaces gen_discretized_event_volume of
// /root/reference/train/scripts/utils/events_utils.py:118-175 (calc_floor_ceil_delta :118-126,
// create_update :128-145): the time axis of the event set is rescaled to [0, bins-1] over its own
// [t_min, t_max], every event is split linearly between its floor and its ceil bin, positive
// polarity goes to planes [0, bins), negative (polarity == 0) to [bins, 2*bins).
//
// Arithmetic follows the reference's CPU torch evaluation step by step (this file is built with
// -ffp-contract=off): scale = f32(1 / f32(t_max - t_min)) * f32(bins - 1)   (int / tensor is
// reciprocal * int), ts = clamp(f32(t - t_min) * scale, 0, bins-1), floor(ts + 1e-8f),
// ceil(ts - 1e-8f), weights (floor(ts)+1) - ts and ts - floor(ts + 1e-8f).  The only difference is
// the accumulation ORDER (float atomics instead of a sequential put_): results agree to f32
// summation error.  HBM-bound: 13 B read per event, two f32 atomics per event.
#include "common.h"

namespace v2ce {
namespace {

__global__ __launch_bounds__(256) void time_range_kernel(const int64_t *__restrict__ ts, long long n,
                                                         long long *range) {
    long long lo = 0x7fffffffffffffffll, hi = -0x7fffffffffffffffll - 1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long t = ts[i];
        lo = t < lo ? t : lo;
        hi = t > hi ? t : hi;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(range, lo);
        atomicMax(range + 1, hi);
    }
}

__global__ __launch_bounds__(256) void voxelize_kernel(const int64_t *__restrict__ ts, const int16_t *__restrict__ x,
                                                       const int16_t *__restrict__ y, const int8_t *__restrict__ p,
                                                       long long n, const long long *__restrict__ range, int bins,
                                                       int H, int W, float *__restrict__ vol) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long t_min = range[0], t_max = range[1];
    const float scale = (1.0f / (float)(t_max - t_min)) * (float)(bins - 1);       // events_utils.py:159
    float t = (float)(ts[i] - t_min) * scale;
    t = fminf(fmaxf(t, 0.0f), (float)(bins - 1));                                  // :160
    const float fl = floorf(t + 1e-8f), ce = ceilf(t - 1e-8f);                     // :119-120
    const float ce_fake = floorf(t) + 1.0f;                                        // :121
    const float d_ce = t - fl, d_fl = ce_fake - t;                                 // :123-124
    const int xi = x[i], yi = y[i];
    if (xi < 0 || xi >= W || yi < 0 || yi >= H) return;        // the host wrapper rejects these (:129-130)
    const long long plane = p[i] == 0 ? bins : 0;                                  // :155, :134-136
    const long long pix = (long long)W * yi + xi;
    atomicAdd(vol + (long long)H * W * ((long long)fl + plane) + pix, d_fl);       // :164-168
    atomicAdd(vol + (long long)H * W * ((long long)ce + plane) + pix, d_ce);       // :170-173
}

}  // namespace
}  // namespace v2ce

using namespace v2ce;

extern "C" int v2ce_voxelize_events(const int64_t *ts, const int16_t *x, const int16_t *y, const int8_t *p,
                                    int64_t n, int bins, int H, int W, float *volume, int64_t *t_range,
                                    v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(ts && x && y && p && volume && t_range, V2CE_ERR_BAD_ARG, "v2ce_voxelize_events: null pointer");
    V2CE_REQUIRE(n > 0 && bins >= 2 && H > 0 && W > 0, V2CE_ERR_BAD_ARG,
                 "v2ce_voxelize_events: needs n > 0, bins >= 2, H, W > 0");
    hipStream_t st = as_stream(stream);
    const long long init[2] = {0x7fffffffffffffffll, -0x7fffffffffffffffll - 1};
    V2CE_HIP_CHECK(hipMemcpyAsync(t_range, init, sizeof(init), hipMemcpyHostToDevice, st));
    V2CE_HIP_CHECK(hipMemsetAsync(volume, 0, (size_t)2 * bins * H * W * sizeof(float), st));
    const long long nb = (n + 255) / 256;
    hipLaunchKernelGGL(time_range_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, st, ts, (long long)n,
                       reinterpret_cast<long long *>(t_range));
    hipLaunchKernelGGL(voxelize_kernel, dim3((unsigned)nb), dim3(256), 0, st, ts, x, y, p, (long long)n,
                       reinterpret_cast<const long long *>(t_range), bins, H, W, volume);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}
