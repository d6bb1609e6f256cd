// preproc.hip -- frame ingest on the device: uint8 grayscale frames -> normalised frame-pair units.
//
// Replaces /root/reference/v2ce.py:45-64 (image_pre_processing) for frames that already have the
// target height (cv2.resize is then the identity): x = u8 / 255 (f32), pair stacking
// [N-1, 2, H, W] = (frame i, frame i+1), transforms.Normalize(mean, std) = (x - mean) / std.
// Every operation is a separate correctly rounded f32 operation (built with the EXACT flags), so
// the result is bit-identical to the host path.
#include "common.h"

namespace v2ce {
namespace {

__global__ __launch_bounds__(256) void preprocess_pairs_kernel(const unsigned char *__restrict__ fr,
                                                               long long hw, long long total,
                                                               float mean, float stdv,
                                                               float *__restrict__ units) {
    // one thread per output element; units[(i*2 + c)*hw + p] = f(frames[(i + c)*hw + p])
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long long p = e % hw;
    const long long ic = e / hw;
    const long long i = ic >> 1, c = ic & 1;
    const float x = (float)fr[(i + c) * hw + p] / 255.0f;
    units[e] = (x - mean) / stdv;
}

}  // namespace
}  // namespace v2ce

using namespace v2ce;

extern "C" int v2ce_preprocess_pairs(const uint8_t *frames, int N, int H, int W, float mean,
                                     float stdv, float *units, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(frames && units, V2CE_ERR_BAD_ARG, "v2ce_preprocess_pairs: null pointer");
    V2CE_REQUIRE(N >= 2 && H > 0 && W > 0, V2CE_ERR_BAD_ARG, "v2ce_preprocess_pairs: need >= 2 frames");
    const long long hw = (long long)H * W, total = (long long)(N - 1) * 2 * hw;
    const long long blocks = (total + 255) / 256;
    V2CE_REQUIRE(blocks < (1ll << 31), V2CE_ERR_UNSUPPORTED, "v2ce_preprocess_pairs: too large");
    hipLaunchKernelGGL(preprocess_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       frames, hw, total, mean, stdv, units);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}
