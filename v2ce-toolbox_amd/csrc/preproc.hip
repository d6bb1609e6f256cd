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

// The same with the resize of v2ce.py:57 in front (cv2.resize(img, (out_w, out_h)), INTER_LINEAR
// convention: half-pixel centres, edge clamp), restated exactly like the host path glue._resize_bilinear:
// source coordinate (o + 0.5) * (in / out) - 0.5 in f64, weights cast to f32, horizontal blend of the
// two source rows first, then the vertical blend -- separate f32 operations.
__global__ __launch_bounds__(256) void preprocess_pairs_resize_kernel(const unsigned char *__restrict__ fr, int H, int W,
                                                                      int oh, int ow, long long total, float mean,
                                                                      float stdv, float *__restrict__ units) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long long ohw = (long long)oh * ow;
    const int p = (int)(e % ohw);
    const long long ic = e / ohw;
    const long long i = ic >> 1, c = ic & 1;
    const int oy = p / ow, ox = p - oy * ow;
    const double ys = ((double)oy + 0.5) * ((double)H / (double)oh) - 0.5;
    const double xs = ((double)ox + 0.5) * ((double)W / (double)ow) - 0.5;
    const double y0d = floor(ys), x0d = floor(xs);
    const float fy = (float)(ys - y0d), fx = (float)(xs - x0d);
    const long long y0 = (long long)y0d, x0 = (long long)x0d;
    const int y0c = (int)(y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0)), y1c = (int)(y0 + 1 < 0 ? 0 : (y0 + 1 > H - 1 ? H - 1 : y0 + 1));
    const int x0c = (int)(x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0)), x1c = (int)(x0 + 1 < 0 ? 0 : (x0 + 1 > W - 1 ? W - 1 : x0 + 1));
    const unsigned char *img = fr + (i + c) * (long long)H * W;
    const float a = (float)img[(long long)y0c * W + x0c] / 255.0f, b = (float)img[(long long)y0c * W + x1c] / 255.0f;
    const float cc = (float)img[(long long)y1c * W + x0c] / 255.0f, d = (float)img[(long long)y1c * W + x1c] / 255.0f;
    const float top = a * (1.0f - fx) + b * fx;
    const float bot = cc * (1.0f - fx) + d * fx;
    const float x = top * (1.0f - fy) + bot * fy;
    units[e] = (x - mean) / stdv;
}

}  // namespace
}  // namespace v2ce

using namespace v2ce;

extern "C" int v2ce_preprocess_pairs_resize(const uint8_t *frames, int N, int H, int W, int out_h, int out_w, float mean,
                                            float stdv, float *units, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(frames && units, V2CE_ERR_BAD_ARG, "v2ce_preprocess_pairs_resize: null pointer");
    V2CE_REQUIRE(N >= 2 && H > 0 && W > 0 && out_h > 0 && out_w > 0, V2CE_ERR_BAD_ARG,
                 "v2ce_preprocess_pairs_resize: need >= 2 frames and positive sizes");
    const long long total = (long long)(N - 1) * 2 * out_h * out_w;
    const long long blocks = (total + 255) / 256;
    V2CE_REQUIRE(blocks < (1ll << 31), V2CE_ERR_UNSUPPORTED, "v2ce_preprocess_pairs_resize: too large");
    hipLaunchKernelGGL(preprocess_pairs_resize_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), frames, H, W,
                       out_h, out_w, total, mean, stdv, units);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

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
