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

// The same with the resize of v2ce.py:57-58 in front: cv2.resize(img, (out_w, out_h)), INTER_LINEAR, restated from
// OpenCV's scalar algorithm exactly like the host path glue._resize_bilinear (resize.cpp, resizeGeneric_):
//   scale = 1. / ((double)out / in);  f = (float)((d + 0.5) * scale - 0.5);  s = floor(f);  f -= s  (in float)
//   x axis: s < 0 -> (f, s) = (0, 0);  s >= W - 1 -> S[W - 1] copied;   y axis: rows clipped, f kept
//   horizontal pass S[s] * (1 - f) + S[s + 1] * f on both source rows, then r0 * (1 - g) + r1 * g
// (separately rounded f32 operations); an exact 2 x 2 decimation is OpenCV's INTER_AREA fast path.
__device__ __forceinline__ void linear_tap(int d, int n_in, int n_out, bool zero_at_edges, int &s0, int &s1, float &f, bool &copy) {
    const double scale = 1.0 / ((double)n_out / (double)n_in);
    f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floor((double)f);
    f = f - (float)s;
    copy = false;
    if (zero_at_edges) {
        if (s < 0) { f = 0.0f; s = 0; }
        if (s >= n_in - 1) { f = 0.0f; s = n_in - 1; copy = true; }
        s0 = s;
        s1 = s + 1 < n_in ? s + 1 : n_in - 1;
    } else {
        s0 = s < 0 ? 0 : (s > n_in - 1 ? n_in - 1 : s);
        s1 = s + 1 < 0 ? 0 : (s + 1 > n_in - 1 ? n_in - 1 : s + 1);
    }
}

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
    const unsigned char *img = fr + (i + c) * (long long)H * W;
    auto px = [&](int y, int x) { return (float)img[(long long)y * W + x] / 255.0f; };
    float x;
    if (W == 2 * ow && H == 2 * oh) {
        const float a = px(2 * oy, 2 * ox), b = px(2 * oy, 2 * ox + 1), cc = px(2 * oy + 1, 2 * ox), d = px(2 * oy + 1, 2 * ox + 1);
        x = ((a + b) + (cc + d)) * 0.25f;
    } else {
        int x0, x1, y0, y1;
        float fx, fy;
        bool cpx, cpy;
        linear_tap(ox, W, ow, true, x0, x1, fx, cpx);
        linear_tap(oy, H, oh, false, y0, y1, fy, cpy);
        const float top = cpx ? px(y0, x0) : px(y0, x0) * (1.0f - fx) + px(y0, x1) * fx;
        const float bot = cpx ? px(y1, x0) : px(y1, x0) * (1.0f - fx) + px(y1, x1) * fx;
        x = top * (1.0f - fy) + bot * fy;
    }
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
