// sampler.hip -- the two ablation samplers of the reference's stage-2 study on gfx950.
//
// Replaces /root/reference/train/scripts/stage2/sample_methods/random_even_sample.py:115-169
// (sample_voxel_baseline: floor(y) events at uniform-random or evenly spaced times + one
// Bernoulli(frac(y)) event, all TEN bins) and pure_slope_sample.py:57-149 (slope-distributed times
// from the un-relocated voxel values; bin 9 folded into bin 8).  Both end with
// np.sort(records, order='timestamp') over a whole frame, which orders equal timestamps by the
// remaining fields: the output order is the lexicographic order of (timestamp, x, y, polarity).
//
// That total order is the whole design: every event becomes ONE 64-bit key
//     frame | timestamp - base | x | y | polarity          (field widths from the problem size)
// written in any order (one atomic per wave reserves the wave's slots), a device-wide LSD radix
// sort over exactly the significant bits (rocPRIM, header-only) orders all frames at once, and a
// last kernel unpacks the keys into the SoA event arrays.  No per-frame segmentation, no
// dependence of the result on the write order.  HBM-bound like LDATI; these are ablation
// baselines, not the CLI's path, so the library sort is used as it is.
//
// Arithmetic: built with -ffp-contract=off and correctly rounded '/' and sqrt; every f32
// operation of the reference is one operation here, in its order (oracle/sample_methods.py is the
// line-by-line CPU restatement, pinned by the reference's own outputs).
#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace v2ce {
namespace {

constexpr int kC = 10;   // time bins of a voxel grid: the samplers use all ten (the reference's C)

struct SamplerParams {
    const float *vox;   // [B][2][10][H][W]
    int B, H, W, HW;
    int mode, rng_mode;
    float DELTA, FPS, VS, VS2, INV;   // f32(1/(fps*10)), f32(fps), f32(vs), f32(vs*vs), f32(1/vs)
    float off[kC];                    // f32(arange)[c] + f32(t0)
    const float *u_int, *u_dec, *u_bern;
    const float *pooled;              // pure-slope: voxel values pooled over the pixel neighbourhood (slope only), or null
    int replay_M;
    unsigned long long seed;
    int frame_base;
    long long ts_base;
    int tb, xb, yb;                   // key field widths
    unsigned long long *keys;
    unsigned long long *cursor;       // [1]
    unsigned long long *counts;       // [B] (count kernel)
    int *max_int;                     // [1]
    int *status;                      // [1]
};

__device__ __forceinline__ void philox4(unsigned long long seed, unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                        unsigned (&out)[4]) {
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (unsigned)p1; c2 = n2; c3 = (unsigned)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// draw j of (kind, frame b, plane pi, bin c, pixel): replayed tensor or Philox (same counter layout as LDATI's,
// the stream id 32*kind + 10*pi + c in the third word)
__device__ __forceinline__ float draw(const SamplerParams &P, int kind, int b, int pi, int c, int pix, int j) {
    if (P.rng_mode == V2CE_RNG_REPLAY) {
        const long long v = ((long long)(b * 2 + pi) * kC + c) * P.HW + pix;
        return kind == 0 ? P.u_int[v * P.replay_M + j] : kind == 1 ? P.u_dec[v] : P.u_bern[v];
    }
    unsigned o[4];
    philox4(P.seed, (unsigned)pix, (unsigned)j >> 2, (unsigned)(32 * kind + 10 * pi + c), (unsigned)(P.frame_base + b), o);
    const unsigned sel = (unsigned)j & 3u;
    const unsigned w = sel == 0 ? o[0] : sel == 1 ? o[1] : sel == 2 ? o[2] : o[3];
    return (float)(w >> 8) * (1.0f / 16777216.0f);
}

// integer part, fractional part of bin c (after the pure-slope fold of bin 9 into bin 8)
__device__ __forceinline__ void split_voxel(const SamplerParams &P, const float (&y)[kC], int c, float &ip, float &dp) {
    float v = y[c];
    if (P.mode == V2CE_SAMPLER_PURE_SLOPE) {
        if (c == 8) v = y[8] + y[9];                        // pure_slope_sample.py:92
        if (c == 9) v = 0.0f;                               // :93
        ip = (float)(int)floorf(v);                         // :95 floor().int()
    } else {
        ip = floorf(v);                                     // random_even_sample.py:125
    }
    dp = v - ip;                                            // :126 / pure_slope :96
}

__device__ __forceinline__ long long to_us(float t, float off) {
    t = t + off;
    t = t * 1e6f;
    return (long long)t;
}

struct Slope {
    float k, bb;
};

__device__ __forceinline__ Slope slope_of(const SamplerParams &P, const float (&y)[kC], int c) {
    // reflect padding (pure_slope_sample.py:24): the neighbours of bin 0 are (y1, y1), of bin 9 (y8, y8)
    const float l = c == 0 ? y[1] : y[c - 1], r = c == kC - 1 ? y[kC - 2] : y[c + 1];
    const float sxy = r - l;                                // :38
    const float k0 = (3.0f * sxy) / 6.0f;                   // :52
    Slope s;
    s.k = (k0 / P.VS2) / (y[c] + 1e-8f);                    // :88
    s.bb = P.INV - (P.VS * s.k) / 2.0f;                     // :91
    return s;
}

__device__ __forceinline__ float slope_time(const SamplerParams &P, const Slope &s, float u) {
    if (s.k == 0.0f) return (u / P.FPS) / 10.0f;                        // :106 / :131
    const float q = s.bb * s.bb + (2.0f * s.k) * u;
    return (-s.bb + __builtin_sqrtf(q)) / s.k;                          // :105 / :130
}

// events of one (frame, plane, pixel) column: floor parts and Bernoulli events of the ten bins
__device__ __forceinline__ int column_events(const SamplerParams &P, const float (&y)[kC], int b, int pi, int pix,
                                             int (&n)[kC], unsigned &bern_mask, int &max_n) {
    int total = 0;
    bern_mask = 0;
    max_n = 0;
#pragma unroll
    for (int c = 0; c < kC; ++c) {
        float ip, dp;
        split_voxel(P, y, c, ip, dp);
        n[c] = ip > 0.0f ? (int)ip : 0;
        max_n = n[c] > max_n ? n[c] : max_n;
        const bool hit = draw(P, 2, b, pi, c, pix, 0) < dp;             // torch.bernoulli(frac): u < p
        bern_mask |= hit ? 1u << c : 0u;
        total += n[c] + (hit ? 1 : 0);
    }
    return total;
}

__global__ __launch_bounds__(256) void sampler_count_kernel(SamplerParams P) {
    const int pix = blockIdx.x * 256 + threadIdx.x, pi = blockIdx.y, b = blockIdx.z;
    int total = 0, max_n = 0;
    if (pix < P.HW) {
        float y[kC];
        const float *src = P.vox + ((long long)(b * 2 + pi) * kC) * P.HW + pix;
#pragma unroll
        for (int c = 0; c < kC; ++c) y[c] = src[(long long)c * P.HW];
        int n[kC];
        unsigned mask;
        total = column_events(P, y, b, pi, pix, n, mask, max_n);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        total += __shfl_xor(total, o);
        const int m2 = __shfl_xor(max_n, o);
        max_n = m2 > max_n ? m2 : max_n;
    }
    if ((threadIdx.x & 63) == 0) {
        if (total) atomicAdd(P.counts + b, (unsigned long long)total);
        if (max_n) atomicMax(P.max_int, max_n);
    }
}

__global__ __launch_bounds__(256) void sampler_emit_kernel(SamplerParams P) {
    const int pix = blockIdx.x * 256 + threadIdx.x, pi = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63;
    float y[kC], yp[kC];
    int n[kC];
    unsigned mask = 0;
    int total = 0, max_n = 0;
    if (pix < P.HW) {
        const float *src = P.vox + ((long long)(b * 2 + pi) * kC) * P.HW + pix;
#pragma unroll
        for (int c = 0; c < kC; ++c) y[c] = src[(long long)c * P.HW];
        total = column_events(P, y, b, pi, pix, n, mask, max_n);
        if (P.pooled) {
            const float *ps = P.pooled + ((long long)(b * 2 + pi) * kC) * P.HW + pix;
#pragma unroll
            for (int c = 0; c < kC; ++c) yp[c] = ps[(long long)c * P.HW];
        }
    }
    // the wave's slots: inclusive scan, one atomic per wave
    int incl = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        incl += lane >= o ? v : 0;
    }
    const int wave_total = __shfl(incl, 63);
    if (wave_total == 0) return;
    unsigned long long base = 0;
    if (lane == 63) base = atomicAdd(P.cursor, (unsigned long long)wave_total);
    base = __shfl(base, 63);
    if (total == 0) return;
    unsigned long long *dst = P.keys + base + (unsigned long long)(incl - total);

    const int h = pix / P.W, w = pix - h * P.W;
    const unsigned long long pol = pi == 0 ? 1ull : 0ull;              // P index 0 = positive events (polarity 1)
    const unsigned long long lowbits = (((unsigned long long)w << P.yb) | (unsigned long long)h) << 1 | pol;
    const int tshift = P.xb + P.yb + 1;
    const unsigned long long fbits = (unsigned long long)b << (P.tb + tshift);
    bool bad = false;
    auto put = [&](long long ts) {
        long long rel = ts - P.ts_base;
        if (rel < 0 || rel >= (1ll << P.tb)) {
            bad = true;
            rel = rel < 0 ? 0 : (1ll << P.tb) - 1;
        }
        *dst++ = fbits | ((unsigned long long)rel << tshift) | lowbits;
    };
#pragma unroll
    for (int c = 0; c < kC; ++c) {
        const int nc = n[c];
        const bool hit = (mask >> c) & 1u;
        if (nc == 0 && !hit) continue;
        const float off = P.off[c];
        if (P.mode == V2CE_SAMPLER_PURE_SLOPE) {
            const Slope s = slope_of(P, P.pooled ? yp : y, c);       // pure_slope_sample.py:88-91: from y_pooled
            for (int j = 0; j < nc; ++j) put(to_us(slope_time(P, s, draw(P, 0, b, pi, c, pix, j)), off));
            if (hit) put(to_us(slope_time(P, s, draw(P, 1, b, pi, c, pix, 0)), off));
        } else if (P.mode == V2CE_SAMPLER_EVEN) {
            const float ip = floorf(y[c]);
            for (int j = 0; j < nc; ++j) put(to_us(((float)j / (ip + 1.0f)) * P.DELTA, off));      // :138-140
            if (hit) put(to_us((ip / (ip + 1.0f)) * P.DELTA, off));                                 // :152-153
        } else {
            for (int j = 0; j < nc; ++j) put(to_us(draw(P, 0, b, pi, c, pix, j) * P.DELTA, off));   // :134
            if (hit) put(to_us(draw(P, 1, b, pi, c, pix, 0) * P.DELTA, off));                       // :149
        }
    }
    if (bad) atomicOr(P.status, 1);
}

// pure_slope_sample.py:79-85: y pooled over the k x k pixel neighbourhood of every (frame, polarity, bin) plane, zero
// padding.  'weighted' (F.conv2d with [[1,2,1],[2,4,2],[1,2,1]]/16): taps in row-major order, one multiply and one
// add per tap; 'avg' (nn.AvgPool2d(k, 1, k//2), count_include_pad): window sum in row-major order / k^2.  Sums of
// arbitrary f32 values: the last bit depends on the summation order (the reference's is its conv backend's), so
// what is derived from them is compared at 1 us, not bit for bit (oracle/sample_methods.py pool_voxels: same order).
__global__ __launch_bounds__(256) void sampler_pool_kernel(const float *__restrict__ vox, int H, int W, int weighted, int k,
                                                           float *__restrict__ pooled) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const long long plane = (long long)blockIdx.y * H * W;
    const int h = pix / W, w = pix - h * W, r = weighted ? 1 : k / 2;
    float acc = 0.0f;
    for (int dh = -r; dh <= r; ++dh)
        for (int dw = -r; dw <= r; ++dw) {
            const int hh = h + dh, ww = w + dw;
            const float v = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? vox[plane + (long long)hh * W + ww] : 0.0f;
            if (weighted) acc = acc + ((float)((2 - (dh < 0 ? -dh : dh)) * (2 - (dw < 0 ? -dw : dw))) / 16.0f) * v;
            else acc = acc + v;
        }
    pooled[plane + pix] = weighted ? acc : acc / (float)(k * k);
}

__global__ __launch_bounds__(256) void sampler_unpack_kernel(const unsigned long long *__restrict__ keys, long long n,
                                                             long long ts_base, int tb, int xb, int yb,
                                                             int64_t *__restrict__ ts, int16_t *__restrict__ x,
                                                             int16_t *__restrict__ y, int8_t *__restrict__ p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned long long k = keys[i];
    p[i] = (int8_t)(k & 1ull);
    k >>= 1;
    y[i] = (int16_t)(k & ((1ull << yb) - 1));
    k >>= yb;
    x[i] = (int16_t)(k & ((1ull << xb) - 1));
    k >>= xb;
    ts[i] = (int64_t)(k & ((1ull << tb) - 1)) + ts_base;
}

int bits_for(long long v) {   // bits that hold 0 .. v
    int b = 1;
    while ((v >> b) != 0) ++b;
    return b;
}

// validated launch parameters; returns V2CE_OK or an error code (message already set)
int make_params(const float *vox, int B, int H, int W, const v2ce_sampler_options *o, SamplerParams &P) {
    V2CE_REQUIRE(vox && o, V2CE_ERR_BAD_ARG, "v2ce_sampler: null pointer");
    V2CE_REQUIRE(B > 0 && H > 0 && W > 0 && H <= 32767 && W <= 32767 && (long long)H * W < (1ll << 31), V2CE_ERR_BAD_ARG,
                 "v2ce_sampler: needs B, H, W > 0 and H, W <= 32767");
    V2CE_REQUIRE(o->mode >= V2CE_SAMPLER_RANDOM && o->mode <= V2CE_SAMPLER_PURE_SLOPE, V2CE_ERR_BAD_ARG,
                 "v2ce_sampler: unknown mode");
    V2CE_REQUIRE(o->rng_mode == V2CE_RNG_REPLAY || o->rng_mode == V2CE_RNG_PHILOX, V2CE_ERR_BAD_ARG,
                 "v2ce_sampler: unknown rng_mode");
    V2CE_REQUIRE(o->fps > 0, V2CE_ERR_BAD_ARG, "v2ce_sampler: fps must be positive");
    if (o->rng_mode == V2CE_RNG_REPLAY) {
        V2CE_REQUIRE(o->u_bern, V2CE_ERR_BAD_ARG, "v2ce_sampler: replay mode needs u_bern");
        V2CE_REQUIRE(o->mode == V2CE_SAMPLER_EVEN || (o->u_dec && (o->u_int || o->replay_M == 0)), V2CE_ERR_BAD_ARG,
                     "v2ce_sampler: replay mode needs u_int and u_dec");
        V2CE_REQUIRE(o->replay_M >= 0, V2CE_ERR_BAD_ARG, "v2ce_sampler: replay_M < 0");
    }
    const double fps = o->fps, vs = 1.0 / (fps * kC), step = 1.0 / fps / kC;
    V2CE_REQUIRE((long long)std::ceil((1.0 / fps) / step) == kC, V2CE_ERR_UNSUPPORTED,
                 "v2ce_sampler: arange(0, 1/fps, 1/fps/10) does not have 10 elements (the reference raises as well)");
    P.vox = vox; P.B = B; P.H = H; P.W = W; P.HW = H * W;
    P.mode = o->mode; P.rng_mode = o->rng_mode;
    P.DELTA = (float)vs; P.FPS = (float)fps; P.VS = (float)vs; P.VS2 = (float)(vs * vs); P.INV = (float)(1.0 / vs);
    for (int c = 0; c < kC; ++c) P.off[c] = (float)((double)c * step) + (float)o->t0;
    P.u_int = o->u_int; P.u_dec = o->u_dec; P.u_bern = o->u_bern; P.replay_M = o->replay_M;
    P.pooled = o->mode == V2CE_SAMPLER_PURE_SLOPE ? o->pooled : nullptr;
    P.seed = o->seed; P.frame_base = o->frame_base;
    // key layout: timestamps of a frame lie in [t0, t0 + 1/fps] * 1e6 up to f32 rounding of the sum
    const double t0us = o->t0 * 1e6, margin = 4096.0 + std::fabs(t0us) * 0x1p-18 + std::fabs(t0us + 1e6 / fps) * 0x1p-18;
    P.ts_base = (long long)std::floor(t0us - margin);
    P.tb = bits_for((long long)std::ceil(1e6 / fps + 2 * margin));
    P.xb = bits_for(W - 1); P.yb = bits_for(H - 1);
    V2CE_REQUIRE(P.tb + P.xb + P.yb + 1 + bits_for(B - 1) <= 64, V2CE_ERR_UNSUPPORTED,
                 "v2ce_sampler: (frame, timestamp, x, y, polarity) does not fit a 64-bit key");
    return V2CE_OK;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t sort_temp_bytes(long long n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)n, 0u, 64u,
                             (hipStream_t)0);
    return bytes;
}

}  // namespace
}  // namespace v2ce

using namespace v2ce;

extern "C" int v2ce_sampler_count(const float *vox, int B, int H, int W, const v2ce_sampler_options *options,
                                  int64_t *frame_counts, int32_t *max_int, v2ce_stream_t stream) {
    clear_error();
    SamplerParams P{};
    if (const int rc = make_params(vox, B, H, W, options, P)) return rc;
    V2CE_REQUIRE(frame_counts && max_int, V2CE_ERR_BAD_ARG, "v2ce_sampler_count: null output");
    hipStream_t st = as_stream(stream);
    V2CE_HIP_CHECK(hipMemsetAsync(frame_counts, 0, sizeof(int64_t) * B, st));
    V2CE_HIP_CHECK(hipMemsetAsync(max_int, 0, sizeof(int32_t), st));
    P.counts = reinterpret_cast<unsigned long long *>(frame_counts);
    P.max_int = max_int;
    hipLaunchKernelGGL(sampler_count_kernel, dim3((P.HW + 255) / 256, 2, B), dim3(256), 0, st, P);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

extern "C" int v2ce_sampler_pool(const float *vox, int B, int H, int W, int pooling_type, int pooling_kernel_size,
                                 float *pooled, v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(vox && pooled && B > 0 && H > 0 && W > 0, V2CE_ERR_BAD_ARG, "v2ce_sampler_pool: bad argument");
    V2CE_REQUIRE(pooling_type == V2CE_POOL_AVG || pooling_type == V2CE_POOL_WEIGHTED, V2CE_ERR_BAD_ARG,
                 "v2ce_sampler_pool: pooling_type %d", pooling_type);
    V2CE_REQUIRE(pooling_type != V2CE_POOL_AVG || (pooling_kernel_size >= 1 && pooling_kernel_size <= 15 && (pooling_kernel_size & 1)),
                 V2CE_ERR_UNSUPPORTED, "v2ce_sampler_pool: pooling_kernel_size %d (odd sizes 1..15 keep H x W)", pooling_kernel_size);
    hipLaunchKernelGGL(sampler_pool_kernel, dim3((H * W + 255) / 256, B * 2 * kC), dim3(256), 0, as_stream(stream), vox, H, W,
                       pooling_type == V2CE_POOL_WEIGHTED ? 1 : 0, pooling_kernel_size, pooled);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}

extern "C" size_t v2ce_sampler_workspace_bytes(int64_t total_events) {
    if (total_events <= 0) return 256;
    return 256 + 2 * align256((size_t)total_events * 8) + align256(sort_temp_bytes(total_events));
}

extern "C" int v2ce_sampler_emit(const float *vox, int B, int H, int W, const v2ce_sampler_options *options,
                                 int64_t total_events, int64_t *ts, int16_t *x, int16_t *y, int8_t *p,
                                 void *workspace, size_t workspace_bytes, int32_t *status, v2ce_stream_t stream) {
    clear_error();
    SamplerParams P{};
    if (const int rc = make_params(vox, B, H, W, options, P)) return rc;
    V2CE_REQUIRE(status, V2CE_ERR_BAD_ARG, "v2ce_sampler_emit: null status");
    hipStream_t st = as_stream(stream);
    V2CE_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    if (total_events <= 0) return V2CE_OK;
    V2CE_REQUIRE(ts && x && y && p && workspace, V2CE_ERR_BAD_ARG, "v2ce_sampler_emit: null pointer");
    V2CE_REQUIRE(workspace_bytes >= v2ce_sampler_workspace_bytes(total_events), V2CE_ERR_WORKSPACE,
                 "v2ce_sampler_emit: workspace too small");
    char *ws = static_cast<char *>(workspace);
    const size_t kbytes = align256((size_t)total_events * 8);
    P.cursor = reinterpret_cast<unsigned long long *>(ws);
    P.keys = reinterpret_cast<unsigned long long *>(ws + 256);
    unsigned long long *sorted = reinterpret_cast<unsigned long long *>(ws + 256 + kbytes);
    void *temp = ws + 256 + 2 * kbytes;
    size_t temp_bytes = workspace_bytes - (256 + 2 * kbytes);
    P.status = status;
    V2CE_HIP_CHECK(hipMemsetAsync(P.cursor, 0, 8, st));
    hipLaunchKernelGGL(sampler_emit_kernel, dim3((P.HW + 255) / 256, 2, B), dim3(256), 0, st, P);
    V2CE_HIP_CHECK(hipGetLastError());
    const unsigned end_bit = (unsigned)(P.tb + P.xb + P.yb + 1 + bits_for(B - 1));
    V2CE_HIP_CHECK(rocprim::radix_sort_keys(temp, temp_bytes, P.keys, sorted, (size_t)total_events, 0u, end_bit, st));
    hipLaunchKernelGGL(sampler_unpack_kernel, dim3((unsigned)((total_events + 255) / 256)), dim3(256), 0, st, sorted,
                       (long long)total_events, P.ts_base, P.tb, P.xb, P.yb, ts, x, y, p);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}
