// sn.hip -- spectral-norm power iteration for gfx950 (GEMV-class, HBM-bound on W_bar).
//
// Replaces /root/reference/scripts/spectral_norm.py:19-31 (_update_u_v with power_iterations=1):
//     v = l2normalize(W^T u);  u = l2normalize(W v);  sigma = u . (W v)
// with W = W_bar viewed [rows = Cout][cols = Cin*27].  u and v persist across calls (the reference
// mutates them on EVERY forward, eval included), so they are updated in place.  The division of
// the weights by sigma is fused into the weight re-layout (v2ce_pack_weights).
//
// Four small launches per layer: W^T u (thread per column, coalesced over columns), normalise,
// W v (workgroup per row), finalise.  Sums are accumulated in f64 so that the result is at least as
// close to the exact value as any f32 summation order (tolerance budget: DESIGN.md).
#include "common.h"

namespace v2ce {
namespace {

__device__ __forceinline__ double block_sum(double v, double *sh) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    double s = 0.0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) s += sh[i];
    return s;
}

constexpr int kRowChunks = 16;   // W^T u is split over row chunks to fill the chip; partials are
                                 // reduced in a fixed order (bitwise reproducible, no atomics)

// part[c][j] = sum_{i in chunk c} W[i][j] * u[i]
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float *__restrict__ w,
                                                      const float *__restrict__ u, int rows,
                                                      int cols, double *__restrict__ part) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    const int per = (rows + kRowChunks - 1) / kRowChunks;
    const int lo = blockIdx.y * per, hi = (lo + per < rows) ? lo + per : rows;
    double s = 0.0;
#pragma unroll 4
    for (int i = lo; i < hi; ++i) s += (double)w[(long long)i * cols + j] * (double)u[i];
    part[(long long)blockIdx.y * cols + j] = s;
}

// t = sum of the row-chunk partials; v = t / (|t| + eps)     (single workgroup)
__global__ __launch_bounds__(1024) void sn_normalize_kernel(const double *__restrict__ part, int n,
                                                            float *__restrict__ t,
                                                            float *__restrict__ out) {
    __shared__ double sh[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double a = 0.0;
#pragma unroll
        for (int c = 0; c < kRowChunks; ++c) a += part[(long long)c * n + i];
        const float ti = (float)a;
        t[i] = ti;
        s += (double)ti * (double)ti;
    }
    __syncthreads();
    const float norm = (float)sqrt(block_sum(s, sh));
    const float den = norm + 1e-12f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = t[i] / den;
}

// s[i] = sum_j W[i][j] * v[j]     (workgroup per row)
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float *__restrict__ w,
                                                     const float *__restrict__ v, int cols,
                                                     float *__restrict__ s) {
    __shared__ double sh[4];
    const float *row = w + (long long)blockIdx.x * cols;
    double acc = 0.0;
    for (int j = threadIdx.x; j < cols; j += 256) acc += (double)row[j] * (double)v[j];
    const double tot = block_sum(acc, sh);
    if (threadIdx.x == 0) s[blockIdx.x] = (float)tot;
}

// u = s / (|s| + eps); sigma = u . s      (single workgroup)
__global__ __launch_bounds__(1024) void sn_finalize_kernel(const float *__restrict__ s, int n,
                                                           float *__restrict__ u,
                                                           float *__restrict__ sigma) {
    __shared__ double sh[16];
    double q = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) q += (double)s[i] * (double)s[i];
    const float norm = (float)sqrt(block_sum(q, sh));
    const float den = norm + 1e-12f;
    double d = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float ui = s[i] / den;
        u[i] = ui;
        d += (double)ui * (double)s[i];
    }
    const double dot = block_sum(d, sh);
    if (threadIdx.x == 0) sigma[0] = (float)dot;
}

}  // namespace
}  // namespace v2ce

using namespace v2ce;

extern "C" size_t v2ce_sn_workspace_bytes(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return ((size_t)rows + (size_t)cols) * sizeof(float) + (size_t)kRowChunks * cols * sizeof(double);
}

extern "C" int v2ce_sn_power_iter(float *u, float *v, const float *w_bar, int rows, int cols,
                                  float *sigma, void *workspace, size_t workspace_bytes,
                                  v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(u && v && w_bar && sigma && workspace, V2CE_ERR_BAD_ARG, "v2ce_sn_power_iter: null pointer");
    V2CE_REQUIRE(rows > 0 && cols > 0, V2CE_ERR_BAD_ARG, "v2ce_sn_power_iter: bad shape");
    V2CE_REQUIRE(workspace_bytes >= v2ce_sn_workspace_bytes(rows, cols), V2CE_ERR_WORKSPACE,
                 "v2ce_sn_power_iter: workspace %zu < %zu", workspace_bytes,
                 v2ce_sn_workspace_bytes(rows, cols));
    hipStream_t st = as_stream(stream);
    double *part = static_cast<double *>(workspace);                  // [kRowChunks][cols]
    float *t = reinterpret_cast<float *>(part + (size_t)kRowChunks * cols);   // [cols]
    float *s = t + cols;                                                       // [rows]
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3((cols + 255) / 256, kRowChunks), dim3(256), 0, st, w_bar, u,
                       rows, cols, part);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, st, part, cols, t, v);
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(rows), dim3(256), 0, st, w_bar, v, cols, s);
    hipLaunchKernelGGL(sn_finalize_kernel, dim3(1), dim3(1024), 0, st, s, rows, u, sigma);
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}
