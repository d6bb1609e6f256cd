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


// ---------------------------------------------------------------------------------------------
// All spectral-norm layers of a forward pass in FIVE launches (v2ce_sn_update_batch) instead of seven per
// layer: the per-layer sequence is launch-bound (84 launches, 1.25 ms alone) and, overlapped with the
// encoder, its many small workgroups keep taking CUs away from the persistent conv kernels, which need a
// whole CU each and walk their tiles statically (measured: 0.9 ms per forward).  Same arithmetic and
// summation orders as the kernels above; max |W| rides on the W v pass (max |W / sigma| = max |W| / |sigma|:
// correctly rounded division is monotone), so the weights are read three times, not four.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxBatch = 16;   // (the batch travels to the kernels by value: 16 x 152 bytes + the prefixes stay below the 4 KB of kernel arguments)
struct SnLayer {
    const float *w;
    float *u, *v;
    _Float16 *packed;           // v2ce_pack_weights_f16x2 buffer: hi plane | lo plane | {max |w/sigma|, pre-scale}
    double *part;               // [kRowChunks][cols]
    float *t, *s, *rowmax, *sigma;
    int rows, cols, k3, cin;
    int up_c0;                  // > 0: `packed` is a v2ce_pack_weights_f16x2_up buffer whose first up_c0 input channels are also folded
    int wt;                     // 1: `packed` is a v2ce_pack_weights_f16x2_wt buffer (36 tap slots; written by conv3d_wt.hip's pack passes)
    _Float16 *packed_skip;      // up_c0 > 0 and non-null: ALSO the Winograd-T planes of input channels [up_c0, cin) (v2ce_pack_weights_f16x2_wt_slice)
    int no_pack, no_iterate;    // v2ce_sn_layer.flags (round 6: w_bar packed once, 1 / sigma carried in the epilogue scale)
    const float *bn_scale, *sigma_src;
    float *scale_out, *inv_sigma_out;
    float wmax;
};
struct SnBatch {
    SnLayer L[kMaxBatch];
    int n;
    int col_blk[kMaxBatch + 1];   // prefix of ceil(cols / 256)
    int row_blk[kMaxBatch + 1];   // prefix of rows
    int el_blk[kMaxBatch + 1];    // prefix of (rows / 32) * (Cin / 16): pack workgroups
};

static_assert(sizeof(SnBatch) <= 3584, "SnBatch travels by value in the kernel arguments");

__device__ __forceinline__ int find_layer(const int *prefix, int n, int b) {
    int l = 0;
    while (l + 1 < n && b >= prefix[l + 1]) ++l;
    return l;
}

__global__ __launch_bounds__(256) void sn_batch_wt_u_kernel(SnBatch B) {
    const int l = find_layer(B.col_blk, B.n, blockIdx.x);
    const SnLayer &P = B.L[l];
    const int j = (blockIdx.x - B.col_blk[l]) * 256 + threadIdx.x;
    if (j >= P.cols) return;
    const int per = (P.rows + kRowChunks - 1) / kRowChunks;
    const int lo = blockIdx.y * per, hi = (lo + per < P.rows) ? lo + per : P.rows;
    double s = 0.0;
#pragma unroll 4
    for (int i = lo; i < hi; ++i) s += (double)P.w[(long long)i * P.cols + j] * (double)P.u[i];
    P.part[(long long)blockIdx.y * P.cols + j] = s;
}

// t[j] = sum of the row-chunk partials, all layers' columns in parallel (one workgroup per layer spent 54 us here)
__global__ __launch_bounds__(256) void sn_batch_colsum_kernel(SnBatch B) {
    const int l = find_layer(B.col_blk, B.n, blockIdx.x);
    const SnLayer &P = B.L[l];
    const int i = (blockIdx.x - B.col_blk[l]) * 256 + threadIdx.x;
    if (i >= P.cols) return;
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < kRowChunks; ++c) a += P.part[(long long)c * P.cols + i];
    P.t[i] = (float)a;
}

__global__ __launch_bounds__(1024) void sn_batch_normalize_kernel(SnBatch B) {
    __shared__ double sh[16];
    const SnLayer &P = B.L[blockIdx.x];
    const int n = P.cols;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {          // same order of summation as sn_normalize_kernel
        const float ti = P.t[i];
        s += (double)ti * (double)ti;
    }
    __syncthreads();
    const float norm = (float)sqrt(block_sum(s, sh));
    const float den = norm + 1e-12f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) P.v[i] = P.t[i] / den;
}

__global__ __launch_bounds__(256) void sn_batch_w_v_kernel(SnBatch B) {
    __shared__ double sh[4];
    __shared__ float mx[4];
    const int l = find_layer(B.row_blk, B.n, blockIdx.x);
    const SnLayer &P = B.L[l];
    const int r = blockIdx.x - B.row_blk[l];
    const float *row = P.w + (long long)r * P.cols;
    double acc = 0.0;
    float m = 0.0f;
#pragma unroll 6
    for (int j = threadIdx.x; j < P.cols; j += 256) {       // (several loads in flight; the sum's order is unchanged)
        const float wv = row[j];
        acc += (double)wv * (double)P.v[j];
        m = fmaxf(m, fabsf(wv));
    }
    const double tot = block_sum(acc, sh);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) mx[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        P.s[r] = (float)tot;
        P.rowmax[r] = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
    }
}

__global__ __launch_bounds__(1024) void sn_batch_finalize_kernel(SnBatch B) {
    __shared__ double sh[16];
    __shared__ float mxs[16];
    const SnLayer &P = B.L[blockIdx.x];
    const int n = P.rows;
    double q = 0.0;
    float m = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        q += (double)P.s[i] * (double)P.s[i];
        m = fmaxf(m, P.rowmax[i]);
    }
    const float norm = (float)sqrt(block_sum(q, sh));
    const float den = norm + 1e-12f;
    double d = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float ui = P.s[i] / den;
        P.u[i] = ui;
        d += (double)ui * (double)P.s[i];
    }
    const double dot = block_sum(d, sh);
    if (P.scale_out) {                                         // the convolution's epilogue scale with 1 / sigma in it (round 6)
        const float sgf = (float)dot;
        for (int i = threadIdx.x; i < n; i += blockDim.x) P.scale_out[i] = P.bn_scale[i] / sgf;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) mxs[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, mxs[i]);
        const float sg = (float)dot;
        P.sigma[0] = sg;
        if (P.inv_sigma_out) P.inv_sigma_out[0] = 1.0f / sg;
        if (P.no_pack) return;                                 // (the packed planes and their tail are the caller's: packed once)
        if (P.wt) {                                            // Winograd-T planes: the bound 1.5 max |w / sigma| of conv3d_wt.hip's pack, pass 0
            float *tail = reinterpret_cast<float *>(P.packed + 2ll * P.rows * P.cin * 36);
            const float bound = 1.5f * fabsf(m / sg);
            tail[0] = bound; tail[1] = pow2_prescale(bound); tail[2] = 0.0f; tail[3] = 0.0f;
            return;
        }
        if (P.packed_skip) {                                   // the skip channels' Winograd-T planes: the same bound on |G|
            float *tail = reinterpret_cast<float *>(P.packed_skip + 2ll * P.rows * (P.cin - P.up_c0) * 36);
            const float bound = 1.5f * fabsf(m / sg);
            tail[0] = bound; tail[1] = pow2_prescale(bound); tail[2] = 0.0f; tail[3] = 0.0f;
        }
        float *tail = reinterpret_cast<float *>(P.packed + 2ll * P.rows * P.cols);
        const float am = fabsf(m / sg);                        // = max |w / sigma| of weights_absmax_kernel
        tail[0] = am;
        tail[1] = pow2_prescale(am);
        if (P.up_c0) { tail[2] = 0.0f; tail[3] = 0.0f; }      // (an up buffer's tail is 16 bytes; the fold passes may still raise tail[0])
    }
}

// pack_weights_f16x2_kernel's (conv3d.hip) values for every layer of the batch, with both sides coalesced: a
// workgroup takes 32 output channels x one 16-channel group -- 32 runs of 16 * k3 contiguous floats of W -- and
// writes, per tap and plane, the 32 x 16 halves that are contiguous in the packed layout [plane][tap][cg][co][16]
// (the element-wise form read W with a stride of k3 floats per lane: 162 us for 151 MB).
// K3 > 0: every layer of the batch has that kernel volume (the index arithmetic is then multiply-shift instead of two
// integer divisions per element, which made the kernel VALU-bound: 150 us)
template <int K3>
__global__ __launch_bounds__(256) void sn_batch_pack_kernel(SnBatch B) {
    extern __shared__ _Float16 pk_smem[];               // [2][k3][32][16] halves
    const int l = find_layer(B.el_blk, B.n, blockIdx.x);
    const SnLayer &P = B.L[l];
    const int CG = P.cin / 16, k3 = K3 > 0 ? K3 : P.k3, run = 16 * k3;
    const int blk = blockIdx.x - B.el_blk[l];            // (co block, cg)
    const int cg = blk % CG, co0 = (blk / CG) * 32;
    const long long n = (long long)P.rows * P.cols;
    float *tail = reinterpret_cast<float *>(P.packed + 2 * n);
    // (tail[0] may have been raised behind the finalize kernel by the folded sums of an up layer: the scale is derived here)
    // a pack-only layer (V2CE_SN_NO_ITERATE) brings its sigma and max |w|: max |w / sigma| = max |w| / |sigma| (the correctly
    // rounded division is monotone), the value weights_absmax_kernel finds
    const float sigma = P.no_iterate ? P.sigma_src[0] : P.sigma[0];
    const float amax = P.no_iterate ? fabsf(P.wmax / sigma) : tail[0];
    const float w_scale = pow2_prescale(amax);
    if (blk == 0 && threadIdx.x == 0) {
        tail[1] = w_scale;
        if (P.no_iterate) tail[0] = amax;
    }
    // (a tap's 512 halves are padded by one dword: consecutive lanes hold consecutive taps, 1 KiB apart = one LDS bank)
    _Float16 *hi = pk_smem, *lo = pk_smem + k3 * 514;
    constexpr int U = 6;                                 // loads in flight per thread (two workgroups per CU: latency-bound otherwise)
    for (int e0 = threadIdx.x; e0 < 32 * run; e0 += 256 * U) {
        float vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u;
            const int col = e / run, rem = e - col * run;
            vv[u] = e < 32 * run ? P.w[((long long)(co0 + col) * P.cin + cg * 16) * k3 + rem] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u;
            if (e < 32 * run) {
                const int col = e / run, rem = e - col * run;    // rem = j * k3 + tap
                const int j = rem / k3, tap = rem - j * k3;
                float v = vv[u] / sigma;
                v *= w_scale;
                const _Float16 h = (_Float16)v;
                hi[tap * 514 + col * 16 + j] = h;
                lo[tap * 514 + col * 16 + j] = (_Float16)(v - (float)h);
            }
        }
    }
    __syncthreads();
    // per tap: 512 halves = 256 dwords of each plane, contiguous at ((tap * CG + cg) * rows + co0) * 16
    const unsigned *hi32 = reinterpret_cast<const unsigned *>(hi), *lo32 = reinterpret_cast<const unsigned *>(lo);
    for (int tap = 0; tap < k3; ++tap) {
        const long long o = (((long long)tap * CG + cg) * P.rows + co0) * 16;       // halves
        reinterpret_cast<unsigned *>(P.packed + o)[threadIdx.x] = hi32[tap * 257 + threadIdx.x];
        reinterpret_cast<unsigned *>(P.packed + n + o)[threadIdx.x] = lo32[tap * 257 + threadIdx.x];
    }
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

static size_t sn_batch_layer_bytes(int rows, int cols) {
    // part f64 [kRowChunks][cols] | t [cols] | s [rows] | rowmax [rows] | sigma [4]
    return (size_t)kRowChunks * cols * 8 + ((size_t)cols + 2 * (size_t)rows + 4) * 4;
}

extern "C" size_t v2ce_sn_batch_workspace_bytes(const v2ce_sn_layer *layers, int n) {
    if (!layers || n <= 0 || n > kMaxBatch) return 0;
    size_t b = 0;
    for (int l = 0; l < n; ++l) b += (sn_batch_layer_bytes(layers[l].rows, layers[l].cols) + 15) / 16 * 16;
    return b;
}

extern "C" int v2ce_sn_update_batch(const v2ce_sn_layer *layers, int n, void *workspace, size_t workspace_bytes,
                                    v2ce_stream_t stream) {
    clear_error();
    V2CE_REQUIRE(layers && workspace && n > 0 && n <= kMaxBatch, V2CE_ERR_BAD_ARG, "v2ce_sn_update_batch: 1..%d layers", kMaxBatch);
    V2CE_REQUIRE(workspace_bytes >= v2ce_sn_batch_workspace_bytes(layers, n), V2CE_ERR_WORKSPACE,
                 "v2ce_sn_update_batch: workspace %zu < %zu", workspace_bytes, v2ce_sn_batch_workspace_bytes(layers, n));
    SnBatch B{};
    B.n = n;
    int n_iter = 0;
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    for (int l = 0; l < n; ++l) {
        const v2ce_sn_layer &in = layers[l];
        const bool no_pack = (in.flags & V2CE_SN_NO_PACK) != 0, no_it = (in.flags & V2CE_SN_NO_ITERATE) != 0;
        V2CE_REQUIRE((in.flags & ~3) == 0 && !(no_pack && no_it), V2CE_ERR_BAD_ARG, "v2ce_sn_update_batch: layer %d: bad flags %d", l, in.flags);
        V2CE_REQUIRE(in.w_bar && (no_it || (in.u && in.v)) && (no_pack || in.packed) && in.rows > 0 && in.cols > 0 && (in.k3 == 27 || in.k3 == 1) &&
                     in.cols % in.k3 == 0 && (in.cols / in.k3) % 16 == 0, V2CE_ERR_BAD_ARG,
                     "v2ce_sn_update_batch: layer %d: bad argument", l);
        V2CE_REQUIRE(!no_it || (in.sigma_src && in.wmax >= 0.0f && in.up_c0 == 0 && in.wt == 0 && !in.packed_skip), V2CE_ERR_BAD_ARG,
                     "v2ce_sn_update_batch: layer %d: a pack-only layer brings sigma_src and wmax and has plain planes", l);
        V2CE_REQUIRE(!in.scale_out || (in.bn_scale && !no_it), V2CE_ERR_BAD_ARG, "v2ce_sn_update_batch: layer %d: scale_out needs bn_scale and an iterated layer", l);
        n_iter += no_it ? 0 : 1;
        SnLayer &o = B.L[l];
        o.w = in.w_bar; o.u = in.u; o.v = in.v; o.packed = static_cast<_Float16 *>(in.packed);
        o.no_pack = no_pack; o.no_iterate = no_it;
        o.bn_scale = in.bn_scale; o.scale_out = in.scale_out; o.inv_sigma_out = in.inv_sigma_out; o.sigma_src = in.sigma_src; o.wmax = in.wmax;
        o.rows = in.rows; o.cols = in.cols; o.k3 = in.k3; o.cin = in.cols / in.k3;
        o.up_c0 = in.up_c0;
        o.wt = in.wt;
        o.packed_skip = static_cast<_Float16 *>(in.packed_skip);
        V2CE_REQUIRE(!in.packed_skip || in.up_c0 > 0, V2CE_ERR_BAD_ARG, "v2ce_sn_update_batch: layer %d: packed_skip needs up_c0", l);
        V2CE_REQUIRE(in.wt == 0 || (in.wt == 1 && in.k3 == 27 && in.up_c0 == 0), V2CE_ERR_BAD_ARG,
                     "v2ce_sn_update_batch: layer %d: wt must be 0 or 1, on a 3x3x3 layer without up_c0", l);
        V2CE_REQUIRE(in.up_c0 == 0 || (in.k3 == 27 && in.up_c0 > 0 && in.up_c0 % 16 == 0 && in.up_c0 < o.cin), V2CE_ERR_BAD_ARG,
                     "v2ce_sn_update_batch: layer %d: up_c0 must be a multiple of 16 below Cin of a 3x3x3 layer", l);
        o.part = reinterpret_cast<double *>(ws);
        o.t = reinterpret_cast<float *>(o.part + (size_t)kRowChunks * in.cols);
        o.s = o.t + in.cols;
        o.rowmax = o.s + in.rows;
        o.sigma = o.rowmax + in.rows;
        ws += (sn_batch_layer_bytes(in.rows, in.cols) + 15) / 16 * 16;
        B.col_blk[l + 1] = B.col_blk[l] + (in.cols + 255) / 256;
        B.row_blk[l + 1] = B.row_blk[l] + in.rows;
        V2CE_REQUIRE(in.rows % 32 == 0 && in.k3 * 2056 <= 64 * 1024, V2CE_ERR_UNSUPPORTED,
                     "v2ce_sn_update_batch: layer %d: rows %% 32 != 0 or k3 too large", l);
        // pack: (32 output channels, 16-channel group); the Winograd-T layers are packed by conv3d_wt.hip
        B.el_blk[l + 1] = B.el_blk[l] + ((in.wt || no_pack) ? 0 : (in.rows / 32) * (in.cols / in.k3 / 16));
    }
    V2CE_REQUIRE(n_iter == 0 || n_iter == n, V2CE_ERR_UNSUPPORTED,
                 "v2ce_sn_update_batch: pack-only layers (V2CE_SN_NO_ITERATE) go into a call of their own");
    hipStream_t st = as_stream(stream);
    if (n_iter) {
        hipLaunchKernelGGL(sn_batch_wt_u_kernel, dim3(B.col_blk[n], kRowChunks), dim3(256), 0, st, B);
        hipLaunchKernelGGL(sn_batch_colsum_kernel, dim3(B.col_blk[n]), dim3(256), 0, st, B);
        hipLaunchKernelGGL(sn_batch_normalize_kernel, dim3(n), dim3(1024), 0, st, B);
        hipLaunchKernelGGL(sn_batch_w_v_kernel, dim3(B.row_blk[n]), dim3(256), 0, st, B);
        hipLaunchKernelGGL(sn_batch_finalize_kernel, dim3(n), dim3(1024), 0, st, B);
    }
    // decoder conv1 layers (v2ce_conv3d_fwd_up2): the folded sums of W / sigma join the maximum the common pre-scale is derived from
    const float *uf_w[kMaxBatch], *uf_sigma[kMaxBatch];
    void *uf_packed[kMaxBatch];
    int uf_rows[kMaxBatch], uf_cin[kMaxBatch], uf_c0[kMaxBatch], n_uf = 0;
    for (int l = 0; l < n; ++l)
        if (B.L[l].up_c0 && !B.L[l].no_pack && n_uf < 8) {
            uf_w[n_uf] = B.L[l].w; uf_sigma[n_uf] = B.L[l].sigma; uf_packed[n_uf] = B.L[l].packed;
            uf_rows[n_uf] = B.L[l].rows; uf_cin[n_uf] = B.L[l].cin; uf_c0[n_uf] = B.L[l].up_c0;
            ++n_uf;
        }
    {
        int n_up = 0;
        for (int l = 0; l < n; ++l) n_up += (B.L[l].up_c0 && !B.L[l].no_pack) ? 1 : 0;
        V2CE_REQUIRE(n_up <= 8, V2CE_ERR_UNSUPPORTED, "v2ce_sn_update_batch: at most eight layers with up_c0");
    }
    if (n_uf) {
        const int rc = v2ce_up_fold_batch(uf_w, uf_rows, uf_cin, uf_c0, uf_sigma, uf_packed, n_uf, 0, st);
        if (rc != V2CE_OK) return rc;
    }
    // Winograd-T layers (v2ce_conv3d_fwd_wt): max |G| of W / sigma, then the planes -- two launches for all of them
    const float *wt_w[kMaxBatch], *wt_sigma[kMaxBatch];
    void *wt_packed[kMaxBatch];
    int wt_rows[kMaxBatch], wt_cin[kMaxBatch], wt_tot[kMaxBatch], wt_ci0[kMaxBatch], n_wt = 0;
    for (int l = 0; l < n; ++l) {
        if (B.L[l].no_pack) continue;
        if (B.L[l].wt) {
            wt_w[n_wt] = B.L[l].w; wt_sigma[n_wt] = B.L[l].sigma; wt_packed[n_wt] = B.L[l].packed;
            wt_rows[n_wt] = B.L[l].rows; wt_cin[n_wt] = B.L[l].cin; wt_tot[n_wt] = B.L[l].cin; wt_ci0[n_wt] = 0;
            ++n_wt;
        } else if (B.L[l].packed_skip) {                   // decoder conv1: its skip channels for the Winograd-T launch
            wt_w[n_wt] = B.L[l].w; wt_sigma[n_wt] = B.L[l].sigma; wt_packed[n_wt] = B.L[l].packed_skip;
            wt_rows[n_wt] = B.L[l].rows; wt_cin[n_wt] = B.L[l].cin - B.L[l].up_c0; wt_tot[n_wt] = B.L[l].cin; wt_ci0[n_wt] = B.L[l].up_c0;
            ++n_wt;
        }
    }
    if (n_wt) {                                            // (pass 0 -- the bound on |G| -- is what the finalize kernel has just written)
        const int rc = v2ce_wt_pack_batch(wt_w, wt_sigma, wt_packed, wt_rows, wt_cin, n_wt, 1, st, wt_tot, wt_ci0);
        if (rc != V2CE_OK) return rc;
    }
    int k3max = 1;
    bool all27 = true;
    for (int l = 0; l < n; ++l) {
        k3max = layers[l].k3 > k3max ? layers[l].k3 : k3max;
        all27 = all27 && layers[l].k3 == 27;
    }
    if (B.el_blk[n] == 0) {}                               // (every layer is a Winograd-T layer)
    else if (all27) hipLaunchKernelGGL(sn_batch_pack_kernel<27>, dim3(B.el_blk[n]), dim3(256), (size_t)27 * 2056, st, B);
    else hipLaunchKernelGGL(sn_batch_pack_kernel<0>, dim3(B.el_blk[n]), dim3(256), (size_t)k3max * 2056, st, B);
    if (n_uf) {
        const int rc = v2ce_up_fold_batch(uf_w, uf_rows, uf_cin, uf_c0, uf_sigma, uf_packed, n_uf, 1, st);
        if (rc != V2CE_OK) return rc;
    }
    V2CE_HIP_CHECK(hipGetLastError());
    return V2CE_OK;
}
