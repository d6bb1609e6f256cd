// Load probe: unaligned dwordx4 buffer loads (correctness) and issue+landing cost of gathers by width.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: 80 dword loads/lane (rows of 34 floats), 1: 20 dwordx4 loads/lane, +misalign
__global__ __launch_bounds__(256) void probe(const float *x, int rowstride, long long cstride, int misalign, float *sink,
                                              unsigned long long *out, int *bad) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (long long)blockIdx.x * 8192), 0, 1 << 30, 0x00020000);
    float acc = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        float r[80];
#pragma unroll
        for (int i = 0; i < 80; ++i) {
            const int ch = i / 5, e = threadIdx.x + 256 * (i % 5), row = e / 34, col = e % 34;
            r[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (row * rowstride + col + misalign) * 4, (int)(ch * cstride * 4), 0));
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 80; ++i) acc += r[i];
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
    } else {
        u32x4 r[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int ch = i / 2, q = threadIdx.x + 256 * (i % 2), row = q / 9, col = 4 * (q % 9);
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (row * rowstride + col + misalign) * 4, (int)(ch * cstride * 4), 0);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int ch = i / 2, q = threadIdx.x + 256 * (i % 2), row = q / 9, col = 4 * (q % 9);
            const unsigned rr[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = __builtin_bit_cast(float, rr[e]);
                const float want = (float)((((long long)blockIdx.x * 8192 + ch * cstride + row * rowstride + col + misalign + e)) % 1000003);
                if (v != want) atomicAdd(bad + e + 4 * (((row * rowstride + col + misalign) & 3) != 0), 1);
                acc += v;
            }
        }
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
    }
    if (acc == 12345.678f) sink[0] = acc;
#endif
}

int main() {
    const size_t n = (size_t)1 << 28;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 1000003);
    float *x, *sink; unsigned long long *out; int *bad;
    hipMalloc(&x, n * 4); hipMalloc(&sink, 4); hipMalloc(&out, 1 << 20); hipMalloc(&bad, 32);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int mis : {0, 1, 3}) for (int mode : {0, 1}) {
        const int blocks = 256;
        std::vector<unsigned long long> o(blocks * 8);
        hipMemset(bad, 0, 32);
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, x, 346, 90000LL, mis, sink, out, bad);
            else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, x, 346, 90000LL, mis, sink, out, bad);
            hipDeviceSynchronize();
        }
        int nbv[8]; hipMemcpy(nbv, bad, 32, hipMemcpyDeviceToHost); int nb = 0; for (int k = 0; k < 8; ++k) nb += nbv[k];
        if (mode) printf("  mismatches by element, aligned: %d %d %d %d  unaligned: %d %d %d %d\n", nbv[0], nbv[1], nbv[2], nbv[3], nbv[4], nbv[5], nbv[6], nbv[7]);
        hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> a, b;
        for (int k = 0; k < blocks * 4; ++k) { a.push_back(o[2 * k]); b.push_back(o[2 * k + 1]); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        printf("%s misalign %d floats: issue %6llu cyc, landed %6llu cyc per wave (%d B/lane), mismatches %d\n",
               mode ? "32 x dwordx4" : "80 x dword  ", mis, a[a.size() / 2], b[b.size() / 2], mode ? 512 : 320, nb);
    }
    return 0;
}
