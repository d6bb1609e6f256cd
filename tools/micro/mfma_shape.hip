// Does the MFMA shape change the clock the chip holds under an fp16-MFMA-dense load?  Bare loops on
// random operands, one wave per SIMD, same output tile per wave (64 x 128 f32 accumulators = 128 regs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void loop(const _Float16 *in, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[8];
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f16x8 *>(in + ((blockIdx.x * 13 + i * 64 + lane) % 4096) * 8);
    for (int i = 0; i < 8; ++i) b[i] = *reinterpret_cast<const f16x8 *>(in + ((blockIdx.x * 7 + 1024 + i * 64 + lane) % 4096) * 8);
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[2][4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k)          // 2 k-steps of 16 = the K of one 16x16x32 step
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        acc[q][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q + 2 * k], b[f + 4 * k], acc[q][f], 0, 0, 0);
        }
        for (int q = 0; q < 2; ++q) for (int f = 0; f < 4; ++f) for (int r = 0; r < 16; ++r) s += acc[q][f][r];
    } else {
        f32x4 acc[4][8] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int f = 0; f < 8; ++f)
                    acc[q][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q], b[f], acc[q][f], 0, 0, 0);
        }
        for (int q = 0; q < 4; ++q) for (int f = 0; f < 8; ++f) for (int r = 0; r < 4; ++r) s += acc[q][f][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    _Float16 *in; float *out;
    hipMalloc(&in, 4096 * 8 * 2); hipMalloc(&out, 1024 * 256 * 4);
    _Float16 h[4096 * 8];
    unsigned x = 12345;
    for (int i = 0; i < 4096 * 8; ++i) { x = x * 1664525u + 1013904223u; h[i] = (_Float16)(((x >> 8) & 0xffff) / 65536.0f * 4.f - 2.f); }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    for (int rep = 0; rep < 3; ++rep) for (int shape : {32, 16}) {
        hipEventRecord(e0);
        if (shape == 32) hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        else hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * (shape == 32 ? 16 * 2.0 * 32 * 32 * 16 : 32 * 2.0 * 16 * 16 * 32);
        printf("mfma %s: %.2f ms, %.0f TFLOP/s (fp16 dense)\n", shape == 32 ? "32x32x16" : "16x16x32", ms, flop / ms / 1e9);
    }
    return 0;
}
