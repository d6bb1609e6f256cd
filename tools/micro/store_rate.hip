// Store-rate probe: how many cycles does a wave spend per global store instruction, by access shape?
//   hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// mode 0: dword per lane, 32 lanes contiguous (128 B) x 2 channel planes per instruction, 128 instr
// mode 1: dwordx4 per lane, 8 lanes = 128 B run, 8 channel planes per instruction, 32 instr
// mode 2: dword per lane, fully contiguous 256 B per instruction (consecutive instr consecutive)
// mode 3: dwordx4 per lane, fully contiguous 1 KiB per instruction
template <int MODE>
__global__ __launch_bounds__(256) void probe(float *y, long long cstride, int rowstride, unsigned long long *out, int waves_active, int pre_mfma) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= waves_active) return;
    float *base = y + (long long)blockIdx.x * 4096 + wave * 1024;   // spatial offset of this wave's tile
    float v = lane * 1.0f;
    if (pre_mfma) {   // an MFMA-dense phase first (random-ish operands), like the main loop of a conv kernel
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        f16x8 x, w;
        for (int c = 0; c < 8; ++c) { x[c] = (_Float16)(0.37f * ((lane * 7 + c * 13) % 31) - 5.f); w[c] = (_Float16)(0.21f * ((lane * 3 + c * 5) % 29) - 3.f); }
        for (int i = 0; i < pre_mfma; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, w, a3, 0, 0, 0);
        }
        v += a0[0] + a1[1] + a2[2] + a3[3];
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 128; ++i) {
            const int ch = 2 * (i / 4) + (lane >> 5), row = i & 3;
            base[(long long)ch * cstride + row * rowstride + (lane & 31)] = v + i;
        }
    } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int ch = 8 * (i / 4) + (lane >> 3), row = i & 3;
            *reinterpret_cast<f32x4 *>(base + (long long)ch * cstride + row * rowstride + 4 * (lane & 7)) = f32x4{v, v + i, v, v};
        }
    } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 128; ++i) base[i * 64 + lane] = v + i;
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) *reinterpret_cast<f32x4 *>(base + i * 256 + 4 * lane) = f32x4{v, v + i, v, v};
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}

template <int MODE>
void run(const char *name, int blocks, int waves_active, float *y, unsigned long long *out, int pre_mfma = 0, int rowstride = 346) {
    const long long cstride = 260 * (long long)rowstride;   // floats between channel planes
    std::vector<unsigned long long> h(blocks * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(out, 0, blocks * 8 * sizeof(unsigned long long));
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, y, cstride, rowstride, out, waves_active, pre_mfma);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> a, b;
    for (int k = 0; k < blocks * 4; ++k) if (h[2 * k]) { a.push_back(h[2 * k]); b.push_back(h[2 * k + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-44s pre-mfma %5d blocks %4d waves/blk %d: issue %6llu cyc, issue+drain %6llu cyc (median per wave; 32 KiB per wave)\n", name, pre_mfma, blocks,
           waves_active, a[a.size() / 2], b[b.size() / 2]);
}

int main() {
    float *y; unsigned long long *out;
    hipMalloc(&y, (size_t)1 << 31); hipMalloc(&out, 1 << 20);
    for (int pre : {0, 4000}) for (int blocks : {256, 1452}) {
        run<0>("dword/lane, 2 planes x 128 B per instr", blocks, 4, y, out, pre);
        run<1>("dwordx4/lane, 8 planes x 128 B per instr", blocks, 4, y, out, pre);
        run<3>("dwordx4/lane, contiguous 1 KiB per instr", blocks, 4, y, out, pre);
        // the same pieces with rows padded to 352 floats: every 128-byte piece is one aligned cache line
        run<0>("dword/lane, 2 planes x 128 B, pitch 352", blocks, 4, y, out, pre, 352);
        run<1>("dwordx4/lane, 8 planes x 128 B, pitch 352", blocks, 4, y, out, pre, 352);
    }
    return 0;
}
