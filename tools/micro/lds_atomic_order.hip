// lds_atomic_order.hip -- does one wave-instruction of ds_add_rtn_u32 serve lanes that hit the SAME
// LDS word in ascending lane order?  (Undocumented; a stable counting sort could take its ranks
// straight from the returned values instead of a ballot match-any if it holds.)
// Every wave owns a private table; lanes draw random keys (range R) under random exec masks; the
// returned value must equal base + number of lower active lanes with the same key.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ unsigned rng(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

__global__ __launch_bounds__(256) void probe(int iters, int R, unsigned long long *bad, unsigned long long *total) {
    __shared__ unsigned tab[4][1024];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long nb = 0, nt = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = lane; i < 1024; i += 64) tab[wid][i] = 7u * i;
        __builtin_amdgcn_wave_barrier();
        const unsigned key = rng(s) % (unsigned)R;
        const bool act = (rng(s) & 7u) != 0u || (it & 1);
        unsigned got = 0;
        if (act) got = atomicAdd(&tab[wid][key], 1u);
        // expected: base + lower active lanes with the same key
        unsigned long long peers = __ballot(act);
        for (int b = 0; b < 10; ++b) {
            const bool bit = (key >> b) & 1;
            const unsigned long long m = __ballot(act && bit);
            peers &= bit ? m : ~m;
        }
        const unsigned want = 7u * key + (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
        if (act) { nt++; if (got != want) nb++; }
        __builtin_amdgcn_wave_barrier();
    }
    atomicAdd(bad, nb);
    atomicAdd(total, nt);
}

int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    for (int R : {1, 2, 3, 7, 32, 33, 64, 117, 234, 512, 1024}) {
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, 2000, R, d, d + 1);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("R=%4d: %llu of %llu returned values out of lane order\n", R, h[0], h[1]);
    }
    return 0;
}
