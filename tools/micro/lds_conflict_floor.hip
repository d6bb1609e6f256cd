// lds_conflict_floor.hip -- what SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE reads for LDS accesses whose addresses are RANDOM
// (a counting sort's histogram atomics and its scattered record writes: ldati_bucket_sort_kernel, ldati_tile_dense_kernel),
// against the same instructions on consecutive addresses.  Lanes of one 32-lane group that fall on the same bank with
// different addresses are served in extra cycles whatever the layout: for uniformly random words the expected maximum load of
// 32 balls in 32 bins is ~3.5, i.e. ~70 % of the LDS cycles are "conflict" cycles by construction -- no padding or swizzle of
// a histogram removes that; only fewer random accesses per record do.  One kernel per pattern, so that rocprofv3 --pmc lists
// them apart:  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -- ./lds_conflict_floor
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned rng(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>   // 0 = ds_add_rtn_u32 random word, 1 = ds_write_b32 random word, 2 = ds_add_rtn_u32 consecutive, 3 = ds_write_b32 consecutive,
                      // 4 = ds_add_rtn_u32 on keys that are SORTED within the wave (random, then ordered by lane: runs of equal / adjacent words)
__global__ __launch_bounds__(256) void pattern(int iters, int words, unsigned *sink) {
    extern __shared__ unsigned tab[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < words; i += 256) tab[i] = 0;
    __syncthreads();
    unsigned s = (blockIdx.x * 256 + tid) * 2654435761u + 99u, acc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned a;
        if (MODE == 0 || MODE == 1) a = rng(s) % (unsigned)words;
        else if (MODE == 4) a = ((rng(s) % 64u) * 0 + (unsigned)lane * ((unsigned)words / 64u) + rng(s) % ((unsigned)words / 64u));
        else a = (unsigned)((tid + 64 * it) % words);
        if (MODE == 0 || MODE == 2 || MODE == 4) acc += atomicAdd(&tab[a], 1u);
        else tab[a] = acc + it;
    }
    __syncthreads();
    if (tid == 0) sink[blockIdx.x] = acc + tab[lane];
}

int main() {
    unsigned *d;
    hipMalloc(&d, 4096 * 4);
    const int words = 8192;   // 32 KB table: a sort group's (key, category) histogram / record buffer
    hipLaunchKernelGGL(pattern<0>, dim3(1024), dim3(256), words * 4, 0, 4000, words, d);
    hipLaunchKernelGGL(pattern<1>, dim3(1024), dim3(256), words * 4, 0, 4000, words, d);
    hipLaunchKernelGGL(pattern<2>, dim3(1024), dim3(256), words * 4, 0, 4000, words, d);
    hipLaunchKernelGGL(pattern<3>, dim3(1024), dim3(256), words * 4, 0, 4000, words, d);
    hipLaunchKernelGGL(pattern<4>, dim3(1024), dim3(256), words * 4, 0, 4000, words, d);
    hipDeviceSynchronize();
    printf("done: pattern<0> random atomics, <1> random writes, <2> consecutive atomics, <3> consecutive writes, <4> lane-ordered random atomics\n");
    return 0;
}
