// v_permlane16_swap_b32 on gfx950: which lanes of the two operands change places.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/permlane16_swap.hip -o /tmp/pl16 && /tmp/pl16
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *p) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    p[threadIdx.x] = r[0];
    p[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int o = 0; o < 2; ++o) {
        printf("r[%d]:", o);
        for (int i = 0; i < 64; ++i) printf(" %u", h[64 * o + i]);
        printf("\n");
    }
    return 0;
}
