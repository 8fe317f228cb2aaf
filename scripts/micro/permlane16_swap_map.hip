// Prints what v_permlane16_swap_b32 does on gfx950: for lanes 0, 16, 32, 48 the (first, second) results given first = lane, second = 1000 + lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x * 2] = r[0];
    out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 128 * 4);
    k<<<1, 64>>>(d);
    unsigned h[128];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 16) printf("lane %2d: first' = %u  second' = %u\n", l, h[2 * l], h[2 * l + 1]);
    return 0;
}
