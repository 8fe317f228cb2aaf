#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o, unsigned s) {
    unsigned a = o[threadIdx.x], b = a;
    asm volatile("v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(a) : "v"(s));
    asm volatile("v_add_u16 %0, %1, %0" : "+v"(b) : "v"(s));
    o[threadIdx.x] = a; o[64 + threadIdx.x] = b;
}
int main() {
    unsigned h[128]; for (int i = 0; i < 128; ++i) h[i] = 0x0001E000u + i;
    unsigned* d; hipMalloc(&d, sizeof h); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 0x2000u);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("in 0x0001E000 + 0x2000: sdwa(preserve) = 0x%08x, vop2 = 0x%08x\n", h[0], h[64]);
    return 0;
}
