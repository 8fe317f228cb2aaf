// Build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value mfma_same_wave_fillers.hip -o mfma_same_wave_fillers ; run on an MI355X.
// How many independent vector instructions of the SAME wave fit behind each MFMA for free (one wave per SIMD, random operands)?
// NF v_fma_f32 (and optionally one v_exp_f32) are placed after every v_mfma_f32_16x16x32_bf16 by the compiler's own order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NF, int NEXP>
__global__ __launch_bounds__(256) void k(float* out, const u32x4* ops, int iters) {
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, ops[(i * 2 + 0) * 512 + threadIdx.x]);
        b[i] = __builtin_bit_cast(bf16x8, ops[(i * 2 + 1) * 512 + threadIdx.x]);
    }
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float w[16];
    for (int j = 0; j < 16; ++j) w[j] = threadIdx.x * 0.01f + j;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i + 16 * u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7], b[(i + u) & 7], acc[i + 16 * u], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < NF; ++f) w[(i * NF + f) & 15] = fmaf(w[(i * NF + f) & 15], 1.0001f, 0.5f);
#pragma unroll
                for (int f = 0; f < NEXP; ++f) w[(i + 7 * f) & 15] = __builtin_amdgcn_exp2f(w[(i + 7 * f) & 15] * 0.001f);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    for (int j = 0; j < 16; ++j) s += w[j];
    if (s == 12345.678f) out[0] = s;
}

template <int NF, int NEXP>
void run(float* out, const u32x4* d, const char* name) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NF, NEXP>), dim3(256), dim3(256), 0, 0, out, d, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NF, NEXP>), dim3(256), dim3(256), 0, 0, out, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double flop = 256.0 * 4 * iters * 16.0 * 16 * 16 * 32 * 2;
    printf("%-40s %8.1f us  %7.1f TFLOP/s\n", name, ms * 1e3, flop / (ms * 1e-3) / 1e12);
}

int main() {
    const int n = 16 * 512 * 4;
    uint32_t* h = (uint32_t*)malloc(n * 4);
    u32x4* d;
    float* out;
    hipMalloc(&d, n * 4);
    hipMalloc(&out, 64);
    uint32_t st = 12345u;
    for (int i = 0; i < n; ++i) {
        uint32_t w = 0;
        for (int hh = 0; hh < 2; ++hh) {
            st = st * 1664525u + 1013904223u;
            w |= (((st >> 16) & 0x8000u) | ((0x78u + ((st >> 12) & 7u)) << 7) | ((st >> 20) & 0x7fu)) << (16 * hh);
        }
        h[i] = w;
    }
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    run<0, 0>(out, d, "16x16x32 alone");
    run<1, 0>(out, d, "+ 1 v_fma per MFMA");
    run<2, 0>(out, d, "+ 2 v_fma per MFMA");
    run<3, 0>(out, d, "+ 3 v_fma per MFMA");
    run<1, 1>(out, d, "+ 1 v_fma + (v_mul, v_exp) per MFMA");
    run<0, 1>(out, d, "+ (v_mul, v_exp) per MFMA");
    return 0;
}
