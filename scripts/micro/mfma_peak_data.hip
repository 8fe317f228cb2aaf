// Build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value mfma_peak_data.hip -o mfma_peak_data ; run on an MI355X.
// Sustained rate of a bare v_mfma_f32_32x32x16_bf16 stream (operands in registers, 1 or 2 waves per SIMD, all 256 CUs) as a
// function of the operand DATA: zero, constant, small-range and full-random bf16.  The chip lowers its clock with switching
// activity (DVFS), so the 2.5 PFLOP/s dense peak is not reachable on random operands.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(512) void k(float* out, const u32x4* ops, int iters) {
    // 8 A and 8 B fragments per lane, read once
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, ops[(i * 2 + 0) * 512 + threadIdx.x]);
        b[i] = __builtin_bit_cast(bf16x8, ops[(i * 2 + 1) * 512 + threadIdx.x]);
    }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + u) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    if (s == 12345.678f) out[0] = s;
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
// same FLOPs with v_mfma_f32_16x16x32_bf16 (16 cycles each, 32 accumulator tiles of 16x16)
__global__ __launch_bounds__(512) void k16(float* out, const u32x4* ops, int iters) {
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, ops[(i * 2 + 0) * 512 + threadIdx.x]);
        b[i] = __builtin_bit_cast(bf16x8, ops[(i * 2 + 1) * 512 + threadIdx.x]);
    }
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[(i + 16 * (u & 1))] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7], b[(i + u) & 7], acc[(i + 16 * (u & 1))], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const int n = 16 * 512 * 4;
    uint32_t* h = (uint32_t*)malloc(n * 4);
    u32x4* d;
    float* out;
    hipMalloc(&d, n * 4);
    hipMalloc(&out, 64);
    const char* names[4] = {"all zero", "constant 0.5", "random mantissa, |v| in [0.5,1), one sign", "random sign / exponent 2^-7..2 / mantissa"};
    for (int mode = 0; mode < 4; ++mode) {
        uint32_t st = 12345u;
        for (int i = 0; i < n; ++i) {
            uint32_t w = 0;
            for (int hh = 0; hh < 2; ++hh) {
                st = st * 1664525u + 1013904223u;
                uint32_t v = 0;
                if (mode == 1) v = 0x3f00u;
                if (mode == 2) v = 0x3f00u | ((st >> 20) & 0x7fu);
                if (mode == 3) v = ((st >> 16) & 0x8000u) | ((0x78u + ((st >> 12) & 7u)) << 7) | ((st >> 20) & 0x7fu);
                w |= v << (16 * hh);
            }
            h[i] = w;
        }
        hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
        for (int threads = 256; threads <= 512; threads += 256) {
            const int iters = 20000;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, d, iters);
            hipEventRecord(e0);
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 3;
            const double flop = 256.0 * (threads / 64) * iters * 8.0 * 32 * 32 * 16 * 2;
            printf("%-46s %d waves/SIMD: %8.1f us  %7.1f TFLOP/s  (%.2f GHz-equivalent at 32 cyc/MFMA)\n", names[mode], threads / 256,
                   ms * 1e3, flop / (ms * 1e-3) / 1e12, (double)iters * 8 * 32 * (threads / 256) / (ms * 1e-3) / 1e9);
            hipLaunchKernelGGL(k16, dim3(256), dim3(threads), 0, 0, out, d, iters);
            hipEventRecord(e0);
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k16, dim3(256), dim3(threads), 0, 0, out, d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 3;
            const double flop16 = 256.0 * (threads / 64) * iters * 16.0 * 16 * 16 * 32 * 2;
            printf("%-46s %d waves/SIMD: %8.1f us  %7.1f TFLOP/s  [16x16x32]\n", names[mode], threads / 256, ms * 1e3, flop16 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
