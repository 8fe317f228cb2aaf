// Build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value mfma_valu_overlap.hip -o mfma_valu_overlap ; run on an MI355X.
// Micro-benchmark: do VALU instructions of one wave execute in the shadow of another wave's MFMAs on the same SIMD?
// 512-thread workgroups, one per CU: waves 0-3 issue chains of independent v_mfma_f32_32x32x16_bf16, waves 4-7 issue
// v_fma_f32 (mode 1), v_exp_f32 (mode 2) or ds_read_b128 (mode 3).  Prints time of each side alone and together.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_valu, int prio) {
    __shared__ f32x4 lds[1024];
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x < 1024) lds[threadIdx.x & 1023] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    if (wave < 4) {
        if (prio == 1) __builtin_amdgcn_s_setprio(3);
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f); b[j] = (__bf16)0.5f; }
        float w[16];
        for (int j = 0; j < 16; ++j) w[j] = threadIdx.x * 0.01f + j;
        for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                if (MODE >= 10) asm volatile("s_nop %0" ::"n"(MODE >= 10 ? MODE - 10 : 0));  // pad: keep the next MFMA away from the issue port
                if (MODE == 4) {  // same-wave interleave: 8 independent v_fma behind every MFMA
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[(i & 1) * 8 + j] = fmaf(w[(i & 1) * 8 + j], 1.0001f, 0.5f);
                }
            }
        }
        float s = 0.f;
        for (int j = 0; j < 16; ++j) s += w[j];
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 12345.f) out[0] = s;
    } else {
        if (prio == 2) __builtin_amdgcn_s_setprio(3);
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 0.01f + j;
        f32x4 q = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < n_valu; ++it) {
            if (MODE == 1 || MODE >= 10) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
            } else if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]) * 0.001f;
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) q += lds[(threadIdx.x + j * 64 + it) & 1023];
            }
        }
        float s = q[0] + q[1];
        for (int j = 0; j < 16; ++j) s += v[j];
        if (s == 12345.f) out[1] = s;
    }
}

template <int MODE>
float run(float* out, int nm, int nv, int prio) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, nm, nv, prio);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, nm, nv, prio);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* out; hipMalloc(&out, 64);
    const int NM = 20000;  // 160k MFMAs x 32 cycles = 5.1M cycles ~ 2.1 ms
    printf("mfma alone: %.1f us\n", run<1>(out, NM, 0, 0));
    const int nv1 = 20000, nv2 = 20000, nv3 = 20000;
    printf("fma   alone %.1f us | both %.1f us | both+prio(mfma) %.1f us | both+prio(valu) %.1f us  (%d x 64 v_fma)\n", run<1>(out, 0, nv1, 0), run<1>(out, NM, nv1, 0), run<1>(out, NM, nv1, 1), run<1>(out, NM, nv1, 2), nv1);
    printf("exp   alone %.1f us | both %.1f us | both+prio(mfma) %.1f us | both+prio(valu) %.1f us  (%d x 16 v_exp + 16 v_mul)\n", run<2>(out, 0, nv2, 0), run<2>(out, NM, nv2, 0), run<2>(out, NM, nv2, 1), run<2>(out, NM, nv2, 2), nv2);
    printf("same-wave interleave (8 v_fma per MFMA, 64 per 8 MFMAs): %.1f us\n", run<4>(out, NM, 0, 0));
    printf("pad s_nop 0: mfma alone %.1f | both %.1f\n", run<10>(out, NM, 0, 0), run<10>(out, NM, nv1, 0));
    printf("pad s_nop 1: mfma alone %.1f | both %.1f\n", run<11>(out, NM, 0, 0), run<11>(out, NM, nv1, 0));
    printf("pad s_nop 3: mfma alone %.1f | both %.1f\n", run<13>(out, NM, 0, 0), run<13>(out, NM, nv1, 0));
    printf("pad s_nop 5: mfma alone %.1f | both %.1f\n", run<15>(out, NM, 0, 0), run<15>(out, NM, nv1, 0));
    printf("pad s_nop 7: mfma alone %.1f | both %.1f\n", run<17>(out, NM, 0, 0), run<17>(out, NM, nv1, 0));
    printf("ldsrd alone %.1f us | both %.1f us | both+prio(mfma) %.1f us | both+prio(valu) %.1f us  (%d x 16 ds_read_b128)\n", run<3>(out, 0, nv3, 0), run<3>(out, NM, nv3, 0), run<3>(out, NM, nv3, 1), run<3>(out, NM, nv3, 2), nv3);
    return 0;
}
