// Build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value trans_rate.hip -o trans_rate ; run on an MI355X.
// Issue cost of transcendental instructions: one wave per SIMD, 16 independent chains, N instructions per lane.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
extern "C" __device__ _Float16 __ocml_exp2_f16(_Float16);
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float v[16];
    _Float16 hv[16];
    for (int j = 0; j < 16; ++j) { v[j] = threadIdx.x * 0.001f + j * 0.01f; hv[j] = (_Float16)v[j]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) v[j] = fmaf(v[j], 1.0001f, 0.5f);
            if (MODE == 1) v[j] = __builtin_amdgcn_exp2f(v[j]);
            if (MODE == 2) v[j] = __builtin_amdgcn_rcpf(v[j]);
            if (MODE == 3) hv[j] = __builtin_amdgcn_rcph(hv[j]);
            if (MODE == 4) hv[j] = (_Float16)__ocml_exp2_f16(hv[j]);
            if (MODE == 5) v[j] = __builtin_amdgcn_rsqf(v[j]);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += v[j] + (float)hv[j];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE> void run(float* out, const char* name) {
    const int iters = 100000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s %8.1f us  -> %.2f ns per wave-instruction\n", name, ms * 1e3, ms * 1e6 / (iters * 16.0));
}
int main() {
    float* out; hipMalloc(&out, 64);
    run<0>(out, "v_fma_f32"); run<1>(out, "v_exp_f32"); run<2>(out, "v_rcp_f32"); run<3>(out, "v_rcp_f16"); run<4>(out, "v_exp_f16"); run<5>(out, "v_rsq_f32");
    return 0;
}
