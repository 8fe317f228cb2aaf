// Backward of the U-Net's single-head self-attention (reference fastgen/networks/EDM/network.py:160-196: AttentionOp and its
// hand-written backward; :290-296 in UNetBlock.forward) - part of the training step, SURVEY §8(f)1.
//
//   forward (per image):  S = q k^T / sqrt(C),  P = softmax_rows(S),  O = P v            q, k, v, O: [T][C], T = 256 | 64, C = 256
//   backward, given dO:   dv = P^T dO,  dP = dO v^T,  dS = P o (dP - rowsum(dP o P)),  dq = dS k / sqrt(C),  dk = dS^T q / sqrt(C)
//
// 0.17 GFLOP per image - 0.4 % of the block's backward - so this first version favours obviously-correct structure over
// speed: every product is the SAME small batched kernel  C[m][n] = scale * sum_k A[m][k] B[n][k]  (both operands contiguous
// along the contraction: one 16-byte load per lane feeds v_mfma_f32_32x32x16_bf16), with explicit bf16 transposes in between
// and fp32 logits / dP in global memory.
#include "common.h"
#include "misc.h"

namespace {

// C[b][m][n] = scale * sum_k A[b][m][k] * Bm[b][n][k]; one wave per 32x32 tile; M, N % 32 == 0, K % 16 == 0.
// T: operand (and, with OUT_ACT, result) storage type; float operands run on the exact-fp32 MFMA (common.h mma16).
template <bool OUT_ACT, typename TA>
__global__ __launch_bounds__(64) void nt_gemm_kernel(const TA* __restrict__ A, const TA* __restrict__ Bm, void* __restrict__ C,
                                                      int M, int N, int K, float scale) {
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    // Workgroup L (x fastest) runs on XCD L % 8: with N / 32 = 8 every XCD would own one column of tiles of EVERY batch element and
    // fetch all of A (measured 4.5x the operands).  Re-deal so that the tiles of one batch element sit side by side on one XCD.
    const int tpb = (int)(gridDim.x * gridDim.y), T = tpb * (int)gridDim.z;
    const int L = (int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z);
    const int v = (L & 7) * (T >> 3) + min(L & 7, T & 7) + (L >> 3);
    const size_t b = v / tpb;
    const int tile = v - (int)b * tpb;
    const int n0 = (tile % (int)gridDim.x) * 32, m0 = (tile / (int)gridDim.x) * 32;
    const TA* ap = A + (b * M + m0 + m) * K + 8 * h;
    const TA* bp = Bm + (b * N + n0 + m) * K + 8 * h;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) mma16(acc, load_frag(ap + k0), load_frag(bp + k0));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const size_t o = (b * M + m0 + (i & 3) + 8 * (i >> 2) + 4 * h) * N + n0 + m;
        if (OUT_ACT)
            reinterpret_cast<TA*>(C)[o] = (TA)(acc[i] * scale);
        else
            reinterpret_cast<float*>(C)[o] = acc[i] * scale;
    }
}

// out[b][c][r] = in[b][r][c]   (bf16, R and Cc multiples of 32)
template <typename T>
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int Cc) {
    __shared__ T tile[32][34];
    const size_t b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) tile[i][tx] = in[(b * R + r0 + i) * Cc + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) out[(b * Cc + c0 + i) * R + r0 + tx] = tile[tx][i];
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float u = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, u) : v + u;
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

// P[row][k] = softmax_k(S[row][k] * sc); one workgroup (T threads) per row
template <typename TP>
__global__ void softmax_rows_kernel(const float* __restrict__ S, TP* __restrict__ P, int T, float sc) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const float s = S[row * T + threadIdx.x] * sc;
    const float mx = block_reduce(s, true, red);
    const float e = expf(s - mx);
    const float sum = block_reduce(e, false, red);
    P[row * T + threadIdx.x] = (TP)(e / sum);
}

// dS[row][k] = P (dP - sum_k dP P); one workgroup (T threads) per row
template <typename TP>
__global__ void attn_ds_kernel(const TP* __restrict__ P, const float* __restrict__ dP, TP* __restrict__ dS, int T) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const float p = (float)P[row * T + threadIdx.x], g = dP[row * T + threadIdx.x];
    const float d = block_reduce(p * g, false, red);
    dS[row * T + threadIdx.x] = (TP)(p * (g - d));
}

// dqkv[b][t][c*3 + plane] <- dq[b][t][c] | dk[b][t][c] | dvt[b][c][t]   (the reference's channel order, :160-163)
template <typename TP>
__global__ void qkv_interleave_kernel(const TP* __restrict__ dq, const TP* __restrict__ dk, const TP* __restrict__ dvt,
                                      TP* __restrict__ out, int T, int C, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t bt = i / C;
        const int t = (int)(bt % T);
        const int64_t b = bt / T;
        TP* o = out + (bt * C + c) * 3;
        o[0] = dq[i];
        o[1] = dk[i];
        o[2] = dvt[(b * C + c) * T + t];
    }
}

__global__ void add_inplace_f32_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) a[i] += b[i];
}
template <typename TP>
__global__ void add_to_bf16_kernel(const float* __restrict__ a, const float* __restrict__ b, TP* __restrict__ out, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) out[i] = (TP)(a[i] + b[i]);
}

// dtype: storage of the operands (1 bf16, 0 fp32)
template <bool OB>
void nt_gemm(int dtype, const void* A, const void* Bm, void* C, int batch, int M, int N, int K, float scale, hipStream_t s) {
    if (dtype)
        hipLaunchKernelGGL((nt_gemm_kernel<OB, __bf16>), dim3(N / 32, M / 32, batch), dim3(64), 0, s, (const __bf16*)A, (const __bf16*)Bm, C, M, N, K, scale);
    else
        hipLaunchKernelGGL((nt_gemm_kernel<OB, float>), dim3(N / 32, M / 32, batch), dim3(64), 0, s, (const float*)A, (const float*)Bm, C, M, N, K, scale);
}
void transpose(int dtype, const void* in, void* out, int batch, int R, int Cc, hipStream_t s) {
    if (dtype)
        hipLaunchKernelGGL(transpose_bf16_kernel<__bf16>, dim3(Cc / 32, R / 32, batch), dim3(256), 0, s, (const __bf16*)in, (__bf16*)out, R, Cc);
    else
        hipLaunchKernelGGL(transpose_bf16_kernel<float>, dim3(Cc / 32, R / 32, batch), dim3(256), 0, s, (const float*)in, (float*)out, R, Cc);
}
#define ATT_T(dtype, ...)     \
    do {                      \
        if (dtype) {          \
            typedef __bf16 TP; \
            __VA_ARGS__;      \
        } else {              \
            typedef float TP;  \
            __VA_ARGS__;      \
        }                     \
    } while (0)

}  // namespace

// bytes of scratch: 4 transposes [B][T][C] + 4 [B][T][T] in the storage type + 2 fp32 [B][T][T]
size_t attention_backward_scratch_bytes(int dtype, int B, int T, int C) {
    const size_t e = dtype ? 2 : 4;
    return (size_t)B * T * C * e * 4 + (size_t)B * T * T * (e * 4 + 4 * 2) + 4096;
}

// q, k, dO, dq, dk: [B][T][C]; vt, dvt: [B][C][T]; all in the storage type.  T in {64, 256}, C % 32 == 0.
int launch_attention_backward(int dtype, const void* q, const void* k, const void* vt, const void* dO, void* dq, void* dk, void* dvt,
                              void* scratch, int B, int T, int C, hipStream_t s) {
    if ((T != 64 && T != 256) || (C % 32)) return (int)hipErrorInvalidValue;
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        void* r = p;
        p += (bytes + 255) & ~(size_t)255;
        return r;
    };
    const size_t esz = dtype ? 2 : 4;
    const size_t tc = (size_t)B * T * C * esz, tt2 = (size_t)B * T * T * esz, tt4 = (size_t)B * T * T * 4;
    void *v = take(tc), *kT = take(tc), *qT = take(tc), *dOT = take(tc);
    void *P = take(tt2), *PT = take(tt2), *dS = take(tt2), *dST = take(tt2);
    float *S = (float*)take(tt4), *dP = (float*)take(tt4);
    const float sc = 1.0f / sqrtf((float)C);
    transpose(dtype, vt, v, B, C, T, s);   // v [T][C]
    transpose(dtype, k, kT, B, T, C, s);   // k^T [C][T]
    transpose(dtype, q, qT, B, T, C, s);
    transpose(dtype, dO, dOT, B, T, C, s);
    nt_gemm<false>(dtype, q, k, S, B, T, T, C, 1.0f, s);
    ATT_T(dtype, hipLaunchKernelGGL(softmax_rows_kernel<TP>, dim3((unsigned)((size_t)B * T)), dim3(T), 0, s, S, (TP*)P, T, sc));
    nt_gemm<false>(dtype, dO, v, dP, B, T, T, C, 1.0f, s);
    ATT_T(dtype, hipLaunchKernelGGL(attn_ds_kernel<TP>, dim3((unsigned)((size_t)B * T)), dim3(T), 0, s, (const TP*)P, dP, (TP*)dS, T));
    transpose(dtype, P, PT, B, T, T, s);
    transpose(dtype, dS, dST, B, T, T, s);
    nt_gemm<true>(dtype, dS, kT, dq, B, T, C, T, sc, s);    // dq[q][c] = sum_k dS[q][k] k[k][c]
    nt_gemm<true>(dtype, dST, qT, dk, B, T, C, T, sc, s);   // dk[k][c] = sum_q dS[q][k] q[q][c]
    nt_gemm<true>(dtype, dOT, PT, dvt, B, C, T, T, 1.0f, s);  // dv^T[c][k] = sum_q dO[q][c] P[q][k]
    return (int)hipGetLastError();
}

// Forward-mode derivative of the attention (the reference's AttentionOp.jvp, EDM/network.py:186-196, followed by the tangent of
// the value product): Sd = (qd k^T + q kd^T) / sqrt(C), Pd = P o (Sd - rowsum(P o Sd)), od = Pd v + P vd.
// q, k, qd, kd, od: [B][T][C]; vt, vtd: [B][C][T]; bf16.  Same scratch size as the backward.
int launch_attention_jvp(int dtype, const void* q, const void* k, const void* vt, const void* qd, const void* kd, const void* vtd, void* od,
                         void* scratch, int B, int T, int C, hipStream_t s) {
    if ((T != 64 && T != 256) || (C % 32)) return (int)hipErrorInvalidValue;
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        void* r = p;
        p += (bytes + 255) & ~(size_t)255;
        return r;
    };
    const size_t esz = dtype ? 2 : 4;
    const size_t tc = (size_t)B * T * C * esz, tt2 = (size_t)B * T * T * esz, tt4 = (size_t)B * T * T * 4;
    void *t0 = take(tc), *t1 = take(tc), *t2 = take(tc), *t3 = take(tc);  // scratch sized like the backward's 4 transposes
    void *P = take(tt2), *Pd = take(tt2), *u0 = take(tt2), *u1 = take(tt2);
    float *S = (float*)take(tt4), *Sd = (float*)take(tt4);
    (void)u0, (void)u1;
    const float sc = 1.0f / sqrtf((float)C);
    nt_gemm<false>(dtype, q, k, S, B, T, T, C, 1.0f, s);
    ATT_T(dtype, hipLaunchKernelGGL(softmax_rows_kernel<TP>, dim3((unsigned)((size_t)B * T)), dim3(T), 0, s, S, (TP*)P, T, sc));
    // Sd: two products into the two fp32 buffers, summed by the (reused) dS kernel's input: S <- qd k^T, Sd <- q kd^T
    nt_gemm<false>(dtype, qd, k, S, B, T, T, C, sc, s);
    nt_gemm<false>(dtype, q, kd, Sd, B, T, T, C, sc, s);
    {
        const int64_t total = (int64_t)B * T * T;
        const int64_t blocks = (total + 255) / 256;
        hipLaunchKernelGGL(add_inplace_f32_kernel, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, s, Sd, S, total);
    }
    ATT_T(dtype, hipLaunchKernelGGL(attn_ds_kernel<TP>, dim3((unsigned)((size_t)B * T)), dim3(T), 0, s, (const TP*)P, Sd, (TP*)Pd, T));
    // od[q][c] = sum_k Pd[q][k] v[k][c] + P[q][k] vd[k][c]: B operands are vt / vtd ([c][k], contraction-contiguous) as they lie
    float* o0 = S;   // [B][T][C] fp32 fits in the [B][T][T] buffers when C <= T; otherwise use the transposes' space
    float* o1 = Sd;
    if (C > T) {
        o0 = (float*)t0;  // 4 * tc bytes = 2 fp32 [B][T][C] tensors
        o1 = (float*)t2;
    }
    nt_gemm<false>(dtype, Pd, vt, o0, B, T, C, T, 1.0f, s);
    nt_gemm<false>(dtype, P, vtd, o1, B, T, C, T, 1.0f, s);
    {
        const int64_t total = (int64_t)B * T * C;
        const int64_t blocks = (total + 255) / 256;
        ATT_T(dtype, hipLaunchKernelGGL(add_to_bf16_kernel<TP>, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, s, o0, o1, (TP*)od, total));
    }
    (void)t1, (void)t3;
    return (int)hipGetLastError();
}

int launch_qkv_interleave(int dtype, const void* dq, const void* dk, const void* dvt, void* out, int B, int T, int C, hipStream_t s) {
    const int64_t total = (int64_t)B * T * C;
    const int64_t blocks = (total + 255) / 256;
    ATT_T(dtype, hipLaunchKernelGGL(qkv_interleave_kernel<TP>, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, s, (const TP*)dq,
                                    (const TP*)dk, (const TP*)dvt, (TP*)out, T, C, total));
    return (int)hipGetLastError();
}

// the same two helpers for other small dense layers (disc.hip): batch = 1 matrices, all dimensions multiples of 32 (K: 16)
int launch_nt_gemm(const void* A, const void* Bm, void* C, int M, int N, int K, float scale, int out_bf16, hipStream_t s) {
    if ((M % 32) || (N % 32) || (K % 16)) return (int)hipErrorInvalidValue;
    if (out_bf16)
        nt_gemm<true>(1, A, Bm, C, 1, M, N, K, scale, s);
    else
        nt_gemm<false>(1, A, Bm, C, 1, M, N, K, scale, s);
    return (int)hipGetLastError();
}
int launch_transpose_bf16(const void* in, void* out, int R, int Cc, hipStream_t s) {
    if ((R % 32) || (Cc % 32)) return (int)hipErrorInvalidValue;
    transpose(1, in, out, 1, R, Cc, s);
    return (int)hipGetLastError();
}

