// DiT block pieces that are not GEMMs (reference fastgen/networks/DiT/network.py).  The four GEMMs of a block
// (qkv, proj, fc1, fc2) run on conv_fused_kernel's token modes (conv.hip OUT_TOK / OUT_HEADS: 128 tokens x 128 columns per
// workgroup, bias / GELU(tanh) / adaLN gate / residual epilogues, head-split q | k | v^T outputs).  Here:
//   ln_modulate_kernel     LayerNorm(eps 1e-6, no affine) + apply_adaptive_modulation (:29-41, 187-189, 195-196)
//   dit_attention_kernel   multi-head self-attention, 256 tokens, head_dim 64 / 72 (timm Attention as DiTBlock uses it, :168, 191)
//   patch_embed_kernel     PatchEmbed conv(kernel = stride = patch) + bias + pos_embed (:270, 511)
//   fourier_kernel, cond_kernel   FourierTimeEmbedding features (:67-96), c = t_emb + y_emb + r_emb and silu(c) (:514-533)
//   final_kernel           OutputProjection (adaLN + Linear to patch pixels, :218-225) + unpatchify (:437-455)
// T is the compute type of common.h (float = exact fp32, __bf16, bf16x3 = fp32 tensors with split-bf16 products); tensors are in
// its storage type DT<T>::ST.
#include <type_traits>

#include "common.h"
#include "misc.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ f32x2 ld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
__device__ __forceinline__ f32x2 ld2(const __bf16* p) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 q = *reinterpret_cast<const bf16x2*>(p);
    return f32x2{(float)q[0], (float)q[1]};
}
__device__ __forceinline__ void st2(float* p, f32x2 v) { *reinterpret_cast<f32x2*>(p) = v; }
__device__ __forceinline__ void st2(__bf16* p, f32x2 v) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    *reinterpret_cast<bf16x2*>(p) = bf16x2{(__bf16)v[0], (__bf16)v[1]};
}

// One wave per token: lane l holds the channel pairs 2l + 128j.  LayerNorm statistics in two passes over the registers (fp32).
template <typename ST, int D>
__device__ __forceinline__ void ln_token(const ST* __restrict__ xrow, int lane, float (&v)[D / 64]) {
    constexpr int NP = D / 128;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const f32x2 q = ld2(xrow + 2 * lane + 128 * j);
        v[2 * j] = q[0], v[2 * j + 1] = q[1];
        s += q[0] + q[1];
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < D / 64; ++e) {
        v[e] -= mean;
        ss = fmaf(v[e], v[e], ss);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * (1.0f / D) + 1e-6f);
#pragma unroll
    for (int e = 0; e < D / 64; ++e) v[e] *= rstd;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const __bf16* p) {
    const bf16x4 q = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(__bf16* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; }

// y[tok][c] = LN(x[tok])[c] * (1 + scale[n][c]) + shift[n][c];  mod [B][mod_stride] holds shift at shift_off, scale at scale_off.
// One wave walks FOUR consecutive tokens; lane l holds the channel quads 4 l + 256 j.  The kernel is bound by the NUMBER of its
// vector-memory instructions, not by their bytes (a wave issues one per ~65-150 cycles whatever its width, MI355X_MICROARCH.md /
// the cycle stamps of wgrad.hip and gemm.hip): 4-channel pieces (8 / 16 bytes per lane) instead of pairs, and the modulation
// vectors - the same for every token of an image or frame - loaded once per wave instead of once per token: 36 -> 12.5 instructions
// per token at D = 1152 (DiT-XL/2, B = 256: 72 -> 62 us = 4.9 TB/s).
template <typename ST, int D>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const ST* __restrict__ x, const float* __restrict__ mod, int mod_stride,
                                                          int shift_off, int scale_off, ST* __restrict__ y, int ntok, int tpi) {
    constexpr int NJ = (D + 255) / 256, TPW = 4;
    const int lane = threadIdx.x & 63;
    const int tok0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * TPW;
    if (tok0 >= ntok) return;
    // (the last quad column is ragged when D % 256 != 0: 1152 = 4.5 x 256)
    auto has = [&](int j) { return j < NJ - 1 || D % 256 == 0 || lane < (D % 256) / 4; };
    f32x4 sc[NJ], sh[NJ];
    int n_have = -1;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tok = tok0 + i;
        if (tok >= ntok) break;
        const int n = tok / tpi;
        if (n != n_have) {  // (wave-uniform)
            const float* mrow = mod + (size_t)n * mod_stride;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (has(j)) {
                    sc[j] = ld4(mrow + scale_off + 4 * lane + 256 * j) + 1.0f;
                    sh[j] = ld4(mrow + shift_off + 4 * lane + 256 * j);
                }
            n_have = n;
        }
        const ST* xrow = x + (size_t)tok * D;
        f32x4 v[NJ];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (has(j)) v[j] = ld4(xrow + 4 * lane + 256 * j);
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mean = wave_sum(s) * (1.0f / D);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (has(j)) {
                v[j] -= mean;
                ss = fmaf(v[j][0], v[j][0], ss), ss = fmaf(v[j][1], v[j][1], ss), ss = fmaf(v[j][2], v[j][2], ss), ss = fmaf(v[j][3], v[j][3], ss);
            }
        const float rstd = 1.0f / sqrtf(wave_sum(ss) * (1.0f / D) + 1e-6f);
        ST* yrow = y + (size_t)tok * D;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (has(j)) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(v[j][e] * rstd, sc[j][e], sh[j][e]);
                st4(yrow + 4 * lane + 256 * j, o);
            }
    }
}

// the same with y written as [hi | lo] bf16 planes ([ntok][2 D]: the A operand of the split-bf16 token GEMM, gemm.hip launch_gemm_x3)
template <int D>
__global__ __launch_bounds__(256) void ln_modulate_split_kernel(const float* __restrict__ x, const float* __restrict__ mod, int mod_stride,
                                                                int shift_off, int scale_off, __bf16* __restrict__ y, int ntok, int tpi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= ntok) return;
    const int n = tok / tpi;
    float v[D / 64];
    ln_token<float, D>(x + (size_t)tok * D, lane, v);
    const float* mrow = mod + (size_t)n * mod_stride;
#pragma unroll
    for (int j = 0; j < D / 128; ++j) {
        const int c = 2 * lane + 128 * j;
        const f32x2 sc = ld2(mrow + scale_off + c), sh = ld2(mrow + shift_off + c);
        const float a = fmaf(v[2 * j], 1.0f + sc[0], sh[0]), b = fmaf(v[2 * j + 1], 1.0f + sc[1], sh[1]);
        const bf16x2 hi = {(__bf16)a, (__bf16)b};
        const bf16x2 lo = {(__bf16)(a - (float)hi[0]), (__bf16)(b - (float)hi[1])};
        *reinterpret_cast<bf16x2*>(y + (size_t)tok * 2 * D + c) = hi;
        *reinterpret_cast<bf16x2*>(y + (size_t)tok * 2 * D + D + c) = lo;
    }
}

// OutputProjection + unpatchify: out[n][c][gy p + py][gx p + px] = b[o] + sum_d W[o][d] y[d],  o = (py p + px) C + c,
// y = LN(x) (1 + scale) + shift with {shift, scale} = mod[n][0:D], mod[n][D:2D]  (chunk order of :220)
template <typename ST, int D>
__global__ __launch_bounds__(256) void final_kernel(const ST* __restrict__ x, const float* __restrict__ mod, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ out, int ntok, int grid, int p, int C) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= ntok) return;
    const int tpi = grid * grid, n = tok / tpi, t = tok - n * tpi;
    float v[D / 64];
    ln_token<ST, D>(x + (size_t)tok * D, lane, v);
    const float* mrow = mod + (size_t)n * 2 * D;
#pragma unroll
    for (int j = 0; j < D / 128; ++j) {
        const int c = 2 * lane + 128 * j;
        const f32x2 sh = ld2(mrow + c), sc = ld2(mrow + D + c);
        v[2 * j] = fmaf(v[2 * j], 1.0f + sc[0], sh[0]);
        v[2 * j + 1] = fmaf(v[2 * j + 1], 1.0f + sc[1], sh[1]);
    }
    const int PO = p * p * C;
    const int gy = t / grid, gx = t - gy * grid, res = grid * p;
    for (int o = 0; o < PO; ++o) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < D / 128; ++j) {
            const f32x2 q = ld2(w + (size_t)o * D + 2 * lane + 128 * j);
            a = fmaf(v[2 * j], q[0], a);
            a = fmaf(v[2 * j + 1], q[1], a);
        }
        a = wave_sum(a);
        if (lane == 0) {
            const int c = o % C, pq = o / C, py = pq / p, px = pq - py * p;
            out[(((size_t)n * C + c) * res + gy * p + py) * res + gx * p + px] = a + bias[o];
        }
    }
}

// x0[n][t][d] = bias[d] + pos[t][d] + sum_{c,py,px} W[d][c][py][px] x[n][c][gy p + py][gx p + px]
// One thread = output dim d of FOUR neighbouring tokens (one patch row segment): its weight row (K = C p p floats, 16-byte loads,
// consecutive threads = consecutive rows: coalesced) is read once for the four; the tokens' inputs are wave-uniform (broadcast) loads.
template <typename ST, int K>
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ pos, ST* __restrict__ out, int B, int C, int grid, int p,
                                                          int D) {
    const int tpi = grid * grid, res = grid * p;
    const int64_t total = (int64_t)B * (tpi / 4) * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int d = (int)(i % D);
        const int64_t tg = i / D;
        const int t0 = (int)(tg % (tpi / 4)) * 4, n = (int)(tg / (tpi / 4));
        const int gy = t0 / grid, gx0 = t0 - gy * grid;  // grid % 4 == 0: the four tokens share a patch row
        float wr[K];
#pragma unroll
        for (int k = 0; k < K; k += 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(w + (size_t)d * K + k);
            wr[k] = q[0], wr[k + 1] = q[1], wr[k + 2] = q[2], wr[k + 3] = q[3];
        }
        const float bd = bias[d];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = bd + pos[(size_t)(t0 + j) * D + d];
            int k = 0;
            for (int c = 0; c < C; ++c)
                for (int py = 0; py < p; ++py)
                    for (int px = 0; px < p; ++px, ++k)
                        a = fmaf(wr[k], x[(((size_t)n * C + c) * res + gy * p + py) * res + (gx0 + j) * p + px], a);
            out[((size_t)n * tpi + t0 + j) * D + d] = (ST)a;
        }
    }
}
// The reference's configs (4 latent channels, patch 2; D % 64 == 0): a wave's 64 threads are 64 output dims of the SAME four tokens, so the
// tokens' 64 inputs are scalar loads (8 consecutive floats per (channel, patch row)) and the 64 multiply-adds are straight-line code -
// the generic form above walks (c, py, px) in runtime loops with an address computation and a vector load per term (559 us at B = 256).
template <typename ST>
__global__ __launch_bounds__(256) void patch_embed_c4p2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, const float* __restrict__ pos,
                                                               ST* __restrict__ out, int B, int grid, int D) {
    const int tpi = grid * grid, res = grid * 2;
    const int64_t total = (int64_t)B * (tpi / 4) * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int tg = __builtin_amdgcn_readfirstlane((int)(i / D));  // (uniform per wave: D % 64 == 0, waves start at multiples of 64)
        const int d = (int)(i - (int64_t)tg * D);
        const int t0 = (tg % (tpi / 4)) * 4, n = tg / (tpi / 4);
        const int gy = t0 / grid, gx0 = t0 - gy * grid;
        float wr[16];
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(w + (size_t)d * 16 + k);
            wr[k] = q[0], wr[k + 1] = q[1], wr[k + 2] = q[2], wr[k + 3] = q[3];
        }
        float xs[4][2][8];  // [channel][patch row][2 j + px]
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const float* row = x + (((size_t)n * 4 + c) * res + gy * 2 + py) * res + gx0 * 2;
#pragma unroll
                for (int e = 0; e < 8; ++e) xs[c][py][e] = row[e];
            }
        const float bd = bias[d];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = bd + pos[(size_t)(t0 + j) * D + d];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int px = 0; px < 2; ++px) a = fmaf(wr[(c * 2 + py) * 2 + px], xs[c][py][2 * j + px], a);
            out[((size_t)n * tpi + t0 + j) * D + d] = (ST)a;
        }
    }
}
// (any other K: one thread per output element)
template <typename ST>
__global__ __launch_bounds__(256) void patch_embed_any_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ pos, ST* __restrict__ out, int B, int C, int grid, int p,
                                                              int D) {
    const int tpi = grid * grid, res = grid * p, K = C * p * p;
    const int64_t total = (int64_t)B * tpi * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int d = (int)(i % D);
        const int64_t tok = i / D;
        const int t = (int)(tok % tpi), n = (int)(tok / tpi);
        const int gy = t / grid, gx = t - gy * grid;
        float a = bias[d] + pos[(size_t)t * D + d];
        const float* wr = w + (size_t)d * K;
        for (int c = 0; c < C; ++c)
            for (int py = 0; py < p; ++py)
                for (int px = 0; px < p; ++px)
                    a = fmaf(wr[(c * p + py) * p + px], x[(((size_t)n * C + c) * res + gy * p + py) * res + gx * p + px], a);
        out[i] = (ST)a;
    }
}

// f[b][j] = cos(t f_j) (j < half) | sin(t f_{j - half}),  f_j = exp(-ln(10000) j / half) evaluated in fp32 as torch does
__global__ void fourier_kernel(const float* __restrict__ t, float* __restrict__ f, int B, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, j = i - b * dim, half = dim / 2, jj = j < half ? j : j - half;
    const float freq = expf(-9.210340371976184f * (float)jj / (float)half);
    const float ang = t[b] * freq;
    f[i] = j < half ? cosf(ang) : sinf(ang);
}

// c = t_emb + table[cls] + r_emb (r_emb nullable);  sc = silu(c).  A class index outside the table's `rows` rows (the reference's
// nn.Embedding device-asserts there) reads nothing: the sample's conditioning becomes NaN and *err (host-visible) is raised - the
// handle's next call returns FG_EINVAL.
__global__ void cond_kernel(const float* __restrict__ t_emb, const float* __restrict__ r_emb, const float* __restrict__ table,
                            const int64_t* __restrict__ cls, float* __restrict__ c, float* __restrict__ sc, int B, int D, int rows,
                            int* __restrict__ err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i - b * D;
    const int64_t id = cls[b];
    const bool ok = id >= 0 && id < rows;
    if (!ok && d == 0 && err) *err = 1;
    float v = ok ? t_emb[i] + table[(size_t)id * D + d] : __builtin_nanf("");
    if (r_emb) v += r_emb[i];
    c[i] = v;
    sc[i] = v / (1.0f + expf(-v));
}

// ---- attention ----------------------------------------------------------------------------------------------------------
// One wave = 32 queries of one (image, head), the structure of attn.hip's attention_kernel: S^T = K Q^T with the key on the
// accumulator rows and the query on the lane (in-register softmax), the P^T accumulators feeding O = P V directly.
// q, k: [B][H][T][HD]; vt: [B][H][HD][T]; out: [B][T][H * HD] (the layout `x.transpose(1, 2).reshape(B, N, C)` of the restated
// timm Attention hands to proj).  HD = 64 or 72: the 16-deep MFMA steps cover ceil(HD / 16) * 16 dims, fragments past HD are zero;
// the 32-wide output tiles cover ceil(HD / 32) * 32 dims, columns past HD are computed from rows of the next head (finite, the
// caller leaves 32 rows of slack behind the last head) and never stored.
template <typename T>
struct DPFrag;
template <>
struct DPFrag<__bf16> {
    static __device__ __forceinline__ Frag8<__bf16> make(const f32x16& p, int s) {
        Frag8<__bf16> f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)p[8 * s + j];
        return f;
    }
};
template <>
struct DPFrag<float> {
    static __device__ __forceinline__ Frag8<float> make(const f32x16& p, int s) {
        Frag8<float> f;
        f.lo = f32x4{p[8 * s + 0], p[8 * s + 1], p[8 * s + 2], p[8 * s + 3]};
        f.hi = f32x4{p[8 * s + 4], p[8 * s + 5], p[8 * s + 6], p[8 * s + 7]};
        return f;
    }
};
template <>
struct DPFrag<bf16x3> {
    static __device__ __forceinline__ Frag8<bf16x3> make(const f32x16& p, int s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[8 * s + j];
        Frag8<bf16x3> f;
        split8(v, f.hi, f.lo);
        return f;
    }
};
// 8 consecutive elements (all valid or all past the end) -> operand fragment
template <typename T>
__device__ __forceinline__ Frag8<T> dload(const typename DT<T>::ST* p, bool valid) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (valid) widen8(load_frag(p), v);
    if constexpr (std::is_same<T, bf16x3>::value) {
        Frag8<bf16x3> f;
        split8(v, f.hi, f.lo);
        return f;
    } else if constexpr (std::is_same<T, __bf16>::value) {
        Frag8<__bf16> f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)v[j];
        return f;
    } else {
        Frag8<float> f;
        f.lo = f32x4{v[0], v[1], v[2], v[3]};
        f.hi = f32x4{v[4], v[5], v[6], v[7]};
        return f;
    }
}
// V^T fragment: 4 keys at key0 and 4 at key0 + 8 of one output-dim row (the K order of an accumulator tile used as an operand)
template <typename T>
__device__ __forceinline__ Frag8<T> dload_v(const typename DT<T>::ST* row, int key0) {
    float v[8];
    const f32x4 a = load4(row + key0), b = load4(row + key0 + 8);
    v[0] = a[0], v[1] = a[1], v[2] = a[2], v[3] = a[3], v[4] = b[0], v[5] = b[1], v[6] = b[2], v[7] = b[3];
    if constexpr (std::is_same<T, bf16x3>::value) {
        Frag8<bf16x3> f;
        split8(v, f.hi, f.lo);
        return f;
    } else if constexpr (std::is_same<T, __bf16>::value) {
        Frag8<__bf16> f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)v[j];
        return f;
    } else {
        Frag8<float> f;
        f.lo = a;
        f.hi = b;
        return f;
    }
}

// bf16x3: q, k, v^T arrive as hi and lo bf16 planes (the qkv projection's epilogue wrote them split: conv.hip store4_split)
struct QKVPlanes {
    size_t lo_off;  // elements from a hi plane to its lo plane
};
__device__ __forceinline__ Frag8<bf16x3> dload_planes(const __bf16* p, size_t lo_off, bool valid) {
    Frag8<bf16x3> f;
    f.hi = f.lo = bf16x8{};
    if (valid) {
        f.hi = *reinterpret_cast<const bf16x8*>(p);
        f.lo = *reinterpret_cast<const bf16x8*>(p + lo_off);
    }
    return f;
}
__device__ __forceinline__ Frag8<bf16x3> dload_v_planes(const __bf16* row, size_t lo_off, int key0) {
    Frag8<bf16x3> f;
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(row + key0), b = *reinterpret_cast<const bf16x4*>(row + key0 + 8);
    const bf16x4 c = *reinterpret_cast<const bf16x4*>(row + lo_off + key0), d = *reinterpret_cast<const bf16x4*>(row + lo_off + key0 + 8);
    f.hi = bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    f.lo = bf16x8{c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
    return f;
}

// QT: element type of q / k / v^T as stored (bf16x3: __bf16 planes; otherwise the storage type)
template <typename T, int HD>
__global__ __launch_bounds__(256, 2) void dit_attention_kernel(const typename DT<T>::WT* __restrict__ q, const typename DT<T>::WT* __restrict__ k,
                                                            const typename DT<T>::WT* __restrict__ vt, typename DT<T>::ST* __restrict__ out,
                                                            int BH, int heads, size_t lo_off) {
    typedef typename DT<T>::WT QT;
    constexpr bool X3 = std::is_same<T, bf16x3>::value;
    constexpr int Tn = 256, NT = Tn / 32;
    constexpr int KS = (HD + 15) / 16;  // 16-deep steps of the q k^T contraction
    constexpr int DT_ = (HD + 31) / 32;  // 32-wide tiles of the output dims
    constexpr bool FAST = DT<T>::FAST;
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);  // (image * heads + head, 32-query slab)
    const int bh = gw / NT, q0 = (gw % NT) * 32;
    if (bh >= BH) return;  // wave-uniform
    const int n = bh / heads, hh = bh - n * heads;

    const QT* qrow = q + ((size_t)bh * Tn + q0 + r) * HD + 8 * h;
    const QT* kbase = k + ((size_t)bh * Tn + r) * HD + 8 * h;
    f32x16 st[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[kt][i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const bool valid = kk * 16 + 8 * h < HD;  // HD % 8 == 0: an 8-element fragment is valid or past the end as a whole
        Frag8<T> qf;
        if constexpr (X3) qf = dload_planes(qrow + kk * 16, lo_off, valid);
        else qf = dload<T>(qrow + kk * 16, valid);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            Frag8<T> kf;
            if constexpr (X3) kf = dload_planes(kbase + (size_t)kt * 32 * HD + kk * 16, lo_off, valid);
            else kf = dload<T>(kbase + (size_t)kt * 32 * HD + kk * 16, valid);
            mma16(st[kt], kf, qf);
        }
    }
    // softmax over keys: the other half of this query's logits sits in lane ^ 32
    const float sc = 1.0f / sqrtf((float)HD);
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) m = fmaxf(m, st[kt][i]);
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float z = (st[kt][i] - m) * sc;
            const float e = FAST ? __builtin_amdgcn_exp2f(1.44269504088896341f * z) : expf(z);
            st[kt][i] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32);
    const float inv = FAST ? __builtin_amdgcn_rcpf(sum) : 1.0f / sum;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[kt][i] = FAST ? st[kt][i] * inv : st[kt][i] / sum;

    // O[query][dim] = sum_key P[query][key] V[key][dim]
    typedef typename DT<T>::ST ST;
    const QT* vbase = vt + ((size_t)bh * HD + r) * Tn + 4 * h;
    ST* obase = out + ((size_t)n * Tn + q0) * (heads * HD) + hh * HD + r;
    f32x16 o[DT_];
#pragma unroll
    for (int d = 0; d < DT_; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const Frag8<T> pf = DPFrag<T>::make(st[kt], s);
#pragma unroll
            for (int d = 0; d < DT_; ++d) {
                Frag8<T> vf;
                if constexpr (X3) vf = dload_v_planes(vbase + (size_t)(d * 32) * Tn, lo_off, kt * 32 + 16 * s);
                else vf = dload_v<T>(vbase + (size_t)(d * 32) * Tn, kt * 32 + 16 * s);
                mma16(o[d], pf, vf);
            }
        }
#pragma unroll
    for (int d = 0; d < DT_; ++d)
        if (d * 32 + r < HD) {
#pragma unroll
            for (int i = 0; i < 16; ++i) obase[(size_t)acc_row(i, h) * (heads * HD) + d * 32] = (ST)o[d][i];
        }
}

#define DIT_RET() return (int)hipGetLastError()

template <typename ST>
int ln_mod_d(int D, const void* x, const float* mod, int ms, int so, int co, void* y, int ntok, int tpi, hipStream_t s) {
    if ((ms % 4) || (so % 4) || (co % 4)) return (int)hipErrorInvalidValue;  // (16-byte loads of the modulation vectors)
    dim3 g((ntok + 15) / 16), b(256);  // 4 waves x 4 tokens
    switch (D) {
        case 384: hipLaunchKernelGGL((ln_modulate_kernel<ST, 384>), g, b, 0, s, (const ST*)x, mod, ms, so, co, (ST*)y, ntok, tpi); break;
        case 768: hipLaunchKernelGGL((ln_modulate_kernel<ST, 768>), g, b, 0, s, (const ST*)x, mod, ms, so, co, (ST*)y, ntok, tpi); break;
        case 1024: hipLaunchKernelGGL((ln_modulate_kernel<ST, 1024>), g, b, 0, s, (const ST*)x, mod, ms, so, co, (ST*)y, ntok, tpi); break;
        case 1152: hipLaunchKernelGGL((ln_modulate_kernel<ST, 1152>), g, b, 0, s, (const ST*)x, mod, ms, so, co, (ST*)y, ntok, tpi); break;
        // the causal video DiT's widths (engine_wan.inc)
        case 256: hipLaunchKernelGGL((ln_modulate_kernel<ST, 256>), g, b, 0, s, (const ST*)x, mod, ms, so, co, (ST*)y, ntok, tpi); break;
        case 1536: hipLaunchKernelGGL((ln_modulate_kernel<ST, 1536>), g, b, 0, s, (const ST*)x, mod, ms, so, co, (ST*)y, ntok, tpi); break;
        case 2048: hipLaunchKernelGGL((ln_modulate_kernel<ST, 2048>), g, b, 0, s, (const ST*)x, mod, ms, so, co, (ST*)y, ntok, tpi); break;
        case 5120: hipLaunchKernelGGL((ln_modulate_kernel<ST, 5120>), g, b, 0, s, (const ST*)x, mod, ms, so, co, (ST*)y, ntok, tpi); break;
        default: return (int)hipErrorInvalidValue;
    }
    DIT_RET();
}
template <typename ST>
int final_d(int D, const void* x, const float* mod, const float* w, const float* bias, float* out, int ntok, int grid, int p, int C, hipStream_t s) {
    dim3 g((ntok + 3) / 4), b(256);
    switch (D) {
        case 384: hipLaunchKernelGGL((final_kernel<ST, 384>), g, b, 0, s, (const ST*)x, mod, w, bias, out, ntok, grid, p, C); break;
        case 768: hipLaunchKernelGGL((final_kernel<ST, 768>), g, b, 0, s, (const ST*)x, mod, w, bias, out, ntok, grid, p, C); break;
        case 1024: hipLaunchKernelGGL((final_kernel<ST, 1024>), g, b, 0, s, (const ST*)x, mod, w, bias, out, ntok, grid, p, C); break;
        case 1152: hipLaunchKernelGGL((final_kernel<ST, 1152>), g, b, 0, s, (const ST*)x, mod, w, bias, out, ntok, grid, p, C); break;
        default: return (int)hipErrorInvalidValue;
    }
    DIT_RET();
}

}  // namespace

int launch_dit_ln_modulate_split(int D, const float* x, const float* mod, int mod_stride, int shift_off, int scale_off, void* y, int ntok,
                                 int tokens_per_image, hipStream_t s) {
    dim3 g((ntok + 3) / 4), b(256);
#define LNS(DD) hipLaunchKernelGGL((ln_modulate_split_kernel<DD>), g, b, 0, s, x, mod, mod_stride, shift_off, scale_off, (__bf16*)y, ntok, tokens_per_image)
    switch (D) {
        case 384: LNS(384); break;
        case 768: LNS(768); break;
        case 1024: LNS(1024); break;
        case 1152: LNS(1152); break;
        default: return (int)hipErrorInvalidValue;
    }
#undef LNS
    DIT_RET();
}
// dtype: storage of the token tensors (1 bf16, 0 fp32)
int launch_dit_ln_modulate(int dtype, int D, const void* x, const float* mod, int mod_stride, int shift_off, int scale_off, void* y,
                           int ntok, int tokens_per_image, hipStream_t s) {
    return dtype ? ln_mod_d<__bf16>(D, x, mod, mod_stride, shift_off, scale_off, y, ntok, tokens_per_image, s)
                 : ln_mod_d<float>(D, x, mod, mod_stride, shift_off, scale_off, y, ntok, tokens_per_image, s);
}
int launch_dit_final(int dtype, int D, const void* x, const float* mod, const float* w, const float* bias, float* out, int ntok, int grid,
                     int p, int C, hipStream_t s) {
    return dtype ? final_d<__bf16>(D, x, mod, w, bias, out, ntok, grid, p, C, s) : final_d<float>(D, x, mod, w, bias, out, ntok, grid, p, C, s);
}
int launch_dit_patch_embed(int dtype, const float* x, const float* w, const float* bias, const float* pos, void* out, int B, int C, int grid,
                           int p, int D, hipStream_t s) {
    const bool k16 = C * p * p == 16 && p == 2 && (grid % 4) == 0;  // the reference's configs: 4 latent channels, patch 2
    const int64_t total = k16 ? (int64_t)B * (grid * grid / 4) * D : (int64_t)B * grid * grid * D;
    const int64_t blocks = (total + 255) / 256;
    dim3 g((unsigned)(blocks > 262144 ? 262144 : blocks));
    if (k16 && C == 4 && (D % 64) == 0) {
        if (dtype) hipLaunchKernelGGL(patch_embed_c4p2_kernel<__bf16>, g, dim3(256), 0, s, x, w, bias, pos, (__bf16*)out, B, grid, D);
        else hipLaunchKernelGGL(patch_embed_c4p2_kernel<float>, g, dim3(256), 0, s, x, w, bias, pos, (float*)out, B, grid, D);
    } else if (k16) {
        if (dtype) hipLaunchKernelGGL((patch_embed_kernel<__bf16, 16>), g, dim3(256), 0, s, x, w, bias, pos, (__bf16*)out, B, C, grid, p, D);
        else hipLaunchKernelGGL((patch_embed_kernel<float, 16>), g, dim3(256), 0, s, x, w, bias, pos, (float*)out, B, C, grid, p, D);
    } else {
        if (dtype) hipLaunchKernelGGL(patch_embed_any_kernel<__bf16>, g, dim3(256), 0, s, x, w, bias, pos, (__bf16*)out, B, C, grid, p, D);
        else hipLaunchKernelGGL(patch_embed_any_kernel<float>, g, dim3(256), 0, s, x, w, bias, pos, (float*)out, B, C, grid, p, D);
    }
    DIT_RET();
}
int launch_dit_fourier(const float* t, float* f, int B, int dim, hipStream_t s) {
    hipLaunchKernelGGL(fourier_kernel, dim3((B * dim + 255) / 256), dim3(256), 0, s, t, f, B, dim);
    DIT_RET();
}
int launch_dit_cond(const float* t_emb, const float* r_emb, const float* table, const int64_t* cls, float* c, float* sc, int B, int D,
                    int rows, int* err, hipStream_t s) {
    hipLaunchKernelGGL(cond_kernel, dim3((B * D + 255) / 256), dim3(256), 0, s, t_emb, r_emb, table, cls, c, sc, B, D, rows, err);
    DIT_RET();
}
// mode: FG_DTYPE_* (0 exact fp32, 1 bf16, 2 split-bf16 on fp32 tensors).  256 tokens; head_dim 64 or 72.
int launch_dit_attention(int mode, const void* q, const void* k, const void* vt, void* out, int B, int heads, int head_dim, size_t lo_off,
                         hipStream_t s) {
    const int BH = B * heads;
    dim3 g((BH * 8 + 3) / 4), b(256);
#define DIT_ATT(TT, HD) hipLaunchKernelGGL((dit_attention_kernel<TT, HD>), g, b, 0, s, (const typename DT<TT>::WT*)q, (const typename DT<TT>::WT*)k, \
                                           (const typename DT<TT>::WT*)vt, (typename DT<TT>::ST*)out, BH, heads, lo_off)
    if (head_dim == 72) {
        if (mode == 2) DIT_ATT(bf16x3, 72);
        else if (mode == 1) DIT_ATT(__bf16, 72);
        else DIT_ATT(float, 72);
    } else if (head_dim == 64) {
        if (mode == 2) DIT_ATT(bf16x3, 64);
        else if (mode == 1) DIT_ATT(__bf16, 64);
        else DIT_ATT(float, 64);
    } else {
        return (int)hipErrorInvalidValue;
    }
#undef DIT_ATT
    DIT_RET();
}
