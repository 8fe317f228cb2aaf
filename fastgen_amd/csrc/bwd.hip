// Elementwise / reduction kernels of the U-Net block backward pass (the DMD2 training step differentiates
// `UNetBlock.forward`, fastgen/networks/EDM/network.py:274-299, by autograd; these are its hand-written pieces).
// Activations and activation gradients are NHWC bf16 (the training compute dtype), statistics and parameter gradients fp32.
//
// GroupNorm + SiLU, forward:  y = a x + b  with a = rstd*gamma, b = beta - mean*rstd*gamma  (per image and channel),
//                             act = silu(y)                                              (EDM/network.py:141-149, 276, 283)
// backward, given dact:       dy = dact * silu'(y),  silu'(y) = s (1 + y (1 - s)),  s = sigmoid(y)
//                             P1[n,c] = sum_p dy,  P2[n,c] = sum_p dy * xhat,        xhat = (x - mean) rstd
//                             dgamma[c] += sum_n P2,  dbeta[c] += sum_n P1
//                             S1[n,g] = sum_{c in g} gamma_c P1,  S2[n,g] = sum_{c in g} gamma_c P2
//                             dx = a dy - rstd (S1 + xhat S2) / m,                   m = (C / groups) * H * W
// which is torch's native_group_norm_backward followed by silu_backward, evaluated in fp32.
#include "common.h"
#include "misc.h"

namespace {

__device__ __forceinline__ float silu_fwd(float y) { return y / (1.0f + expf(-y)); }
__device__ __forceinline__ float silu_grad(float y) {
    const float s = 1.0f / (1.0f + expf(-y));
    return s * fmaf(y, 1.0f - s, 1.0f);
}
// the same through v_exp_f32 / v_rcp_f32 (about 1 ulp each) for the bf16 streaming kernels, which the exact version made VALU-bound
__device__ __forceinline__ float silu_grad_fast(float y) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * y));
    return s * fmaf(y, 1.0f - s, 1.0f);
}
// Per-octet coefficients of the apply kernels: the 8 channels' {a, b} as four 16-byte loads, and the (at most two) groups an octet
// touches — {mean, rstd} and {S1, S2} of group g0 and its successor — instead of 24 scalar loads and 8 integer divisions.
// Requires cpg == 4 or cpg >= 8 (every width of this U-Net: 4, 8, 12, 16); `sel[j]` says which of the two groups channel j is in.
struct OctCoef {
    float2 t[8];
    float2 m[2], s[2];
    bool hi[8];
};
__device__ __forceinline__ void load_oct_coef(OctCoef& k, const float2* __restrict__ ab, const float2* __restrict__ mr,
                                              const float2* __restrict__ S, int n, int C, int groups, int cpg, int c0) {
    const f32x4* q = reinterpret_cast<const f32x4*>(ab + (size_t)n * C + c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 v = q[j];
        k.t[2 * j] = make_float2(v[0], v[1]);
        k.t[2 * j + 1] = make_float2(v[2], v[3]);
    }
    const int g0 = c0 / cpg, rem = c0 - g0 * cpg, g1 = min(g0 + 1, groups - 1);
    k.m[0] = mr[(size_t)n * groups + g0], k.m[1] = mr[(size_t)n * groups + g1];
    if (S) k.s[0] = S[(size_t)n * groups + g0], k.s[1] = S[(size_t)n * groups + g1];
#pragma unroll
    for (int j = 0; j < 8; ++j) k.hi[j] = rem + j >= cpg;
}

__device__ __forceinline__ void load8bf(const __bf16* p, float (&v)[8]) {
    const bf16x8 q = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)q[j];
}
__device__ __forceinline__ void store8bf(__bf16* p, const float (&v)[8]) {
    bf16x8 q;
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8*>(p) = q;
}
// the same for fp32 activation storage (the fp32 / split-bf16 compute modes): two 16-byte accesses per octet
__device__ __forceinline__ void load8bf(const float* p, float (&v)[8]) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = lo[0], v[1] = lo[1], v[2] = lo[2], v[3] = lo[3], v[4] = hi[0], v[5] = hi[1], v[6] = hi[2], v[7] = hi[3];
}
__device__ __forceinline__ void store8bf(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// element pointer of channel octet c0 of pixel `pix` in the virtual concat [x1 | x2]
template <typename T>
__device__ __forceinline__ const T* cat_ptr(const T* x1, int C1, const T* x2, int C2, size_t pix, int c0) {
    return (c0 < C1) ? x1 + pix * C1 + c0 : x2 + pix * C2 + (c0 - C1);
}

// Dropout of conv1's operand in training mode (UNetBlock.forward, EDM/network.py:283-284: F.dropout(silu(norm1(x)), p)): the keep
// factors (0 or 1 / (1 - p)) of the 8 consecutive elements starting at flat index e0 (a multiple of 8) of block `blk`'s operand
// come from Philox4x32-10 keyed by the call's seed with counter (e0 / 4 + {0, 1}, blk), so the backward and the forward-mode pass
// regenerate exactly the mask the forward used.  p == 0: all ones.
__device__ __forceinline__ void philox4(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1, c[3] = (uint32_t)p0, c[0] = n0, c[2] = n2;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ void dropout_keep8(const DropArgs& d, uint64_t e0, float (&k)[8]) {
    if (d.p <= 0.f) {
#pragma unroll
        for (int j = 0; j < 8; ++j) k[j] = 1.f;
        return;
    }
    const uint32_t thr = (uint32_t)fminf(d.p * 4294967296.0f, 4294967295.0f);
    const float inv = 1.0f / (1.0f - d.p);
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
        const uint64_t q = e0 / 4 + hq;
        uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), d.block, 0x5eedu};
        philox4(c, (uint32_t)d.seed, (uint32_t)(d.seed >> 32));
#pragma unroll
        for (int j = 0; j < 4; ++j) k[4 * hq + j] = c[j] >= thr ? inv : 0.f;
    }
}

// Gradient tensor t lives at the conv's OUTPUT resolution; fetch what reaches input pixel p (row-major, width `res`) of image n.
// rm 0: same resolution.  rm 1: the forward averaged 2x2 input pixels (down-sampling) -> a quarter of the coarse value.
// rm 2: the forward repeated each input pixel 2x2 (nearest up-sampling) -> the sum of the four fine values.
template <typename T>
__device__ __forceinline__ void fetch_res(const T* t, int Ct, int n, int p, int c0, int res, int rm, float (&v)[8]) {
    if (rm == 0) {
        load8bf(t + ((size_t)n * res * res + p) * Ct + c0, v);
    } else if (rm == 1) {
        const int y = p / res, x = p - y * res, ro = res >> 1;
        load8bf(t + ((size_t)n * ro * ro + (y >> 1) * ro + (x >> 1)) * Ct + c0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= 0.25f;
    } else {
        const int y = p / res, x = p - y * res, ro = res << 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float u[8];
            load8bf(t + ((size_t)n * ro * ro + (2 * y + (d >> 1)) * ro + 2 * x + (d & 1)) * Ct + c0, u);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += u[j];
        }
    }
}

// ---- act = silu(a x + b) (MODE 0), a x + b (MODE 1) or x (MODE 2, ab unused): the conv operand, over the virtual concat, that
// the weight-gradient kernel contracts with -----------------------------------------------------------------------------------
// `res` is the OUTPUT resolution; rm 1: output pixel = mean of the 2x2 activated input pixels (input resolution 2 res), rm 2:
// output pixel (y, x) = activated input pixel (y/2, x/2) (input resolution res/2) - Conv2d.forward's resampling, :114-123.
template <int MODE, typename T>
__global__ __launch_bounds__(256) void gn_act_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2,
                                                     const float2* __restrict__ ab, T* __restrict__ out, int64_t total_oct,
                                                     int res, int rm, DropArgs drop) {
    const int C = C1 + C2, OC = C >> 3, HW = res * res;
    const int ri = rm == 1 ? res * 2 : (rm == 2 ? res / 2 : res);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_oct; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % OC) * 8;
        const size_t pix = (size_t)(i / OC);
        const int n = (int)(pix / HW), p = (int)(pix - (size_t)n * HW);
        const int y = p / res, x = p - y * res;
        const float2* abp = MODE == 2 ? nullptr : ab + (size_t)n * C + c0;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
        const int nsrc = rm == 1 ? 4 : 1;
        for (int d = 0; d < nsrc; ++d) {
            const int sy = rm == 1 ? 2 * y + (d >> 1) : (rm == 2 ? y >> 1 : y);
            const int sx = rm == 1 ? 2 * x + (d & 1) : (rm == 2 ? x >> 1 : x);
            float v[8];
            load8bf(cat_ptr(x1, C1, x2, C2, (size_t)n * ri * ri + sy * ri + sx, c0), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = v[j];
                if (MODE != 2) {
                    t = fmaf(t, abp[j].x, abp[j].y);
                    if (MODE == 0) t = silu_f<true>(t);  // the bf16 conv prologue's SiLU, bit for bit: the kept forward's conv1 consumes this tensor
                }
                o[j] += t;
            }
        }
        if (rm == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] *= 0.25f;
        }
        if (drop.p > 0.f) {
            float keep[8];
            dropout_keep8(drop, (uint64_t)pix * C + c0, keep);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] *= keep[j];
        }
        store8bf(out + pix * C + c0, o);
    }
}

// ---- P1, P2: one workgroup per (64 channels, image); 8 octets x 32 pixel lanes ------------------------------------------------
// x (concat) NHWC, dact [B,HW,Cd] with this tensor's channels at [0, C); MODE as above.  P[n][c] = {P1, P2}.
template <int MODE, typename T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2,
                                                            int C2, const T* __restrict__ dact, int Cd,
                                                            const float2* __restrict__ ab, const float2* __restrict__ mr,
                                                            float2* __restrict__ P, int res, int rm, DropArgs drop) {
    const int C = C1 + C2, HW = res * res;
    const int groups = min(32, C / 4), cpg = C / groups;
    const int n = blockIdx.y, oct = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c0 = blockIdx.x * 64 + oct * 8;
    __shared__ float s1[32][65], s2[32][65];
    float p1[8], p2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p1[j] = p2[j] = 0.f;
    if (c0 < C) {
        float a[8], b[8], mean[8], rstd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 t = ab[(size_t)n * C + c0 + j];
            const float2 m = mr[(size_t)n * groups + (c0 + j) / cpg];
            a[j] = t.x, b[j] = t.y, mean[j] = m.x, rstd[j] = m.y;
        }
        for (int p = pl; p < HW; p += 32) {
            const size_t pix = (size_t)n * HW + p;
            float xv[8], dv[8];
            load8bf(cat_ptr(x1, C1, x2, C2, pix, c0), xv);
            fetch_res(dact, Cd, n, p, c0, res, rm, dv);
            if (drop.p > 0.f) {  // the operand was silu(y) * keep: its gradient passes through the same factors
                float keep[8];
                dropout_keep8(drop, (uint64_t)pix * C + c0, keep);
#pragma unroll
                for (int j = 0; j < 8; ++j) dv[j] *= keep[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dy = MODE == 0 ? dv[j] * silu_grad_fast(fmaf(xv[j], a[j], b[j])) : dv[j];
                p1[j] += dy;
                p2[j] = fmaf(dy, (xv[j] - mean[j]) * rstd[j], p2[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[pl][oct * 8 + j] = p1[j], s2[pl][oct * 8 + j] = p2[j];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
        float a1 = 0.f, a2 = 0.f;
        for (int r = 0; r < 32; ++r) a1 += s1[r][threadIdx.x], a2 += s2[r][threadIdx.x];
        P[(size_t)n * C + blockIdx.x * 64 + threadIdx.x] = make_float2(a1, a2);
    }
}

// ---- S[n][g] = {S1, S2}; one workgroup per image -------------------------------------------------------------------------
__global__ void gn_bwd_group_kernel(const float2* __restrict__ P, const float* __restrict__ gamma, float2* __restrict__ S, int C) {
    const int groups = min(32, C / 4), cpg = C / groups;
    const int n = blockIdx.x, g = threadIdx.x;
    if (g >= groups) return;
    float s1 = 0.f, s2 = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        const float2 p = P[(size_t)n * C + c];
        const float gm = gamma ? gamma[c] : 1.0f;  // unweighted sums for the forward-mode (JVP) use
        s1 = fmaf(gm, p.x, s1);
        s2 = fmaf(gm, p.y, s2);
    }
    S[(size_t)n * groups + g] = make_float2(s1, s2);
}

// ---- dgamma[c] += sum_n P2, dbeta[c] += sum_n P1 (fixed order) ---------------------------------------------------------------
// 64 channels x 4 batch lanes per workgroup: the batch loop is B/4 deep (eight loads in flight per thread), combined in a fixed
// order.  Lives in the second half of gn_bwd_group_param_kernel:
// both in ONE launch (workgroups [0, B): group sums of image n; the rest: 64 channels' parameter sums): they only
// depend on P, and a launch of its own costs each of these tiny kernels ~5 us on the stream's critical path.
__global__ __launch_bounds__(256) void gn_bwd_group_param_kernel(const float2* __restrict__ P, const float* __restrict__ gamma,
                                                                 float2* __restrict__ S, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int B, int C) {
    if ((int)blockIdx.x < B) {
        const int groups = min(32, C / 4), cpg = C / groups;
        const int n = blockIdx.x, g = threadIdx.x;
        if (g >= groups) return;
        float s1 = 0.f, s2 = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const float2 p = P[(size_t)n * C + c];
            const float gm = gamma ? gamma[c] : 1.0f;
            s1 = fmaf(gm, p.x, s1);
            s2 = fmaf(gm, p.y, s2);
        }
        S[(size_t)n * groups + g] = make_float2(s1, s2);
        return;
    }
    __shared__ float sg[4][64], sb[4][64];
    const int cl = threadIdx.x & 63, nl = threadIdx.x >> 6, c = ((int)blockIdx.x - B) * 64 + cl;
    float g = 0.f, b = 0.f;
    if (c < C)
#pragma unroll 8
        for (int n = nl; n < B; n += 4) {
            const float2 p = P[(size_t)n * C + c];
            b += p.x;
            g += p.y;
        }
    sg[nl][cl] = g, sb[nl][cl] = b;
    __syncthreads();
    if (nl == 0 && c < C) {
        if (dgamma) dgamma[c] += (sg[0][cl] + sg[1][cl]) + (sg[2][cl] + sg[3][cl]);
        if (dbeta) dbeta[c] += (sb[0][cl] + sb[1][cl]) + (sb[2][cl] + sb[3][cl]);
    }
}

// ---- dx = a dy - rstd (S1 + xhat S2) / m  [+ add_scale * add]; dx is [B,HW,C] over the whole concat ---------------------------
template <int MODE, typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2,
                                                           int C2, const T* __restrict__ dact, int Cd,
                                                           const float2* __restrict__ ab, const float2* __restrict__ mr,
                                                           const float2* __restrict__ S, const T* __restrict__ add, int Ca,
                                                           float add_scale, T* __restrict__ dx, int dxs, T* __restrict__ dx2,
                                                           int dxs2, int accumulate, int64_t total_oct, int res, int rm, DropArgs drop) {
    // destination: channels [0, C1) -> dx (pixel stride dxs), channels [C1, C) -> dx2 (pixel stride dxs2); accumulate: +=
    const int C = C1 + C2, OC = C >> 3, HW = res * res;
    const int groups = min(32, C / 4), cpg = C / groups;
    const float inv_m = 1.0f / ((float)cpg * (float)HW);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_oct; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % OC) * 8;
        const size_t pix = (size_t)(i / OC);
        const int n = (int)(pix / HW), p = (int)(pix - (size_t)n * HW);
        float xv[8], dv[8], av[8];
        load8bf(cat_ptr(x1, C1, x2, C2, pix, c0), xv);
        fetch_res(dact, Cd, n, p, c0, res, rm, dv);
        if (drop.p > 0.f) {
            float keep[8];
            dropout_keep8(drop, (uint64_t)pix * C + c0, keep);
#pragma unroll
            for (int j = 0; j < 8; ++j) dv[j] *= keep[j];
        }
        if (add) fetch_res(add, Ca, n, p, c0, res, rm, av);
        float o[8];
        OctCoef k;
        load_oct_coef(k, ab, mr, S, n, C, groups, cpg, c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 t = k.t[j];
            const float2 m = k.hi[j] ? k.m[1] : k.m[0];
            const float2 s = k.hi[j] ? k.s[1] : k.s[0];
            const float dy = MODE == 0 ? dv[j] * silu_grad_fast(fmaf(xv[j], t.x, t.y)) : dv[j];
            const float xhat = (xv[j] - m.x) * m.y;
            float r = fmaf(t.x, dy, -m.y * fmaf(xhat, s.y, s.x) * inv_m);
            if (add) r = fmaf(add_scale, av[j], r);
            o[j] = r;
        }
        T* dst = (c0 < C1) ? dx + pix * dxs + c0 : dx2 + pix * dxs2 + (c0 - C1);
        if (accumulate) {
            float old[8];
            load8bf(dst, old);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += old[j];
        }
        store8bf(dst, o);
    }
}

// ---- out[n][c] (+)= scale * sum_p t[n,p,c]: bias and embedding-affine gradients -------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ t, int Ct, int C, float* __restrict__ out, int HW,
                                                     float scale, int out_stride) {
    const int n = blockIdx.y, oct = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c0 = blockIdx.x * 64 + oct * 8;
    __shared__ float s1[32][65];
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (c0 < C)
        for (int p = pl; p < HW; p += 32) {
            float v[8];
            load8bf(t + ((size_t)n * HW + p) * Ct + c0, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[pl][oct * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
        float a = 0.f;
        for (int r = 0; r < 32; ++r) a += s1[r][threadIdx.x];
        out[(size_t)n * out_stride + blockIdx.x * 64 + threadIdx.x] = a * scale;
    }
}

// out[c] += sum_n in[n][c]
// (two destinations: a block's conv1 and skip biases receive the same sum)
__global__ __launch_bounds__(256) void batchsum_add_kernel(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ out2,
                                                           int B, int C, int in_stride) {
    __shared__ float sa[4][64];
    const int cl = threadIdx.x & 63, nl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float a = 0.f;
    if (c < C)
#pragma unroll 8
        for (int n = nl; n < B; n += 4) a += in[(size_t)n * in_stride + c];
    sa[nl][cl] = a;
    __syncthreads();
    if (nl == 0 && c < C) {
        const float t = (sa[0][cl] + sa[1][cl]) + (sa[2][cl] + sa[3][cl]);
        out[c] += t;
        if (out2) out2[c] += t;
    }
}

// fp32 -> bf16 with a scale (gradient entering the block: dOut * skip_scale)
template <typename T>
__global__ void scale_to_bf16_kernel(const float* __restrict__ in, T* __restrict__ out, float scale, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) out[i] = (T)(in[i] * scale);
}
// bf16 [.., Cs] channels [c_off, c_off + C) -> fp32 [.., C]
template <typename T>
__global__ void slice_to_f32_kernel(const T* __restrict__ in, int Cs, int c_off, float* __restrict__ out, int C, int64_t npix) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix * C; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i / C;
        const int c = (int)(i - p * C);
        out[i] = (float)in[p * Cs + c_off + c];
    }
}

// Embedding-affine backward (Linear `affine`, EDM/network.py:255, 278): dtemb [B, C] is the pixel sum of the gradient of
// conv0's output.  dW[c][k] += sum_b dtemb[b][c] emb[b][k];  demb[b][k] += sum_c dtemb[b][c] W[c][k].
__global__ void affine_wgrad_kernel(const float* __restrict__ dtemb, const float* __restrict__ emb, float* __restrict__ dw, int B,
                                    int C, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * K) return;
    const int c = i / K, k = i - c * K;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a = fmaf(dtemb[(size_t)b * C + c], emb[(size_t)b * K + k], a);
    dw[i] += a;
}
__global__ void affine_dgrad_kernel(const float* __restrict__ dtemb, const float* __restrict__ w, float* __restrict__ demb, int B,
                                    int C, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, k = i - b * K;
    float a = 0.f;
    int c = 0;
    for (; c + 8 <= C; c += 8) {  // eight rows of w in flight per thread (the sum keeps its order)
        float d[8], v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = dtemb[(size_t)b * C + c + j], v[j] = w[(size_t)(c + j) * K + k];
#pragma unroll
        for (int j = 0; j < 8; ++j) a = fmaf(d[j], v[j], a);
    }
    for (; c < C; ++c) a = fmaf(dtemb[(size_t)b * C + c], w[(size_t)c * K + k], a);
    demb[i] += a;
}

// Weights of the data-gradient convolution: the forward conv kernel run on dY with Wt[ci][co][tap] = W[co][ci][T-1-tap]
// (rows ci >= cin are zero: output channels padded to the kernel's 256-channel granularity).
__global__ void dgrad_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int cin, int cin_pad, int taps) {
    const int64_t total = (int64_t)cin_pad * cout * taps;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int tap = (int)(i % taps);
        const int co = (int)((i / taps) % cout);
        const int ci = (int)(i / ((int64_t)taps * cout));
        wt[i] = ci < cin ? w[((size_t)co * cin + ci) * taps + (taps - 1 - tap)] : 0.f;
    }
}

// ---- whole-network pieces ------------------------------------------------------------------------------------------------------
// gradient entering the output head: dF = c_out[n] * dout (precond_output, EDM/network.py:798-805), NCHW fp32 -> NHWC bf16 padded
// with zero channels to Cp (the matrix-core kernels' granularity)
template <typename T>
__global__ void head_grad_kernel(const float* __restrict__ dout, const float* __restrict__ c_out, T* __restrict__ out, int C,
                                 int Cp, int HW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        const int64_t pix = i / Cp;
        const int64_t n = pix / HW, p = pix - n * HW;
        out[i] = c < C ? (T)(c_out[n] * dout[(n * C + c) * HW + p]) : (T)0.f;
    }
}
// the stem conv's operand: c_in[n] * x_t (precond_input :771-773), NCHW fp32 -> NHWC bf16 padded to Cp channels
template <typename T>
__global__ void stem_operand_kernel(const float* __restrict__ x, const float* __restrict__ c_in, T* __restrict__ out, int C,
                                    int Cp, int HW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        const int64_t pix = i / Cp;
        const int64_t n = pix / HW, p = pix - n * HW;
        out[i] = c < C ? (T)(c_in[n] * x[(n * C + c) * HW + p]) : (T)0.f;
    }
}
// dst[o][i][t] += src[o][i][t] for o < O, i < I out of a padded [Os][Is][T] tensor
__global__ void add_sub_tensor_kernel(const float* __restrict__ src, int Is, float* __restrict__ dst, int O, int I, int T) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= O * I * T) return;
    const int t = idx % T, i = (idx / T) % I, o = idx / (T * I);
    dst[idx] += src[((size_t)o * Is + i) * T + t];
}
// padded conv weight: dst[o][i][t] = o < O ? src[o][i][t] : 0 for o < Op
__global__ void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int Op, int IT) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Op * IT) return;
    dst[idx] = (idx / IT) < O ? src[idx] : 0.f;
}
template <typename T>
__global__ void add_bf16_kernel(T* __restrict__ dst, const T* __restrict__ src, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        dst[i] = (T)((float)dst[i] + (float)src[i]);
}
// dpre = dy * silu'(pre)   (map_layer0 / map_layer1, EDM/network.py:520-521)
__global__ void silu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dpre, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) dpre[i] = dy[i] * silu_grad(pre[i]);
}
// dW[c][k] += scale * sum_b dy[b][c] x[b][k]
__global__ void linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, int B, int C,
                                    int K, float scale, int dy_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * K) return;
    const int c = i / K, k = i - c * K;
    float a = 0.f;
    int b = 0;
    // eight rows' loads in flight at once (the sum keeps its order): one dependent load per iteration made this latency-bound
    for (; b + 8 <= B; b += 8) {
        float d[8], v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = dy[(size_t)(b + j) * dy_stride + c], v[j] = x[(size_t)(b + j) * K + k];
#pragma unroll
        for (int j = 0; j < 8; ++j) a = fmaf(d[j], v[j], a);
    }
    for (; b < B; ++b) a = fmaf(dy[(size_t)b * dy_stride + c], x[(size_t)b * K + k], a);
    dw[i] += a * scale;
}

// out[c][r] = in[r][c] (fp32)
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < Cc) tile[i][tx] = in[(size_t)(r0 + i) * Cc + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < Cc && r0 + tx < R) out[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

// dst[n][p][c] (bf16 NHWC) += src[n][c][p] (fp32 NCHW): a feature tap's gradient joins the encoder output's gradient
template <typename T>
__global__ void add_nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int HW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int64_t n = pix / HW, p = pix - n * HW;
        dst[i] = (T)((float)dst[i] + src[(n * C + c) * HW + p]);
    }
}
// dx[n][c][p] = c_in[n] * da[n][p][c] (+ c_skip[n] * dout[n][c][p]): gradient of the network input (precond_input / precond_output)
template <typename T>
__global__ void input_grad_kernel(const T* __restrict__ da, int Cd, const float* __restrict__ c_in, const float* __restrict__ c_skip,
                                  const float* __restrict__ dout, float* __restrict__ dx, int C, int HW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % HW;
        const int c = (int)((i / HW) % C);
        const int64_t n = i / ((int64_t)HW * C);
        float v = c_in[n] * (float)da[(n * HW + p) * Cd + c];
        if (dout) v = fmaf(c_skip[n], dout[i], v);
        dx[i] = v;
    }
}

// ---- forward-mode (JVP) pieces: tangents of (x_t, t, r) pushed through the network (mean_flow.py:240-250, sCM.py:179) -------------
// GroupNorm(+SiLU) tangent: with P = {sum_p xd, sum_p xd*xhat} per (n, c) and S their UNWEIGHTED group sums,
//   yd = a (xd - (S1 + xhat S2) / m),  actd = silu'(a x + b) yd (MODE 0) | yd (MODE 1);   xd [B,HW,Cd] covers the concat.
template <int MODE, typename T>
__global__ __launch_bounds__(256) void gn_jvp_apply_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2,
                                                           int C2, const T* __restrict__ xd, int Cd,
                                                           const float2* __restrict__ ab, const float2* __restrict__ mr,
                                                           const float2* __restrict__ S, T* __restrict__ out, int64_t total_oct,
                                                           int HW, DropArgs drop) {
    const int C = C1 + C2, OC = C >> 3;
    const int groups = min(32, C / 4), cpg = C / groups;
    const float inv_m = 1.0f / ((float)cpg * (float)HW);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_oct; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % OC) * 8;
        const size_t pix = (size_t)(i / OC);
        const int n = (int)(pix / HW);
        float xv[8], dv[8], o[8];
        load8bf(cat_ptr(x1, C1, x2, C2, pix, c0), xv);
        load8bf(xd + pix * Cd + c0, dv);
        OctCoef k;
        load_oct_coef(k, ab, mr, S, n, C, groups, cpg, c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 t = k.t[j];
            const float2 m = k.hi[j] ? k.m[1] : k.m[0];
            const float2 s = k.hi[j] ? k.s[1] : k.s[0];
            const float xhat = (xv[j] - m.x) * m.y;
            float r = t.x * (dv[j] - fmaf(xhat, s.y, s.x) * inv_m);
            if (MODE == 0) r *= silu_grad_fast(fmaf(xv[j], t.x, t.y));
            o[j] = r;
        }
        if (drop.p > 0.f) {
            float keep[8];
            dropout_keep8(drop, (uint64_t)pix * C + c0, keep);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] *= keep[j];
        }
        store8bf(out + pix * C + c0, o);
    }
}
// Preconditioning coefficients and their t-derivatives (EDM/network.py:755-805): ct[0..7][B] = c_in, dc_in, dc_noise, dr_noise,
// c_skip, dc_skip, c_out, dc_out (the d* already multiplied by the tangents vt / vr).  drop bit 0: no input preconditioning
// (noise labels are t, r themselves); bit 1: none on the output.
__global__ void jvp_coef_kernel(const double* __restrict__ t, const double* __restrict__ r, const float* __restrict__ vt,
                                const float* __restrict__ vr, double sigma_data, double sigma_shift, int drop, float* __restrict__ ct, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double tv = t[b], dv = vt ? (double)vt[b] : 0.0, rv = r ? r[b] : 0.0, drv = (r && vr) ? (double)vr[b] : 0.0;
    const double s2 = sigma_data * sigma_data;
    if (drop & 1) {
        ct[b] = 1.f, ct[B + b] = 0.f, ct[2 * B + b] = (float)dv, ct[3 * B + b] = (float)drv;
    } else {
        const double q = s2 + tv * tv;
        ct[b] = (float)(1.0 / sqrt(q));
        ct[B + b] = (float)(-tv / (q * sqrt(q)) * dv);
        ct[2 * B + b] = tv > 1e-6 ? (float)(dv / (4.0 * tv)) : 0.f;  // c_noise = ln(clamp(t, 1e-6)) / 4
        ct[3 * B + b] = rv > 1e-6 ? (float)(drv / (4.0 * rv)) : 0.f;
    }
    if (drop & 2) {
        ct[4 * B + b] = 0.f, ct[5 * B + b] = 0.f, ct[6 * B + b] = 1.f, ct[7 * B + b] = 0.f;
    } else {
        const double ts = tv - sigma_shift, q = ts * ts + s2;
        ct[4 * B + b] = (float)(s2 / q);
        ct[5 * B + b] = (float)(-2.0 * ts * s2 / (q * q) * dv);
        ct[6 * B + b] = (float)(ts * sigma_data / sqrt(q));
        ct[7 * B + b] = (float)(sigma_data * s2 / (q * sqrt(q)) * dv);
    }
}
// tangent of the mapping network's input (positional embeddings of c_noise and r_noise; the label part has none), [B][N]
__global__ void jvp_embed_kernel(const float* __restrict__ c_noise, const float* __restrict__ r_noise, const float* __restrict__ dc,
                                 const float* __restrict__ dr, const float* __restrict__ freqs, float* __restrict__ out, int B, int N,
                                 int noise_ch) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N) return;
    const int b = idx / N, j = idx % N, half = noise_ch / 2, jj = j % noise_ch;
    const float lab = (j < noise_ch) ? c_noise[b] : r_noise[b];
    const float dl = (j < noise_ch) ? dc[b] : dr[b];
    const float f = freqs[jj % half], ang = lab * f;
    out[idx] = (jj < half) ? cosf(ang) * f * dl : -sinf(ang) * f * dl;
}
// xd_in = c_in vx + dc_in x  (NCHW fp32, per image)
__global__ void jvp_input_kernel(const float* __restrict__ vx, const float* __restrict__ x, const float* __restrict__ c_in,
                                 const float* __restrict__ dc_in, float* __restrict__ out, int CHW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / CHW;
        out[i] = fmaf(c_in[n], vx[i], dc_in[n] * x[i]);
    }
}
// jvp[n][c][p] = c_out Fd + dc_out F + c_skip vx + dc_skip x,  F = the network's raw output (kept by the forward that precedes);
// Fd is NHWC bf16 with channel stride Cf
template <typename T>
__global__ void jvp_output_kernel(const T* __restrict__ fd, int Cf, const float* __restrict__ F_raw, const float* __restrict__ x,
                                  const float* __restrict__ vx, const float* __restrict__ ct, float* __restrict__ jvp, int B, int C, int HW,
                                  int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % HW;
        const int c = (int)((i / HW) % C);
        const int64_t n = i / ((int64_t)HW * C);
        const float cs = ct[4 * B + n], dcs = ct[5 * B + n], co = ct[6 * B + n], dco = ct[7 * B + n];
        jvp[i] = fmaf(co, (float)fd[(n * HW + p) * Cf + c], dco * F_raw[i]) + fmaf(cs, vx[i], dcs * x[i]);
    }
}
__global__ void fill_f32_kernel(float* __restrict__ p, float v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void fill_f2_kernel(float2* __restrict__ p, float2 v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// the keep factors themselves (parity tests feed them to the oracle): out[e] for e < total, total % 8 == 0
__global__ void dropout_mask_kernel(float* __restrict__ out, int64_t total, DropArgs drop) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total / 8; i += (int64_t)gridDim.x * 256) {
        float keep[8];
        dropout_keep8(drop, (uint64_t)i * 8, keep);
#pragma unroll
        for (int j = 0; j < 8; ++j) out[i * 8 + j] = keep[j];
    }
}

inline unsigned ew_blocks(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

}  // namespace

#define BWD_RET() return (int)hipGetLastError()
// storage type of the activation tensors: dtype 1 = bf16 (bf16 compute mode), 0 = fp32 (fp32 / split-bf16 modes)
#define ACT_T(dtype, ...)         \
    do {                          \
        if (dtype) {              \
            typedef __bf16 T;     \
            __VA_ARGS__;          \
        } else {                  \
            typedef float T;      \
            __VA_ARGS__;          \
        }                         \
    } while (0)

int launch_gn_act(int dtype, int mode, const void* x1, int c1, const void* x2, int c2, const float2* ab, void* out, int B, int res, int rm,
                  hipStream_t s, DropArgs drop) {
    if ((c1 % 8) || (c2 % 8)) return (int)hipErrorInvalidValue;
    const int hw = res * res;
    const int64_t total = (int64_t)B * hw * ((c1 + c2) / 8);
    if (mode == 0)
        ACT_T(dtype, hipLaunchKernelGGL((gn_act_kernel<0, T>), dim3(ew_blocks(total)), dim3(256), 0, s, (const T*)x1, c1, (const T*)x2, c2, ab, (T*)out, total, res, rm, drop));
    else if (mode == 2)
        ACT_T(dtype, hipLaunchKernelGGL((gn_act_kernel<2, T>), dim3(ew_blocks(total)), dim3(256), 0, s, (const T*)x1, c1, (const T*)x2, c2, ab, (T*)out, total, res, rm, drop));
    else
        ACT_T(dtype, hipLaunchKernelGGL((gn_act_kernel<1, T>), dim3(ew_blocks(total)), dim3(256), 0, s, (const T*)x1, c1, (const T*)x2, c2, ab, (T*)out, total, res, rm, drop));
    BWD_RET();
}

// GroupNorm(+SiLU) backward.  P: [B][C] float2 scratch, S: [B][groups] float2 scratch.  dgamma / dbeta are accumulated.
int launch_gn_bwd(int dtype, int mode, const void* x1, int c1, const void* x2, int c2, const void* dact, int cd, const float2* ab,
                  const float2* mr, const float* gamma, float2* P, float2* S, float* dgamma, float* dbeta, const void* add, int ca,
                  float add_scale, void* dx, int B, int res, int rm, hipStream_t s, void* dx2, int accumulate, DropArgs drop) {
    const int C = c1 + c2, hw = res * res;
    // default destination: one [B, hw, C] tensor; with dx2 the two halves of the concat go to their own (dense) tensors
    const size_t esz = dtype ? 2 : 4;
    char* d1 = (char*)dx;
    char* d2 = dx2 ? (char*)dx2 : d1 + (size_t)c1 * esz;
    const int s1 = dx2 ? c1 : C, s2 = dx2 ? c2 : C;
    if ((c1 % 8) || (c2 % 8) || C < 16) return (int)hipErrorInvalidValue;
    const int groups = C / 4 < 32 ? C / 4 : 32;
    if (C / groups != 4 && C / groups < 8) return (int)hipErrorInvalidValue;  // load_oct_coef: an octet spans at most two groups
    dim3 rg((C + 63) / 64, B);
    if (mode == 0)
        ACT_T(dtype, hipLaunchKernelGGL((gn_bwd_reduce_kernel<0, T>), rg, dim3(256), 0, s, (const T*)x1, c1, (const T*)x2, c2, (const T*)dact, cd, ab, mr, P, res, rm, drop));
    else
        ACT_T(dtype, hipLaunchKernelGGL((gn_bwd_reduce_kernel<1, T>), rg, dim3(256), 0, s, (const T*)x1, c1, (const T*)x2, c2, (const T*)dact, cd, ab, mr, P, res, rm, drop));
    if (dgamma || dbeta)
        hipLaunchKernelGGL(gn_bwd_group_param_kernel, dim3(B + (C + 63) / 64), dim3(256), 0, s, P, gamma, S, dgamma, dbeta, B, C);
    else
        hipLaunchKernelGGL(gn_bwd_group_kernel, dim3(B), dim3(32), 0, s, P, gamma, S, C);
    const int64_t total = (int64_t)B * hw * (C / 8);
    if (mode == 0)
        ACT_T(dtype, hipLaunchKernelGGL((gn_bwd_apply_kernel<0, T>), dim3(ew_blocks(total)), dim3(256), 0, s, (const T*)x1, c1, (const T*)x2, c2, (const T*)dact, cd, ab, mr, S, (const T*)add, ca, add_scale, (T*)d1, s1, (T*)d2, s2, accumulate, total, res, rm, drop));
    else
        ACT_T(dtype, hipLaunchKernelGGL((gn_bwd_apply_kernel<1, T>), dim3(ew_blocks(total)), dim3(256), 0, s, (const T*)x1, c1, (const T*)x2, c2, (const T*)dact, cd, ab, mr, S, (const T*)add, ca, add_scale, (T*)d1, s1, (T*)d2, s2, accumulate, total, res, rm, drop));
    (void)groups;
    BWD_RET();
}

int launch_colsum(int dtype, const void* t, int ct, int C, float* out, int B, int hw, float scale, hipStream_t s, int out_stride) {
    if (C % 8) return (int)hipErrorInvalidValue;
    ACT_T(dtype, hipLaunchKernelGGL(colsum_kernel<T>, dim3((C + 63) / 64, B), dim3(256), 0, s, (const T*)t, ct, C, out, hw, scale,
                                    out_stride > 0 ? out_stride : C));
    BWD_RET();
}
int launch_batchsum_add(const float* in, float* out, int B, int C, hipStream_t s, float* out2, int in_stride) {
    hipLaunchKernelGGL(batchsum_add_kernel, dim3((C + 63) / 64), dim3(256), 0, s, in, out, out2, B, C, in_stride > 0 ? in_stride : C);
    BWD_RET();
}
int launch_scale_to_act(int dtype, const float* in, void* out, float scale, int64_t total, hipStream_t s) {
    ACT_T(dtype, hipLaunchKernelGGL(scale_to_bf16_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, s, in, (T*)out, scale, total));
    BWD_RET();
}
int launch_slice_to_f32(int dtype, const void* in, int cs, int c_off, float* out, int C, int64_t npix, hipStream_t s) {
    ACT_T(dtype, hipLaunchKernelGGL(slice_to_f32_kernel<T>, dim3(ew_blocks(npix * C)), dim3(256), 0, s, (const T*)in, cs, c_off, out, C, npix));
    BWD_RET();
}
int launch_affine_bwd(const float* dtemb, const float* emb, const float* w, float* dw, float* demb, int B, int C, int K, hipStream_t s) {
    if (dw) hipLaunchKernelGGL(affine_wgrad_kernel, dim3((C * K + 255) / 256), dim3(256), 0, s, dtemb, emb, dw, B, C, K);
    if (demb) hipLaunchKernelGGL(affine_dgrad_kernel, dim3((B * K + 255) / 256), dim3(256), 0, s, dtemb, w, demb, B, C, K);
    BWD_RET();
}
int launch_dgrad_weights(const float* w, float* wt, int cout, int cin, int cin_pad, int taps, hipStream_t s) {
    hipLaunchKernelGGL(dgrad_weights_kernel, dim3(ew_blocks((int64_t)cin_pad * cout * taps)), dim3(256), 0, s, w, wt, cout, cin, cin_pad, taps);
    BWD_RET();
}
int launch_head_grad(int dtype, const float* dout, const float* c_out, void* out, int B, int C, int Cp, int hw, hipStream_t s) {
    const int64_t total = (int64_t)B * hw * Cp;
    ACT_T(dtype, hipLaunchKernelGGL(head_grad_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, s, dout, c_out, (T*)out, C, Cp, hw, total));
    BWD_RET();
}
int launch_stem_operand(int dtype, const float* x, const float* c_in, void* out, int B, int C, int Cp, int hw, hipStream_t s) {
    const int64_t total = (int64_t)B * hw * Cp;
    ACT_T(dtype, hipLaunchKernelGGL(stem_operand_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, s, x, c_in, (T*)out, C, Cp, hw, total));
    BWD_RET();
}
int launch_add_sub_tensor(const float* src, int Is, float* dst, int O, int I, int T, hipStream_t s) {
    hipLaunchKernelGGL(add_sub_tensor_kernel, dim3((O * I * T + 255) / 256), dim3(256), 0, s, src, Is, dst, O, I, T);
    BWD_RET();
}
int launch_pad_rows(const float* src, float* dst, int O, int Op, int IT, hipStream_t s) {
    hipLaunchKernelGGL(pad_rows_kernel, dim3((Op * IT + 255) / 256), dim3(256), 0, s, src, dst, O, Op, IT);
    BWD_RET();
}
int launch_add_act(int dtype, void* dst, const void* src, int64_t total, hipStream_t s) {
    ACT_T(dtype, hipLaunchKernelGGL(add_bf16_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, s, (T*)dst, (const T*)src, total));
    BWD_RET();
}
int launch_silu_bwd(const float* dy, const float* pre, float* dpre, int total, hipStream_t s) {
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, dy, pre, dpre, total);
    BWD_RET();
}
// Linear backward: dw[C][K] += scale * dy^T x, db[C] += sum_b dy, dx[B][K] += dy w  (each optional)
int launch_linear_bwd(const float* dy, const float* x, const float* w, float* dw, float* db, float* dx, int B, int C, int K,
                      float scale, hipStream_t s, int dy_stride) {
    if ((db || dx) && dy_stride > 0 && dy_stride != C) return (int)hipErrorInvalidValue;  // only the weight part takes a stride
    if (dw) hipLaunchKernelGGL(linear_wgrad_kernel, dim3((C * K + 255) / 256), dim3(256), 0, s, dy, x, dw, B, C, K, scale, dy_stride > 0 ? dy_stride : C);
    if (db) hipLaunchKernelGGL(batchsum_add_kernel, dim3((C + 63) / 64), dim3(256), 0, s, dy, db, (float*)nullptr, B, C, C);
    if (dx) hipLaunchKernelGGL(affine_dgrad_kernel, dim3((B * K + 255) / 256), dim3(256), 0, s, dy, w, dx, B, C, K);
    BWD_RET();
}
int launch_transpose_f32(const float* in, float* out, int R, int Cc, hipStream_t s) {
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, s, in, out, R, Cc);
    BWD_RET();
}
int launch_add_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int hw, hipStream_t s) {
    const int64_t total = (int64_t)B * C * hw;
    ACT_T(dtype, hipLaunchKernelGGL(add_nchw_to_nhwc_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, s, src, (T*)dst, C, hw, total));
    BWD_RET();
}
int launch_input_grad(int dtype, const void* da, int cd, const float* c_in, const float* c_skip, const float* dout, float* dx, int B, int C, int hw,
                      hipStream_t s) {
    const int64_t total = (int64_t)B * C * hw;
    ACT_T(dtype, hipLaunchKernelGGL(input_grad_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, s, (const T*)da, cd, c_in, c_skip, dout, dx, C, hw, total));
    BWD_RET();
}
// GroupNorm(+SiLU) tangent of xd (dense [B, hw, C] over the concat) -> out [B, hw, C]; P / S scratch as in launch_gn_bwd
int launch_gn_jvp(int dtype, int mode, const void* x1, int c1, const void* x2, int c2, const void* xd, const float2* ab, const float2* mr,
                  float2* P, float2* S, void* out, int B, int res, hipStream_t s, DropArgs drop) {
    const int C = c1 + c2, hw = res * res;
    if ((c1 % 8) || (c2 % 8) || C < 16) return (int)hipErrorInvalidValue;
    {
        const int groups = C / 4 < 32 ? C / 4 : 32;
        if (C / groups != 4 && C / groups < 8) return (int)hipErrorInvalidValue;  // load_oct_coef
    }
    ACT_T(dtype, hipLaunchKernelGGL((gn_bwd_reduce_kernel<1, T>), dim3((C + 63) / 64, B), dim3(256), 0, s, (const T*)x1, c1, (const T*)x2, c2, (const T*)xd, C, ab, mr, P, res, 0, DropArgs{}));
    hipLaunchKernelGGL(gn_bwd_group_kernel, dim3(B), dim3(32), 0, s, P, (const float*)nullptr, S, C);
    const int64_t total = (int64_t)B * hw * (C / 8);
    if (mode == 0)
        ACT_T(dtype, hipLaunchKernelGGL((gn_jvp_apply_kernel<0, T>), dim3(ew_blocks(total)), dim3(256), 0, s, (const T*)x1, c1, (const T*)x2, c2, (const T*)xd, C, ab, mr, S, (T*)out, total, hw, drop));
    else
        ACT_T(dtype, hipLaunchKernelGGL((gn_jvp_apply_kernel<1, T>), dim3(ew_blocks(total)), dim3(256), 0, s, (const T*)x1, c1, (const T*)x2, c2, (const T*)xd, C, ab, mr, S, (T*)out, total, hw, drop));
    BWD_RET();
}
int launch_jvp_coef(const double* t, const double* r, const float* vt, const float* vr, double sigma_data, double sigma_shift, int drop,
                    float* ct, int B, hipStream_t s) {
    hipLaunchKernelGGL(jvp_coef_kernel, dim3((B + 127) / 128), dim3(128), 0, s, t, r, vt, vr, sigma_data, sigma_shift, drop, ct, B);
    BWD_RET();
}
int launch_jvp_embed(const float* c_noise, const float* r_noise, const float* dc, const float* dr, const float* freqs, float* out, int B,
                     int N, int noise_ch, hipStream_t s) {
    hipLaunchKernelGGL(jvp_embed_kernel, dim3((B * N + 255) / 256), dim3(256), 0, s, c_noise, r_noise, dc, dr, freqs, out, B, N, noise_ch);
    BWD_RET();
}
int launch_jvp_input(const float* vx, const float* x, const float* c_in, const float* dc_in, float* out, int B, int chw, hipStream_t s) {
    const int64_t total = (int64_t)B * chw;
    hipLaunchKernelGGL(jvp_input_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, vx, x, c_in, dc_in, out, chw, total);
    BWD_RET();
}
int launch_jvp_output(int dtype, const void* fd, int cf, const float* out, const float* x, const float* vx, const float* ct, float* jvp, int B, int C,
                      int hw, hipStream_t s) {
    const int64_t total = (int64_t)B * C * hw;
    ACT_T(dtype, hipLaunchKernelGGL(jvp_output_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, s, (const T*)fd, cf, out, x, vx, ct, jvp, B, C, hw, total));
    BWD_RET();
}
int launch_fill_f32(float* p, float v, int n, hipStream_t s) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n);
    BWD_RET();
}
int launch_fill_f2(float2* p, float a, float b, int n, hipStream_t s) {
    hipLaunchKernelGGL(fill_f2_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, make_float2(a, b), n);
    BWD_RET();
}
int launch_dropout_mask(float* out, int64_t total, DropArgs drop, hipStream_t s) {
    if (total % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_blocks(total / 8)), dim3(256), 0, s, out, total, drop);
    BWD_RET();
}
