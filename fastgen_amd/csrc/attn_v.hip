// Single-head self-attention of the EDM U-Net blocks (reference fastgen/networks/EDM/network.py:160-168, 290-296):
//     w = softmax_k( q^T (k / sqrt(C)) )  in fp32,   a[c,q] = sum_k w[q,k] v[c,k]
// C = 256 channels = ONE head of dim 256, T = 256 (16x16) or 64 (8x8) tokens.
//
// One wave owns 32 queries end to end; nothing goes through LDS:
//   1. S^T = K Q^T on the matrix cores with the KEY on the accumulator rows and the QUERY on the lane, so a lane
//      holds half of the T logits of its query (the other half sits in lane^32) and the softmax is in-register.
//   2. The normalised P^T accumulators are reused directly as the A operand of O = P V (an accumulator tile whose
//      column is on the lane feeds the next MFMA that sums over its row index); V is read from a [C][T] plane the
//      qkv projection already wrote, in the K order the accumulator layout dictates.
#include <cstdlib>
#include "common.h"
#include "misc.h"

namespace {

template <typename T>
struct PFrag;  // the P^T accumulator registers 8s..8s+7 as an A operand
template <>
struct PFrag<__bf16> {
    static __device__ __forceinline__ Frag8<__bf16> make(const f32x16& p, int s) {
        Frag8<__bf16> f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = (__bf16)p[8 * s + j];
        return f;
    }
};
template <>
struct PFrag<float> {
    static __device__ __forceinline__ Frag8<float> make(const f32x16& p, int s) {
        Frag8<float> f;
        f.lo = f32x4{p[8 * s + 0], p[8 * s + 1], p[8 * s + 2], p[8 * s + 3]};
        f.hi = f32x4{p[8 * s + 4], p[8 * s + 5], p[8 * s + 6], p[8 * s + 7]};
        return f;
    }
};

// bf16x3 (common.h): q, k, v^T and the output are fp32 in memory; operands are split into hi / lo bf16 in registers on their
// way to the matrix cores (three bf16 MFMAs per product instead of the eight exact-fp32 ones of the float instantiation)
template <>
struct PFrag<bf16x3> {
    static __device__ __forceinline__ Frag8<bf16x3> make(const f32x16& p, int s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[8 * s + j];
        Frag8<bf16x3> f;
        split8(v, f.hi, f.lo);
        return f;
    }
};
// 8 consecutive operand elements of the storage type as a fragment of compute type T
template <typename T>
__device__ __forceinline__ Frag8<T> load_op(const typename DT<T>::ST* p) {
    return load_frag(p);
}
template <>
__device__ __forceinline__ Frag8<bf16x3> load_op<bf16x3>(const float* p) {
    float v[8];
    widen8(load_frag(p), v);
    Frag8<bf16x3> f;
    split8(v, f.hi, f.lo);
    return f;
}

// V fragment for k-step s of key tile kt: element j <-> key kt*32 + 16s + 8(j>>2) + 4h + (j&3)
__device__ __forceinline__ Frag8<__bf16> load_v(const __bf16* row, int key0) {
    Frag8<__bf16> f;
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(row + key0);
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(row + key0 + 8);
    f.v = bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return f;
}
__device__ __forceinline__ Frag8<float> load_v(const float* row, int key0) {
    Frag8<float> f;
    f.lo = *reinterpret_cast<const f32x4*>(row + key0);
    f.hi = *reinterpret_cast<const f32x4*>(row + key0 + 8);
    return f;
}

template <typename T>
__device__ __forceinline__ Frag8<T> load_vop(const typename DT<T>::ST* row, int key0) {
    return load_v(row, key0);
}
template <>
__device__ __forceinline__ Frag8<bf16x3> load_vop<bf16x3>(const float* row, int key0) {
    float v[8];
    widen8(load_v(row, key0), v);
    Frag8<bf16x3> f;
    split8(v, f.hi, f.lo);
    return f;
}

template <typename T, int NT>  // NT = T/32 key tiles; T = compute type (common.h), tensors in its storage type
__global__ __launch_bounds__(256, 2) void attention_kernel(const typename DT<T>::ST* __restrict__ q, const typename DT<T>::ST* __restrict__ k,
                                                        const typename DT<T>::ST* __restrict__ vt, typename DT<T>::ST* __restrict__ out, int B) {
    typedef typename DT<T>::ST ST;
    constexpr int Tn = NT * 32;
    constexpr int D = 256;
    constexpr bool FAST = DT<T>::FAST;
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);  // global wave id = (image, 32-query slab)
    const int n = gw / NT, q0 = (gw % NT) * 32;
    if (n >= B) return;  // wave-uniform

    const ST* qrow = q + ((size_t)n * Tn + q0 + r) * D + 8 * h;
    const ST* kbase = k + ((size_t)n * Tn + r) * D + 8 * h;

    // ---- S^T[key][query] = sum_d K[key][d] Q[query][d] --------------------------------------------------
    f32x16 st[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[kt][i] = 0.f;
#pragma unroll 2
    for (int kk = 0; kk < D / 16; ++kk) {
        const Frag8<T> qf = load_op<T>(qrow + kk * 16);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const Frag8<T> kf = load_op<T>(kbase + (size_t)kt * 32 * D + kk * 16);
            mma16(st[kt], kf, qf);
        }
    }

    // ---- softmax over keys (registers + the partner half-wave) ---------------------------------------------
    const float sc = 0.0625f;  // 1/sqrt(256): the reference scales k, an exact power of two either way
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

    // ---- O[query][dim] = sum_key P[query][key] V[key][dim], OG 32-wide dim tiles at a time --------------------------------------
    // (two at a time for the 256-key form: with the 128 registers of S^T and four output tiles the kernel took 386 registers = one
    // wave per SIMD, and these operand loads straight from global memory need the second wave to hide behind)
    constexpr int OG = NT >= 8 ? 2 : 4;
    const ST* vbase = vt + ((size_t)n * D + r) * Tn + 4 * h;
    ST* obase = out + ((size_t)n * Tn + q0) * D + r;
#pragma unroll 1
    for (int dg = 0; dg < D / (32 * OG); ++dg) {
        f32x16 o[OG];
#pragma unroll
        for (int d = 0; d < OG; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const Frag8<T> pf = PFrag<T>::make(st[kt], s);
#pragma unroll
                for (int d = 0; d < OG; ++d) {
                    const Frag8<T> vf = load_vop<T>(vbase + (size_t)(dg * 32 * OG + d * 32) * Tn, kt * 32 + 16 * s);
                    mma16(o[d], pf, vf);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < OG; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) obase[(size_t)acc_row(i, h) * D + dg * 32 * OG + d * 32] = (ST)o[d][i];
    }
}

// ---- bf16, T = 256: K and V^T tiles staged through LDS -----------------------------------------------------------
// The direct-from-global form above reads every K / V fragment as 64 scattered 16-byte (8-byte) pieces and every wave
// of an image re-reads them.  Here one workgroup = 4 waves = 128 queries of one image; 64-key tiles of K ([key][dim],
// 528-byte pitch: conflict-free ds_read_b128) and of V^T ([dim][key], 136-byte pitch: conflict-free ds_read_b64) are
// copied with coalesced 16-byte loads into a double-buffered LDS ring shared by the four waves.  Q stays in registers
// (64 VGPRs), S^T in 128 accumulators, then P as 64 VGPRs of bf16 fragments next to the 128 accumulators of O.
constexpr int KPITCH = 528, VPITCH = 136;
constexpr int ATT_BUF = 256 * VPITCH;  // >= 64 * KPITCH

__global__ __launch_bounds__(256, 2) void attention_lds_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                               const __bf16* vt, __bf16* __restrict__ out,
                                                               int B) {
    constexpr int Tn = 256, D = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.x >> 1;
    const int q0 = (blockIdx.x & 1) * 128 + wave * 32;

    auto stage_k = [&](int kt64, char* buf) {  // 64 keys x 256 dims: 2048 16-byte chunks, 8 per thread (two batches)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
#pragma unroll
            for (int i = b2 * 4; i < b2 * 4 + 4; ++i) {
                const int idx = tid + 256 * i, row = idx >> 5, c = idx & 31;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(k + ((size_t)n * Tn + kt64 * 64 + row) * D + c * 8);
                *reinterpret_cast<bf16x8*>(buf + row * KPITCH + c * 16) = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stage_v = [&](int kt64, char* buf) {  // 256 dims x 64 keys: 2048 chunks, written as 2 x 8 bytes (pitch % 16 = 8)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
#pragma unroll
            for (int i = b2 * 4; i < b2 * 4 + 4; ++i) {
                const int idx = tid + 256 * i, row = idx >> 3, c = idx & 7;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(vt + ((size_t)n * D + row) * Tn + kt64 * 64 + c * 8);
                bf16x4* dst = reinterpret_cast<bf16x4*>(buf + row * VPITCH + c * 16);
                dst[0] = bf16x4{v[0], v[1], v[2], v[3]};
                dst[1] = bf16x4{v[4], v[5], v[6], v[7]};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Q fragments of this wave's 32 queries (B operand of S^T = K Q^T): 16 KB per wave, re-read from L1/L2 per key tile
    const __bf16* qrow = q + ((size_t)n * Tn + q0 + r) * D + 8 * h;
    f32x16 st[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[kt][i] = 0.f;

    stage_k(0, smem);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        char* cur = smem + (t & 1) * ATT_BUF;
        if (t + 1 < 4) stage_k(t + 1, smem + ((t + 1) & 1) * ATT_BUF);
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
            const Frag8<__bf16> qf = load_frag(qrow + kk * 16);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const Frag8<__bf16> kf = load_frag(reinterpret_cast<const __bf16*>(cur + (sub * 32 + r) * KPITCH + h * 16) + kk * 16);
                mma16(st[t * 2 + sub], kf, qf);
            }
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- softmax over keys (registers + the partner half-wave), as in attention_kernel ----------------------------
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) m = fmaxf(m, st[kt][i]);
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __builtin_amdgcn_exp2f(1.44269504088896341f * 0.0625f * (st[kt][i] - m));
            st[kt][i] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32);
    const float inv = __builtin_amdgcn_rcpf(sum);
    Frag8<__bf16> pf[8][2];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[kt][s2].v[j] = (__bf16)(st[kt][8 * s2 + j] * inv);
        __builtin_amdgcn_sched_barrier(0);  // convert tile by tile: products of all tiles at once would double the footprint
    }

    __builtin_amdgcn_sched_barrier(0);  // keep the softmax / P conversion (peak register use) clear of what follows
    // ---- O = P V over 64-key tiles of V^T, four 32-wide dim tiles at a time (two sweeps over the V^T tiles) --------------
    __bf16* obase = out + ((size_t)n * Tn + q0) * D + r;
#pragma unroll 1
    for (int dg = 0; dg < 2; ++dg) {
        f32x16 o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
        asm volatile("" ::: "memory");  // the V^T loads do not depend on dg: stop LICM from hoisting all of them out of the loop
        stage_v(0, smem);         // buffer 0 is free: the previous phase ended with a barrier
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const char* cur = smem + (t & 1) * ATT_BUF;
            if (t + 1 < 4) stage_v(t + 1, smem + ((t + 1) & 1) * ATT_BUF);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        // element j <-> key 32*sub + 16*s2 + 8*(j>>2) + 4h + (j&3) of the tile
                        const char* p = cur + ((dg * 4 + d) * 32 + r) * VPITCH + (sub * 32 + 16 * s2 + 4 * h) * 2;
                        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p);
                        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(p + 16);
                        Frag8<__bf16> vf;
                        vf.v = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        mma16(o[d], pf[t * 2 + sub][s2], vf);
                    }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) obase[(size_t)acc_row(i, h) * D + (dg * 4 + d) * 32] = (__bf16)o[d][i];
    }
}

// ---- bf16x3, T = 256: fp32 K and V^T tiles split into hi / lo bf16 planes on their way into LDS ------------------------
// The direct-from-global form reads 512 KB of fp32 K / V^T per wave (4 MB per image through L2 in scattered 16-byte pieces) and
// splits every fragment in registers again in each of the eight waves of an image.  Here a workgroup of 4 waves = 128 queries
// stages 64-key tiles once: K as [key][128 dims] (272-byte pitch, hi plane then lo plane), V^T as [128 dims][64 keys]
// (136-byte pitch).  The head dimension goes in two halves so that the split Q fragments of a half (64 registers) fit next to
// the 128 accumulators of S^T within 256 registers (two waves per SIMD, two workgroups per CU on 2 x 34 KB of LDS each).
constexpr int X3_KP = 272, X3_VP = 136;
constexpr int X3_PLANE = 64 * X3_KP;  // = 128 * X3_VP
constexpr int X3_BUF = 2 * X3_PLANE;  // hi + lo

__global__ __launch_bounds__(256, 2) void attention_x3_lds_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                  const float* __restrict__ vt, float* __restrict__ out, int B) {
    constexpr int Tn = 256, D = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.x >> 1;
    const int q0 = (blockIdx.x & 1) * 128 + wave * 32;

    auto stage_k = [&](int s, char* buf) {  // step s = 4 * dim half + key tile: 64 keys x 128 dims, 1024 8-element chunks
        const float* src = k + ((size_t)n * Tn + (s & 3) * 64) * D + (s >> 2) * 128;
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
#pragma unroll
            for (int i = b2 * 2; i < b2 * 2 + 2; ++i) {
                const int idx = tid + 256 * i, row = idx >> 4, c = idx & 15;
                float x[8];
                widen8(load_frag(src + (size_t)row * D + c * 8), x);
                bf16x8 hi, lo;
                split8(x, hi, lo);
                *reinterpret_cast<bf16x8*>(buf + row * X3_KP + c * 16) = hi;
                *reinterpret_cast<bf16x8*>(buf + X3_PLANE + row * X3_KP + c * 16) = lo;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stage_v = [&](int s, char* buf) {  // step s = 4 * dim half + key tile: 128 dims x 64 keys, written as 2 x 8 bytes (pitch % 16 = 8)
        const float* src = vt + ((size_t)n * D + (s >> 2) * 128) * Tn + (s & 3) * 64;
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) {
#pragma unroll
            for (int i = b2; i < b2 + 1; ++i) {
                const int idx = tid + 256 * i, row = idx >> 3, c = idx & 7;
                float x[8];
                widen8(load_frag(src + (size_t)row * Tn + c * 8), x);
                bf16x8 hi, lo;
                split8(x, hi, lo);
                bf16x4* dh = reinterpret_cast<bf16x4*>(buf + row * X3_VP + c * 16);
                dh[0] = bf16x4{hi[0], hi[1], hi[2], hi[3]};
                dh[1] = bf16x4{hi[4], hi[5], hi[6], hi[7]};
                bf16x4* dl = reinterpret_cast<bf16x4*>(buf + X3_PLANE + row * X3_VP + c * 16);
                dl[0] = bf16x4{lo[0], lo[1], lo[2], lo[3]};
                dl[1] = bf16x4{lo[4], lo[5], lo[6], lo[7]};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const float* qrow = q + ((size_t)n * Tn + q0 + r) * D + 8 * h;
    f32x16 st[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[kt][i] = 0.f;

    stage_k(0, smem);
    __syncthreads();
    Frag8<bf16x3> qf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const char* cur = smem + (s & 1) * X3_BUF;
        if ((s & 3) == 0) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) qf[kk] = load_op<bf16x3>(qrow + (s >> 2) * 128 + kk * 16);
        }
        if (s + 1 < 8) stage_k(s + 1, smem + ((s + 1) & 1) * X3_BUF);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const char* p = cur + (sub * 32 + r) * X3_KP + (kk * 16 + 8 * h) * 2;
                Frag8<bf16x3> kf;
                kf.hi = *reinterpret_cast<const bf16x8*>(p);
                kf.lo = *reinterpret_cast<const bf16x8*>(p + X3_PLANE);
                mma16(st[(s & 3) * 2 + sub], kf, qf[kk]);
            }
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- softmax over keys (registers + the partner half-wave), as in attention_kernel ----------------------------
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) m = fmaxf(m, st[kt][i]);
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __builtin_amdgcn_exp2f(1.44269504088896341f * 0.0625f * (st[kt][i] - m));
            st[kt][i] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32);
    const float inv = __builtin_amdgcn_rcpf(sum);
    Frag8<bf16x3> pf[8][2];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = st[kt][8 * s2 + j] * inv;
            split8(v, pf[kt][s2].hi, pf[kt][s2].lo);
        }
        __builtin_amdgcn_sched_barrier(0);  // tile by tile: 16 accumulators become 16 operand registers
    }

    // ---- O = P V over 64-key tiles of V^T, 128 output dims (four 32-wide tiles) per sweep -----------------------------------
    float* obase = out + ((size_t)n * Tn + q0) * D + r;
    stage_v(0, smem);  // buffer 0 is free: the last S^T step used buffer 1 and ended with a barrier
    __syncthreads();
    f32x16 o[4];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const char* cur = smem + (s & 1) * X3_BUF;
        if ((s & 3) == 0) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
        }
        if (s + 1 < 8) stage_v(s + 1, smem + ((s + 1) & 1) * X3_BUF);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    // element j <-> key 32*sub + 16*s2 + 8*(j>>2) + 4h + (j&3) of the tile
                    const char* p = cur + (d * 32 + r) * X3_VP + (sub * 32 + 16 * s2 + 4 * h) * 2;
                    const bf16x4 a0 = *reinterpret_cast<const bf16x4*>(p), a1 = *reinterpret_cast<const bf16x4*>(p + 16);
                    const bf16x4 b0 = *reinterpret_cast<const bf16x4*>(p + X3_PLANE), b1 = *reinterpret_cast<const bf16x4*>(p + X3_PLANE + 16);
                    Frag8<bf16x3> vf;
                    vf.hi = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    vf.lo = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    mma16(o[d], pf[(s & 3) * 2 + sub][s2], vf);
                }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if ((s & 3) == 3) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) obase[(size_t)acc_row(i, h) * D + ((s >> 2) * 4 + d) * 32] = o[d][i];
        }
    }
}

}  // namespace

int launch_attention(int dtype, const void* q, const void* k, const void* vt, void* out, int B, int T, hipStream_t s) {
    if (T != 256 && T != 64) return (int)hipErrorInvalidValue;
    const int nt = T / 32;
    const int waves = B * nt;
    dim3 grid((waves + 3) / 4), block(256);
    if (dtype == 1 && nt == 8) {
        static bool attr_done_dev[16] = {};
        const int dev = fg_device_slot();
        if (dev < 0) return (int)hipErrorInvalidDevice;
        bool& attr_done = attr_done_dev[dev];
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_BUF);
            if (e != hipSuccess) return (int)e;
            attr_done = true;
        }
        if (!q) return 0;  // prepare-only call (sets the attribute outside stream capture)
        hipLaunchKernelGGL(attention_lds_kernel, dim3(B * 2), block, 2 * ATT_BUF, s, (const __bf16*)q, (const __bf16*)k, (const __bf16*)vt, (__bf16*)out, B);
        return (int)hipGetLastError();
    }
    static const bool x3_lds = [] {
        const char* e = getenv("FASTGEN_AMD_ATTN_X3_LDS");  // 0: the direct-from-global kernel (A/B measurements)
        return !(e && e[0] == '0');
    }();
    if (dtype == 2 && nt == 8 && x3_lds) {
        static bool x3_attr_dev[16] = {};
        const int dev = fg_device_slot();
        if (dev < 0) return (int)hipErrorInvalidDevice;
        if (!x3_attr_dev[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_x3_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * X3_BUF);
            if (e != hipSuccess) return (int)e;
            x3_attr_dev[dev] = true;
        }
        if (!q) return 0;
        hipLaunchKernelGGL(attention_x3_lds_kernel, dim3(B * 2), block, 2 * X3_BUF, s, (const float*)q, (const float*)k, (const float*)vt, (float*)out, B);
        return (int)hipGetLastError();
    }
    if (!q) return 0;  // prepare-only call: nothing to set for the direct-from-global kernels
    if (dtype == 2) {  // split-bf16 products on fp32 tensors
        if (nt == 8)
            hipLaunchKernelGGL((attention_kernel<bf16x3, 8>), grid, block, 0, s, (const float*)q, (const float*)k, (const float*)vt, (float*)out, B);
        else
            hipLaunchKernelGGL((attention_kernel<bf16x3, 2>), grid, block, 0, s, (const float*)q, (const float*)k, (const float*)vt, (float*)out, B);
    } else if (dtype) {
        if (nt == 8)
            hipLaunchKernelGGL((attention_kernel<__bf16, 8>), grid, block, 0, s, (const __bf16*)q, (const __bf16*)k, (const __bf16*)vt, (__bf16*)out, B);
        else
            hipLaunchKernelGGL((attention_kernel<__bf16, 2>), grid, block, 0, s, (const __bf16*)q, (const __bf16*)k, (const __bf16*)vt, (__bf16*)out, B);
    } else {
        if (nt == 8)
            hipLaunchKernelGGL((attention_kernel<float, 8>), grid, block, 0, s, (const float*)q, (const float*)k, (const float*)vt, (float*)out, B);
        else
            hipLaunchKernelGGL((attention_kernel<float, 2>), grid, block, 0, s, (const float*)q, (const float*)k, (const float*)vt, (float*)out, B);
    }
    return (int)hipGetLastError();
}
