// Causal video DiT (SURVEY 8(f)3 / 8(a) "CausVid"): the pieces of the reference's CausalWan forward that are not GEMMs
// (fastgen/networks/Wan/network_causal.py; the block linears run on gemm.hip).  bf16 token tensors, fp32 statistics / softmax /
// modulation: the arithmetic the reference runs this network in (`precision = "bfloat16"`, configs/experiments/WanT2V/config_sf.py:19;
// the norms and the adaLN arithmetic in fp32: `_wan_block_forward_inline_cache`, :467-550).
//   wan_patch_embed_kernel   Conv3d(kernel = stride = (1,2,2)) + bias, tokens ordered (frame, row, column)
//   wan_mod_kernel           {shift, scale, gate} x {attention, feed-forward} = scale_shift_table + time_proj(silu(temb)) per frame (:478-480)
//   wan_rope_table_kernel    cos / sin of the chunk's tokens with the temporal offset of `_rope_forward_with_time_offset` (:79-128)
//   rms_rope_kernel          RMSNorm over the full inner dimension ("rms_norm_across_heads") * weight, interleaved-pair RoPE
//                            (`apply_rotary_emb`, :274-289), scattered into the KV cache rows of the chunk (:386-390); also the plain copy of v
//   fa128_kernel             attention of the chunk's queries over the cached frames + the chunk (:377-412) and over the text (:331-360):
//                            head dim 128, any number of keys, online softmax
//   wan_final_kernel         per-frame output modulation + proj_out + un-patchify (fastgen/networks/Wan/network.py:226-262)
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "misc.h"

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The same for 16 latent channels (K = 64 inputs per token, every Wan 2.1 network): a workgroup stages 32 tokens' inputs in LDS
// (8 KB) and every thread keeps the weight rows of TWO adjacent outputs in registers while it walks the 32 tokens (broadcast LDS
// reads): each weight row is fetched once per 32 tokens instead of once per token - the per-output form below spent a 64-term dot
// product's worth of scattered 4-byte loads on every output (412 us per call of the 1.3B network at 480p).
__global__ __launch_bounds__(256) void wan_patch_embed16_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                __bf16* __restrict__ out, int B, int Fr, int H, int W, int D) {
    constexpr int C = 16, K = 64, TOK = 32;
    __shared__ __attribute__((aligned(16))) float xs[TOK][K];
    const int gh = H / 2, gw = W / 2, fs = gh * gw;
    const int64_t ntok = (int64_t)B * Fr * fs, tok0 = (int64_t)blockIdx.x * TOK;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < TOK * K / 256; ++i) {
        const int idx = tid + 256 * i, tl = idx >> 6, k = idx & 63, c = k >> 2, py = (k >> 1) & 1, px = k & 1;
        const int64_t tok = tok0 + tl < ntok ? tok0 + tl : ntok - 1;
        const int hw = (int)(tok % fs), f = (int)((tok / fs) % Fr), b = (int)(tok / ((int64_t)fs * Fr));
        const int gy = hw / gw, gx = hw - gy * gw;
        xs[tl][k] = x[((((size_t)b * C + c) * Fr + f) * H + 2 * gy + py) * W + 2 * gx + px];
    }
    __syncthreads();
    for (int d = 2 * tid; d < D; d += 512) {
        f32x4 w0[16], w1[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            w0[q] = *reinterpret_cast<const f32x4*>(w + (size_t)d * K + 4 * q);
            w1[q] = *reinterpret_cast<const f32x4*>(w + (size_t)(d + 1) * K + 4 * q);
        }
        const float b0 = bias[d], b1 = bias[d + 1];
        for (int t = 0; t < TOK && tok0 + t < ntok; ++t) {
            float a0 = b0, a1 = b1;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[t][4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) a0 = fmaf(w0[q][e], xv[e], a0), a1 = fmaf(w1[q][e], xv[e], a1);
            }
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
            *reinterpret_cast<bf16x2_*>(out + (size_t)(tok0 + t) * D + d) = bf16x2_{(__bf16)a0, (__bf16)a1};
        }
    }
}

// x [B][C][F][H][W] fp32 -> tokens [B][F * gh * gw][D] bf16
__global__ __launch_bounds__(256) void wan_patch_embed_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                              __bf16* __restrict__ out, int B, int C, int Fr, int H, int W, int D) {
    const int gh = H / 2, gw = W / 2, fs = gh * gw;
    const int64_t total = (int64_t)B * Fr * fs * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int d = (int)(i % D);
        const int64_t tok = i / D;
        const int hw = (int)(tok % fs), f = (int)((tok / fs) % Fr), b = (int)(tok / ((int64_t)fs * Fr));
        const int gy = hw / gw, gx = hw - gy * gw;
        float a = bias[d];
        const float* wr = w + (size_t)d * C * 4;
        for (int c = 0; c < C; ++c) {
            const float* xp = x + ((((size_t)b * C + c) * Fr + f) * H + 2 * gy) * W + 2 * gx;
            a = fmaf(wr[c * 4 + 0], xp[0], a);
            a = fmaf(wr[c * 4 + 1], xp[1], a);
            a = fmaf(wr[c * 4 + 2], xp[W], a);
            a = fmaf(wr[c * 4 + 3], xp[W + 1], a);
        }
        out[i] = (__bf16)a;
    }
}

// mod[bf][j][d] = table[j][d] + tproj[bf][j * D + d]
__global__ void wan_mod_kernel(const float* __restrict__ table, const float* __restrict__ tproj, float* __restrict__ mod, int rows, int J, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * J * D) return;
    mod[i] = table[i % ((int64_t)J * D)] + tproj[i];
}
// out[r][0][d] = table[0][d] + temb[r][d], out[r][1][d] = table[1][d] + temb[r][d]   (the output layer's {shift, scale})
__global__ void wan_outmod_kernel(const float* __restrict__ table, const float* __restrict__ temb, float* __restrict__ mod, int rows, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * 2 * D) return;
    const int d = (int)(i % D);
    const int64_t r = i / (2 * (int64_t)D);
    mod[i] = table[i % (2 * (int64_t)D)] + temb[r * D + d];
}
__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] / (1.0f + expf(-x[i]));
}
__global__ void cvt_rows_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (__bf16)x[i];
}

// cs[l][i] = {cos, sin} of pair i (0 .. hd / 2) of chunk token l = (f, gy, gx): pairs [0, nt) take the frame axis at min(start + f, S - 1),
// [nt, nt + nh) the row axis, the rest the column axis; tab: per axis [S][pairs of the axis] {cos, sin} (fp32 of float64 angles, host-built)
__global__ void wan_rope_table_kernel(const float2* __restrict__ tab_t, const float2* __restrict__ tab_h, const float2* __restrict__ tab_w,
                                      float2* __restrict__ cs, int L, int fs, int gw, int start, int S, int nt, int nh, int nw) {
    const int half = nt + nh + nw;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L * half) return;
    const int l = i / half, p = i - l * half;
    const int f = l / fs, hw = l - f * fs, gy = hw / gw, gx = hw - gy * gw;
    float2 v;
    if (p < nt) v = tab_t[(size_t)min(start + f, S - 1) * nt + p];
    else if (p < nt + nh) v = tab_h[(size_t)gy * nh + (p - nt)];
    else v = tab_w[(size_t)gx * nw + (p - nt - nh)];
    cs[i] = v;
}

// One wave per row: y = x * rsqrt(mean(x^2) + eps) * w (w == nullptr: plain copy), then (cs != nullptr) the interleaved-pair rotation
// of every head with the row's table entry cs[row % L]; row r = (batch r / L, token r % L) goes to dst + batch * dst_bs + (dst_row0 +
// token) * ld_dst.  D = 128 NP, lane l holds the pairs l + 64 j of the row (element pair 2 (l + 64 j), +1): pair index inside its
// head = (l + 64 j) % 64 = l.
template <int NP>
__global__ __launch_bounds__(256) void rms_rope_kernel(const __bf16* __restrict__ src, int ld_src, const float* __restrict__ w, float eps,
                                                       const float2* __restrict__ cs, __bf16* __restrict__ dst, int64_t dst_bs, int dst_row0,
                                                       int ld_dst, int rows, int L) {
    constexpr int D = 128 * NP;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const __bf16* x = src + (size_t)row * ld_src;
    float v[2 * NP];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const bf16x2 q = *reinterpret_cast<const bf16x2*>(x + 2 * (lane + 64 * j));
        v[2 * j] = (float)q[0], v[2 * j + 1] = (float)q[1];
        ss = fmaf(v[2 * j], v[2 * j], fmaf(v[2 * j + 1], v[2 * j + 1], ss));
    }
    const int b = row / L, l = row - b * L;
    if (w) {
        const float r = 1.0f / sqrtf(wsum(ss) * (1.0f / D) + eps);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const f32x2 wq = *reinterpret_cast<const f32x2*>(w + 2 * (lane + 64 * j));
            v[2 * j] *= r * wq[0], v[2 * j + 1] *= r * wq[1];
        }
    }
    if (cs) {
        const float2 c = cs[(size_t)l * 64 + lane];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const float a = v[2 * j], bb = v[2 * j + 1];
            v[2 * j] = a * c.x - bb * c.y;
            v[2 * j + 1] = a * c.y + bb * c.x;
        }
    }
    __bf16* y = dst + (size_t)b * dst_bs + (size_t)(dst_row0 + l) * ld_dst;
#pragma unroll
    for (int j = 0; j < NP; ++j) *reinterpret_cast<bf16x2*>(y + 2 * (lane + 64 * j)) = bf16x2{(__bf16)v[2 * j], (__bf16)v[2 * j + 1]};
}

// ---- attention, head dim 128 ------------------------------------------------------------------------------------------------
// grid (ceil(Lq / 128), heads, batch), 256 threads: a wave owns 32 queries end to end; the workgroup walks the keys in tiles of
// 32, each tile's K and V rows (32 x 128 bf16 = 8 KiB each) staged once in LDS for its four waves (double-buffered; the loads of
// tile t + 1 are in flight during tile t).  S^T = K Q^T (keys on the accumulator rows, the QUERY on the lane) and
// O^T = V^T P^T (output dims on the rows, the query on the lane): the online-softmax state of a query is one lane's scalars and
// the P^T accumulators feed the second product directly as its B operand.  V stays row-major in memory and in LDS; its
// K-contiguous fragments (8 keys of one output dim) come from ds_read_b64_tr_b16.  LDS image of both tiles: 256-byte rows,
// 16-byte chunk ch of row r at 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))): conflict-free for the ds_read_b128 row reads of K and
// for the transposed reads of V (cdna_hip_programming.md T10, image (b)).
__device__ __forceinline__ int fa_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ s16x4 fa_tr_read(const char* lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(uint32_t)(uintptr_t)lds_addr);
}

// Timing experiments (scripts/fa_ablate.sh) skip one ingredient of the pipelined step each and compute garbage: they exist only in the
// separately built libfastgen_amd_timing.so (`make timing`, -DFG_TIMING_BUILD); the product kernel has neither the argument nor the tests.
#ifndef FA2_EXP
#define FA2_EXP 0  // fa2_kernel timing experiments (scripts/fa2_exp.sh builds one library per bit: results are garbage): 0 in every shipped build
#endif
#ifdef FG_TIMING_BUILD
#define FA_ABL_PARAM , int abl
#define FA_ABL_ARG , abl
#define FA_ABL(bit) (abl & (bit))
#else
#define FA_ABL_PARAM
#define FA_ABL_ARG
#define FA_ABL(bit) false
#endif

// HD = 128 (the video DiT) or 72 (DiT-XL/2, 256 tokens: timm Attention as DiTBlock uses it, fastgen/networks/DiT/network.py:168, 191).
// HD = 72: the K tile's LDS rows are 144 bytes apart (9 x 16 B: 16 consecutive rows at one chunk fall on 16 different 16-byte slots
// without a swizzle), the V tile's 192; the fifth 16-deep step of the q k^T contraction is half empty (its upper half is zero on the q side, finite row
// spill-over on the k side), the third 32-wide tile of output dims is computed from spill-over and only its first 8 dims stored.
template <int HD, int MINW, bool FA_DMA>
__global__ __launch_bounds__(256, MINW) void fa_kernel(const __bf16* __restrict__ q, int ldq, int64_t q_bs, const __bf16* __restrict__ k,
                                                       const __bf16* __restrict__ v, int ldk, int64_t kv_bs, __bf16* __restrict__ out, int ldo,
                                                       int64_t o_bs, int Lq, int Lkv, float scale_log2e, int nsplit, float* __restrict__ po,
                                                       float* __restrict__ plse, int heads, int batch, int sample_major FA_ABL_PARAM) {
    constexpr int KS = (HD + 15) / 16;   // 16-deep steps of q k^T
    constexpr int DT_ = (HD + 31) / 32;  // 32-wide tiles of output dims
    constexpr int RP = HD * 2;           // LDS row pitch of the K tile
    // ... of the V tile: head dim 72 takes 192 bytes - the four rows of a transposed-read block (64 bytes each, two 16-lane groups)
    // then tile the 256-byte bank window (0, 192, 128, 64); at the K tile's 144 they overlapped pairwise (profiles/r02_dit_attn_pmc:
    // 1.1 conflict cycles per LDS instruction).  The third output tile's columns 72-95 read the row's padding: never stored.
    constexpr int VP = HD == 128 ? RP : 192;
    constexpr int TILE = 32 * RP;        // bytes of the K tile
    constexpr int BUF = TILE + 32 * VP;  // a buffer = K tile | V tile
    constexpr int CPR = HD / 8;          // 16-byte chunks per row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto off = [](int row, int ch) { return HD == 128 ? fa_off(row, ch) : row * RP + ch * 16; };
    auto offv = [](int row, int ch) { return HD == 128 ? fa_off(row, ch) : row * VP + ch * 16; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // Workgroup -> (unit, query tile), unit = (sample, head, key split): the query tiles of a unit read the same K / V rows, so a unit
    // lives on ONE XCD (workgroup w of a 1-D grid runs on XCD w % 8) and its tiles take neighbouring dispatch slots there - they walk
    // the keys together and all but the first read of a K / V tile hit that XCD's L2.  (With the query tile on blockIdx.x every head
    // was streamed through all eight L2s: 1.2 GB fetched per launch for 0.2 GB of K / V at the video DiT's chunk 6.)
    // With nsplit > 1 the workgroup covers the key tiles [t0, t1) of its split and leaves a normalised partial output + the
    // log2-domain log-sum-exp of its keys; fa128_combine_kernel merges the splits.
    const int qtiles = (Lq + 127) / 128, units = batch * heads * nsplit;
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int qt = slot % qtiles;
    int split, head, b;
    if (sample_major) {
        // many samples of few tokens (DiT: 256 tokens, 16 heads x 72 in one [token][3 x 1152] tensor): ALL heads of a sample on one XCD, one
        // after the other.  A head's 144-byte rows straddle the 128-byte lines of its neighbours' rows: dealt head by head to the eight
        // L2s every line was fetched by two or three of them (1.5 GB per launch for 0.6 GB of q / k / v / out at B = 256).
        const int u = slot / qtiles;
        b = (u / heads) * 8 + xcd, head = u % heads, split = 0;
        if (b >= batch) return;
    } else {
        const int unit = (slot / qtiles) * 8 + xcd;
        if (unit >= units) return;  // (the grid is padded to a multiple of 8 units; uniform per workgroup, before any barrier)
        split = unit % nsplit;
        const int bh = unit / nsplit;
        head = bh % heads, b = bh / heads;
    }
    const int q0 = qt * 128 + wave * 32;
    // this lane's query row (clamped: rows past Lq compute on the last row and are not stored)
    const __bf16* qrow = q + (size_t)b * q_bs + (size_t)min(q0 + r, Lq - 1) * ldq + head * HD + 8 * h;
    bf16x8 qf[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        qf[kk] = bf16x8{};
        if (kk * 16 + 8 * h < HD) qf[kk] = *reinterpret_cast<const bf16x8*>(qrow + kk * 16);
    }
    const __bf16* kb = k + (size_t)b * kv_bs + head * HD;
    const __bf16* vb = v + (size_t)b * kv_bs + head * HD;

    // staging: 32 rows x CPR chunks of both tiles over 256 threads
    constexpr int NPC = (32 * CPR + 255) / 256;
    // two register sets: the loads of tile t + 2 are issued while tile t is computed and tile t + 1 (issued an iteration earlier) waits
    // for its turn to be written to LDS - one tile of prefetch left an iteration as long as a loaded L2 / HBM round trip
    constexpr int PD = HD == 128 ? 2 : 1;  // tiles of prefetch (head dim 72: 256 keys = 8 tiles per query tile, one is enough)
    fg_u32x4 kreg[PD][NPC], vreg[PD][NPC];
    auto issue = [&](int t, auto SET_) {
        constexpr int SET = decltype(SET_)::value % PD;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int p = min(tid + 256 * i, 32 * CPR - 1);
            const size_t key = (size_t)min(t * 32 + p / CPR, Lkv - 1);
            kreg[SET][i] = *reinterpret_cast<const fg_u32x4*>(kb + key * ldk + (p % CPR) * 8);
            vreg[SET][i] = *reinterpret_cast<const fg_u32x4*>(vb + key * ldk + (p % CPR) * 8);
        }
    };
    auto park = [&](char* st, auto SET_) {
        constexpr int SET = decltype(SET_)::value % PD;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int p = tid + 256 * i;
            if (p < 32 * CPR) {
                *reinterpret_cast<fg_u32x4*>(st + off(p / CPR, p % CPR)) = kreg[SET][i];
                *reinterpret_cast<fg_u32x4*>(st + TILE + offv(p / CPR, p % CPR)) = vreg[SET][i];
            }
        }
    };
    // HD = 128: the tiles go global -> LDS by LDS-DMA (no VGPR round trip) into a ring of NST buffers, three tiles ahead of the one
    // being computed: with one tile of prefetch (registers, as HD = 72 below keeps it) an iteration lasted as long as a loaded
    // L2 / HBM round trip - 2800 cycles for 512 cycles of MFMA work.  A wave instruction fills 1 KiB = 4 rows of a tile; the
    // swizzle is applied on the lane's SOURCE address (LDS slot (row, pos) holds chunk pos ^ swizzle(row)).  Rows past Lkv are out of
    // the buffer resource's range and read as zeros.
    constexpr bool DMA = HD == 128 && FA_DMA;
    constexpr int NST = DMA ? 4 : 2;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    __amdgpu_buffer_rsrc_t rsK, rsV;
    unsigned dvo[2] = {0, 0};
    if constexpr (DMA) {
        const unsigned bytes = (unsigned)(((size_t)(Lkv - 1) * ldk + HD) * 2);
        rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(kb), 0, bytes, 0x00020000);
        rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(vb), 0, bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wave * 2 + j) * 4 + (lane >> 4), pos = lane & 15;
            dvo[j] = (unsigned)row * (unsigned)ldk * 2u + 16u * (unsigned)(pos ^ (((row & 3) << 2) | ((row >> 2) & 3)));
        }
    }
    auto dma = [&](int t, int stage) {  // tile t -> ring buffer `stage`; 4 instructions per wave
        const int soff = t * 32 * ldk * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr)(smem + stage * BUF + (wave * 2 + j) * 1024), 16, dvo[j], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr)(smem + stage * BUF + TILE + (wave * 2 + j) * 1024), 16, dvo[j], soff, 0, 0);
        }
    };

    f32x16 ot[DT_];
#pragma unroll
    for (int d = 0; d < DT_; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[d][i] = 0.f;
    float m = -INFINITY, lsum = 0.f;

    // transposed-read lane addresses inside a V tile: group (h, c = (lane >> 4) & 1), lane 4 q + p of the group supplies
    // off(r0 + q, c0 + (p >> 1)) + 8 (p & 1); r0 = 16 s + 4 h (+ 8), c0 = 4 dt + 2 c
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, gc = (lane >> 4) & 1;

    const int ntiles = (Lkv + 31) / 32;
    const int t0 = (int)((long long)split * ntiles / nsplit), nt = (int)((long long)(split + 1) * ntiles / nsplit);
    if constexpr (DMA) {
        // (tile indices past the split's end re-load its last tile into a buffer nobody reads: the waits below stay uniform)
        dma(t0, 0);
        dma(min(t0 + 1, nt - 1), 1);
        dma(min(t0 + 2, nt - 1), 2);
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");  // tiles t0 and t0 + 1 have landed, in every wave's part
    } else {
        issue(t0, std::integral_constant<int, 0>{});
        if (PD == 2 && t0 + 1 < nt) issue(t0 + 1, std::integral_constant<int, 1>{});
        park(smem, std::integral_constant<int, 0>{});
        __syncthreads();
    }
    // fragment addresses of this lane inside buffer 0 (the buffer, the K | V split and the 16-key half go into the instruction's
    // immediate offset: the tile loop below is unrolled over the two buffers, so no address arithmetic is left in it)
    const uint32_t s0 = (uint32_t)(uintptr_t)smem;
    uint32_t ka[KS], va[DT_][2];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) ka[kk] = s0 + (uint32_t)off(r, 2 * kk + h);
#pragma unroll
    for (int d = 0; d < DT_; ++d)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) va[d][hi] = s0 + (uint32_t)(offv(4 * h + 8 * hi + gq, 4 * d + 2 * gc + (gp >> 1)) + 8 * (gp & 1));
    auto step = [&](int t, auto BI_) {
        constexpr int BI = decltype(BI_)::value;  // tile t sits in buffer BI
        if (t + PD < nt) issue(t + PD, std::integral_constant<int, BI + PD>{});  // (two sets: set BI held tile t, in LDS since the last iteration)
        bf16x8 kf[KS];
        s16x4 vl[2][DT_], vh[2][DT_];
        if constexpr (HD == 128) {
            // every fragment read of the tile goes out first: this lane's K row (the q k^T chain then runs MFMA behind MFMA instead of
            // one LDS round trip per MFMA) and the transposed V blocks, which land behind the softmax arithmetic
            // (inline asm + a hand-counted wait: left to itself hipcc sinks each read to just before the MFMA that uses it)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[kk]) : "v"(ka[kk]), "n"(BI * BUF));
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int d = 0; d < DT_; ++d) {
                    // rows 16 sx + 4 h + gq (+ 8): the swizzle of a row depends on row & 15 only, so sx is a plain row offset
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vl[sx][d]) : "v"(va[d][0]), "n"(BI * BUF + TILE + 16 * VP * sx));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vh[sx][d]) : "v"(va[d][1]), "n"(BI * BUF + TILE + 16 * VP * sx));
                }
            // LDS returns in order: at most 15 reads outstanding = the K fragments (issued first) have landed
            asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) asm volatile("" : "+v"(kf[kk]));  // (the MFMAs below depend on the wait, not only on the reads)
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // head dim 72 (256 keys = 8 tiles per query tile): the reads stay compiler-visible (hipcc places each next to its MFMA);
            // fragments in flight bought nothing at this length (DiT-XL/2 forward 75.6 ms either way)
            const char* st = smem + BI * BUF;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) kf[kk] = *reinterpret_cast<const bf16x8*>(st + off(r, 2 * kk + h));
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int d = 0; d < DT_; ++d) {
                    const int r0 = 16 * sx + 4 * h, c0 = 4 * d + 2 * gc;
                    vl[sx][d] = fa_tr_read(st + TILE + offv(r0 + gq, c0 + (gp >> 1)) + 8 * (gp & 1));
                    vh[sx][d] = fa_tr_read(st + TILE + offv(r0 + 8 + gq, c0 + (gp >> 1)) + 8 * (gp & 1));
                }
        }
        // S^T[key][query]
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], s, 0, 0, 0);
        // online softmax; keys past Lkv (last tile) are masked
        float mt = -INFINITY;
        if (t == ntiles - 1) {  // (wave-uniform) only the last key tile can be ragged
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (t * 32 + acc_row(i, h) >= Lkv) s[i] = -INFINITY;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) mt = fmaxf(mt, s[i]);
        {  // the other half of the query's keys sits in lane ^ 32: v_permlane32_swap instead of a trip through LDS
            float a_ = mt, b_ = mt;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
            mt = fmaxf(a_, b_);
        }
        const float mn = fmaxf(m, mt);
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * scale_log2e);  // m = -inf at the first tile: exp2(-inf) = 0
        const float mc = mn * scale_log2e;  // (one fma per logit instead of a subtraction and a multiplication)
        // two logits per instruction (v_pk_fma_f32, v_pk_add_f32); the exponentials themselves are scalar quarter-rate instructions
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sc2 = {scale_log2e, scale_log2e}, mc2 = {mc, mc};
        f32x2 e2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x2 z = __builtin_elementwise_fma(f32x2{s[2 * i], s[2 * i + 1]}, sc2, -mc2);
            e2[i] = f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
        }
        const f32x2 p2 = ((e2[0] + e2[1]) + (e2[2] + e2[3])) + ((e2[4] + e2[5]) + (e2[6] + e2[7]));
        float ps = p2[0] + p2[1];
        {
            float a_ = ps, b_ = ps;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
            ps = a_ + b_;
        }
        lsum = lsum * alpha + ps;
        m = mn;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {  // wave-uniform: the running maximum of some query of this wave moved
#pragma unroll
            for (int d = 0; d < DT_; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) ot[d][i] *= alpha;
        }
        // O^T[dim][query] += V^T[dim][key] P^T[key][query]
        if constexpr (HD == 128) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int d = 0; d < DT_; ++d) asm volatile("" : "+v"(vl[sx][d]), "+v"(vh[sx][d]));
        }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) pf[2 * j] = (__bf16)e2[4 * sx + j][0], pf[2 * j + 1] = (__bf16)e2[4 * sx + j][1];
#pragma unroll
            for (int d = 0; d < DT_; ++d) {
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x4 lo = vl[sx][d], hi = vh[sx][d];
                const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, ot[d], 0, 0, 0);
            }
        }
        if (t + 1 < nt) park(smem + (1 - BI) * BUF, std::integral_constant<int, 1 - BI>{});
        __syncthreads();
    };
    if constexpr (DMA) {
        // ---- software-pipelined form: S^T of tile t + 1 (eight chained MFMAs) is issued in between the pieces of tile t's softmax
        // arithmetic - a wave issues in order, so the chain only overlaps the VALU work if the two are interleaved in program
        // order (one MFMA per ~40-60 cycles of VALU: never a stall on the chain) - then O += V P of tile t.  Tile t + 1 must have
        // landed when iteration t starts: one younger tile (4 DMA instructions) stays in flight over the hand-over barrier.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x16 s_a, s_b;  // S^T of the tile whose softmax comes next / of the tile after it: the two swap roles from step to step (no copy)
        {
            bf16x8 kf[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) asm volatile("ds_read_b128 %0, %1" : "=v"(kf[kk]) : "v"(ka[kk]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) asm volatile("" : "+v"(kf[kk]));
#pragma unroll
            for (int i = 0; i < 16; ++i) s_a[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) s_a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], s_a, 0, 0, 0);
        }
        auto pstep = [&](int t, auto BI_, f32x16& sc, f32x16& sn) {
            constexpr int BI = decltype(BI_)::value, BN = (BI + 1) % NST;  // tile t sits in buffer BI, tile t + 1 in BN
            if (!FA_ABL(4)) dma(min(t + 3, nt - 1), (BI + 3) % NST);  // into tile t - 1's buffer: its readers passed the last barrier
            // K rows of tile t + 1 (past the split's end: a re-loaded last tile, result unused), then the V blocks of tile t
            bf16x8 kf[KS];
            s16x4 vl[2][DT_], vh[2][DT_];
            if (!FA_ABL(2)) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[kk]) : "v"(ka[kk]), "n"(BN * BUF));
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int d = 0; d < DT_; ++d) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vl[sx][d]) : "v"(va[d][0]), "n"(BI * BUF + TILE + 16 * VP * sx));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vh[sx][d]) : "v"(va[d][1]), "n"(BI * BUF + TILE + 16 * VP * sx));
                }
            }
            // piece 0: mask of a ragged last tile, this lane's maximum
            float mt = -INFINITY;
            if (t == ntiles - 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (t * 32 + acc_row(i, h) >= Lkv) sc[i] = -INFINITY;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) mt = fmaxf(mt, sc[i]);
            asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");  // in-order returns: the 8 K fragments (issued first) have landed
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) asm volatile("" : "+v"(kf[kk]));
            __builtin_amdgcn_sched_barrier(0);
            {  // (C = the inline constant 0: no sixteen v_mov to clear the accumulator)
                f32x16 z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.f;
                sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], z, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // piece 1: the other half of the query's keys (lane ^ 32), running maximum, rescale factor
            {
                float a_ = mt, b_ = mt;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
                mt = fmaxf(a_, b_);
            }
            const float mn = fmaxf(m, mt);
            const float alpha = __builtin_amdgcn_exp2f((m - mn) * scale_log2e);
            const float mc = mn * scale_log2e;
            const f32x2 sc2 = {scale_log2e, scale_log2e}, mc2 = {mc, mc};
            __builtin_amdgcn_sched_barrier(0);
            sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[1], sn, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // pieces 2-5: four exponentials each
            f32x2 e2[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int i = 2 * c; i < 2 * c + 2; ++i) {
                    const f32x2 z = __builtin_elementwise_fma(f32x2{sc[2 * i], sc[2 * i + 1]}, sc2, -mc2);
                    e2[i] = FA_ABL(1) ? z : f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
                }
                __builtin_amdgcn_sched_barrier(0);
                sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 + c], qf[2 + c], sn, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // piece 6: row sums
            const f32x2 p2 = ((e2[0] + e2[1]) + (e2[2] + e2[3])) + ((e2[4] + e2[5]) + (e2[6] + e2[7]));
            float ps = p2[0] + p2[1];
            {
                float a_ = ps, b_ = ps;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
                ps = a_ + b_;
            }
            lsum = lsum * alpha + ps;
            m = mn;
            __builtin_amdgcn_sched_barrier(0);
            sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[6], qf[6], sn, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // piece 7: P^T fragments; the (rare) rescale of O
            bf16x8 pf[2];
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int j = 0; j < 4; ++j) pf[sx][2 * j] = (__bf16)e2[4 * sx + j][0], pf[sx][2 * j + 1] = (__bf16)e2[4 * sx + j][1];
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                for (int d = 0; d < DT_; ++d)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ot[d][i] *= alpha;
            }
            __builtin_amdgcn_sched_barrier(0);
            sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[7], qf[7], sn, 0, 0, 0);
            // O^T[dim][query] += V^T[dim][key] P^T[key][query]
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (the two halves of an operand are joined BEFORE the tie to the wait: the register allocator can then give the two reads
            // the halves of one 4-register tuple instead of copying them together in front of every MFMA)
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 vv[2][DT_];
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int d = 0; d < DT_; ++d) {
                    const s16x4 lo = vl[sx][d], hi = vh[sx][d];
                    vv[sx][d] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    asm volatile("" : "+v"(vv[sx][d]));
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int d = 0; d < DT_; ++d)
                    if (!FA_ABL(16)) ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv[sx][d]), pf[sx], ot[d], 0, 0, 0);
            // one younger tile (4 instructions) may stay in flight: tile t + 2 has landed; then the hand-over barrier
            if (!FA_ABL(8)) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        };
        static_assert(!DMA || KS == 8, "the interleave above is written for eight contraction steps");
        for (int t = t0; t < nt; t += 4) {
            pstep(t, std::integral_constant<int, 0>{}, s_a, s_b);
            if (t + 1 < nt) pstep(t + 1, std::integral_constant<int, 1>{}, s_b, s_a);
            if (t + 2 < nt) pstep(t + 2, std::integral_constant<int, 2>{}, s_a, s_b);
            if (t + 3 < nt) pstep(t + 3, std::integral_constant<int, 3>{}, s_b, s_a);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may land in an LDS allocation this workgroup has given up
    } else {
        for (int t = t0; t < nt; t += 2) {
            step(t, std::integral_constant<int, 0>{});
            if (t + 1 < nt) step(t + 1, std::integral_constant<int, 1>{});
        }
    }
    // out[query][head * HD + dim]: this lane holds dims 32 d + acc_row(i, h) = 32 d + 8 i4 + 4 h + e of its query
    if (nsplit > 1) {
        if (q0 + r < Lq) {
            const float inv = 1.0f / lsum;
            const size_t row = (size_t)b * Lq + q0 + r, rows = (size_t)batch * Lq;
            float* orow = po + ((size_t)split * rows + row) * (heads * HD) + head * HD;
#pragma unroll
            for (int d = 0; d < DT_; ++d)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
                    if (32 * d + 8 * i4 + 4 * h < HD)
                        *reinterpret_cast<f32x4*>(orow + 32 * d + 8 * i4 + 4 * h) =
                            f32x4{ot[d][4 * i4] * inv, ot[d][4 * i4 + 1] * inv, ot[d][4 * i4 + 2] * inv, ot[d][4 * i4 + 3] * inv};
            if (h == 0) plse[((size_t)split * rows + row) * heads + head] = m * scale_log2e + __builtin_amdgcn_logf(lsum);
        }
        return;
    }
    if (q0 + r < Lq) {
        const float inv = 1.0f / lsum;
        __bf16* orow = out + (size_t)b * o_bs + (size_t)(q0 + r) * ldo + head * HD;
#pragma unroll
        for (int d = 0; d < DT_; ++d)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
                if (32 * d + 8 * i4 + 4 * h < HD) {
                    const bf16x4 o4 = {(__bf16)(ot[d][4 * i4] * inv), (__bf16)(ot[d][4 * i4 + 1] * inv), (__bf16)(ot[d][4 * i4 + 2] * inv),
                                       (__bf16)(ot[d][4 * i4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(orow + 32 * d + 8 * i4 + 4 * h) = o4;
                }
    }
}

// MFMAs with their register files pinned (hipcc, left to itself, keeps this kernel's 64 registers of Q fragments in accumulator
// registers and COPIES four of them into vector registers in front of every S^T MFMA: 13 v_accvgpr_* per MFMA in the tile loop):
// S^T accumulates in vector registers (the softmax reads it), the Q fragments are accumulator-file operands, O^T accumulates in the
// accumulator file.  Inline asm: the compiler pads no hazards here - every reader of a result below is dozens of instructions away
// (the S^T of tile t + 1 is first read by tile t + 1's softmax; O^T by the next tile's rare rescale or, behind an s_nop, by the epilogue).
__device__ __forceinline__ void fa2_mfma_s0(f32x16& d, const bf16x8& kf, const bf16x8& qf) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(kf), "a"(qf));
}
__device__ __forceinline__ void fa2_mfma_s(f32x16& d, const bf16x8& kf, const bf16x8& qf) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(kf), "a"(qf));
}
// (the last link of a chain: d = kf qf + c, so the finished S^T lands in the registers the softmax has just been done with)
__device__ __forceinline__ void fa2_mfma_sl(f32x16& d, const bf16x8& kf, const bf16x8& qf, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "+v"(d) : "v"(kf), "a"(qf), "v"(c));  // ("+": the old set, no copy at the loop's back edge)
}
template <typename VT>
__device__ __forceinline__ void fa2_mfma_o(f32x16& d, const VT& vv, const bf16x8& pf) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(vv), "v"(pf));
}

// ---- attention, head dim 128, 64 queries per wave ---------------------------------------------------------------------------------
// fa_kernel<128> is bound by instruction issue, not by the matrix pipe (profiles/r02_fa_ablation.txt): per 32-key tile a wave issues 16 MFMAs
// beside ~90 vector instructions, 24 LDS fragment reads and 4 LDS-DMA pieces, and the two waves of a SIMD share one issue port.  Here a wave
// owns TWO 32-query blocks (64 queries) and the whole 512-register file (one wave per SIMD, one 256-thread workgroup per CU): every K / V
// fragment read and every DMA piece feeds two MFMAs - 16 % less issue time per query (profiles/r03_fa2_experiments.txt, 2.).  The price is
// that nothing else fills this wave's stalls, so the step is laid out by hand (see "the pipeline" below): the O^T MFMAs lag the softmax
// by a tile and are dealt out, one per slot, between its pieces; fragments are re-read a step before their use; the reference maximum
// moves in eras with the rescale outside the tile loop; the reference check is speculative.  Same tiles and LDS image as fa_kernel
// (K tiles and V tiles in two rings of 6 x 8 KiB, LDS-DMA four tiles ahead); workgroup = 4 waves = 256 queries; grid mapping, key splits
// and the merge pass as fa_kernel.  Used for self-attention over >= 1024 keys (launch_fa).
__global__ __launch_bounds__(256, 1) void fa2_kernel(const __bf16* __restrict__ q, int ldq, int64_t q_bs, const __bf16* __restrict__ k,
                                                     const __bf16* __restrict__ v, int ldk, int64_t kv_bs, __bf16* __restrict__ out, int ldo,
                                                     int64_t o_bs, int Lq, int Lkv, float scale_log2e, int nsplit, float* __restrict__ po,
                                                     float* __restrict__ plse, int heads, int batch, int t_cut FA_ABL_PARAM) {
    constexpr int HD = 128, KS = 8, DT_ = 4, TILE = 32 * 256, NST = 6;
    constexpr int VBASE = NST * TILE;  // LDS: the ring's K tiles, then its V tiles
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int qtiles = (Lq + 255) / 256, units = batch * heads * nsplit;
    int split, bh, qt;
    if (t_cut > 0) {
        // Uneven two-way key split for grids of fewer workgroups than CUs (one sample of the video DiT: 12 heads x 19 query tiles = 228
        // for 256 CUs).  Every (head, query tile) is cut at key tile t_cut: the long pieces [0, t_cut) are dispatched first, one per CU,
        // and the CUs left over work through the short pieces [t_cut, ntiles) meanwhile (launch_fa sizes t_cut so that they just
        // manage) - the launch ends with the long pieces instead of after a whole extra round.  Pieces are dealt to the XCDs in
        // contiguous runs of the (head, tile) order: equal counts per XCD, a head's K / V in at most two L2s.
        const int P = batch * heads * qtiles, per = (P + 7) / 8;
        split = (int)blockIdx.x >= 8 * per ? 1 : 0;
        const int lb = (int)blockIdx.x - split * 8 * per, xcd = lb & 7, slot = lb >> 3;
        const int j = (int)((long long)xcd * P / 8) + slot;
        if (j >= (int)((long long)(xcd + 1) * P / 8)) return;  // (padding of the XCD's run; uniform per workgroup, before any barrier)
        bh = j / qtiles, qt = j - bh * qtiles;
    } else {
        const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        const int unit = (slot / qtiles) * 8 + xcd;
        qt = slot % qtiles;
        if (unit >= units) return;  // (the grid is padded to a multiple of 8 units; uniform per workgroup, before any barrier)
        split = unit % nsplit, bh = unit / nsplit;
    }
    const int head = bh % heads, b = bh / heads;
    const int q0 = qt * 256 + wave * 64;
    // this lane's two query rows (clamped: rows past Lq compute on the last row and are not stored)
    bf16x8 qfA[KS], qfB[KS];
    {
        const __bf16* qa = q + (size_t)b * q_bs + (size_t)min(q0 + r, Lq - 1) * ldq + head * HD + 8 * h;
        const __bf16* qb = q + (size_t)b * q_bs + (size_t)min(q0 + 32 + r, Lq - 1) * ldq + head * HD + 8 * h;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            qfA[kk] = *reinterpret_cast<const bf16x8*>(qa + kk * 16);
            qfB[kk] = *reinterpret_cast<const bf16x8*>(qb + kk * 16);
        }
    }
    const __bf16* kb = k + (size_t)b * kv_bs + head * HD;
    const __bf16* vb = v + (size_t)b * kv_bs + head * HD;
    const unsigned bytes = (unsigned)(((size_t)(Lkv - 1) * ldk + HD) * 2);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(kb), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(vb), 0, bytes, 0x00020000);
    unsigned dvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 4 + (lane >> 4), pos = lane & 15;
        dvo[j] = (unsigned)row * (unsigned)ldk * 2u + 16u * (unsigned)(pos ^ (((row & 3) << 2) | ((row >> 2) & 3)));
    }
    auto dma = [&](int t, int stage) {  // tile t -> ring buffer `stage`; 4 instructions per wave
        const int soff = t * 32 * ldk * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr)(smem + stage * TILE + (wave * 2 + j) * 1024), 16, dvo[j], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr)(smem + VBASE + stage * TILE + (wave * 2 + j) * 1024), 16, dvo[j], soff, 0, 0);
        }
    };
    f32x16 otA[DT_], otB[DT_];
#pragma unroll
    for (int d = 0; d < DT_; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) otA[d][i] = 0.f, otB[d][i] = 0.f;
    float mA = -INFINITY, lA = 0.f, mB = -INFINITY, lB = 0.f;  // (the row sums stay per lane - two lanes per query - and meet in the epilogue)
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, gc = (lane >> 4) & 1;
    const int ntiles = (Lkv + 31) / 32;
    const int t0 = t_cut > 0 ? (split ? t_cut : 0) : (int)((long long)split * ntiles / nsplit);
    const int nt = t_cut > 0 ? (split ? ntiles : t_cut) : (int)((long long)(split + 1) * ntiles / nsplit);
    dma(t0, 0);
    dma(min(t0 + 1, nt - 1), 1);
    dma(min(t0 + 2, nt - 1), 2);
    dma(min(t0 + 3, nt - 1), 3);
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");  // tiles t0 .. t0 + 2 have landed, in every wave's part
    const uint32_t s0 = (uint32_t)(uintptr_t)smem;
    uint32_t ka[KS], va[DT_][2];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) ka[kk] = s0 + (uint32_t)fa_off(r, 2 * kk + h);
#pragma unroll
    for (int d = 0; d < DT_; ++d)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) va[d][hi] = s0 + VBASE + (uint32_t)(fa_off(4 * h + 8 * hi + gq, 4 * d + 2 * gc + (gp >> 1)) + 8 * (gp & 1));

    // ---- the pipeline ---------------------------------------------------------------------------------------------------------------
    // With one wave per SIMD nothing else fills the gaps of this wave's MFMAs: a 32x32x16 MFMA occupies the matrix pipe for 32 cycles and
    // holds vector issue for 8 of them, so ~5-6 vector instructions per MFMA ride for free and a run of MFMAs without vector work (or of
    // vector work without MFMAs) is time lost (MI355X_MICROARCH.md, issue-cost rows).  Per tile a wave has 32 MFMAs and ~210 other
    // instructions; the step below deals them out evenly - 16 slots per block-phase, each ONE MFMA plus one group of softmax work:
    //   phase A (tile t): O_A^T += V(t-1)^T P_A(t-1)^T [8 MFMAs]  |  softmax of S_A(t) -> P_A(t)  |  S_A(t+1) = K(t+1) Q_A^T [8 chained MFMAs]
    //   phase B (tile t): the same for block B; in its last slots the V fragments of tile t are read for the next step.
    // So the O^T MFMAs lag the softmax by one tile (their operands are registers that phase's vector work does not touch), the chain for
    // the next tile's S^T accumulates in a spare register set and only its LAST MFMA writes the set the exponentials have just been done
    // with.  The K fragments of tile t + 2 are read at the END of step t (their registers are free once the chain's MFMAs are issued), a
    // whole step before the chain that uses them, and the V fragments of tile t likewise: no step waits on the LDS reads it issued itself.
    // LDS ring: V(t-1) (its reads may still be returning), tile t, t+1, t+2, t+3 (landed by the step's end), t+4 landing = 6 x 16 KiB.
    f32x16 sA, sB, sn;
    bf16x8 kf[KS];
    s16x8 vv[2][DT_];
    bf16x8 pfA[2], pfB[2];
    {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) asm volatile("ds_read_b128 %0, %1" : "=v"(kf[kk]) : "v"(ka[kk]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) asm volatile("" : "+v"(kf[kk]));
        fa2_mfma_s0(sA, kf[0], qfA[0]);
        fa2_mfma_s0(sB, kf[0], qfB[0]);
#pragma unroll
        for (int kk = 1; kk < KS; ++kk) {
            fa2_mfma_s(sA, kf[kk], qfA[kk]);
            fa2_mfma_s(sB, kf[kk], qfB[kk]);
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[kk]) : "v"(ka[kk]), "n"(TILE));  // K(t0 + 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the first step's counted waits assume a whole step's reads in flight)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {  // "tile t0 - 1": P = 0 against finite V
            pfA[sx] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}, pfB[sx] = pfA[sx];
#pragma unroll
            for (int d = 0; d < DT_; ++d) vv[sx][d] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // Softmax against a REFERENCE maximum that moves in eras, not per tile: p = 2^(c s - c ref), and while a tile's maximum stays below
    // ref + THR / c nothing is rescaled at all (p <= 2^THR: exact in fp32 sums, and bf16 keeps its 8 significant bits at any magnitude).
    // Only when a query's scores jump past that does its block start a new era: ref := the new maximum, O^T and the sum scaled by
    // 2^(c (old ref - new ref)).  That multiply is vector arithmetic on the accumulator file; written inside the tile loop it makes hipcc
    // keep O^T in vector registers across the loop and copy all 128 registers to and fro every tile.  So the era change sits OUTSIDE the
    // inner loop (the loop is left at the end of the step that saw the jump: the tile's own O^T MFMAs come a step later anyway), and
    // inside it O^T is touched by the MFMAs alone.  The row sums stay per lane (two lanes per query); they meet in the epilogue.
    constexpr float THR = 40.0f;
    const float thr = THR / scale_log2e;
    float cmA = 0.f, cmB = 0.f;                // -c ref
    float mthA = -INFINITY, mthB = -INFINITY;  // ref + THR / c: a tile maximum above it starts an era
    float alA = 1.0f, alB = 1.0f;
    bool evA = false, evB = false;  // (scalar: the whole wave leaves the loop or none of it)
    auto SB = [] { __builtin_amdgcn_sched_barrier(0); };
    // (a sched_barrier fences the instruction scheduler only: LLVM's sinking passes still move pure arithmetic down to its first use,
    // across any number of them.  A volatile asm that "uses" the value keeps its producer in the slot it was written in.)
    auto pin1 = [](auto& x) { asm volatile("" : "+v"(x)); };
    auto PIN = [&](auto&... x) { (pin1(x), ...); };
    auto tile_max = [&](float m0, float m1) {  // over both lanes of the query
        float a_, b_, mt;
        asm volatile("v_max_f32 %0, %2, %3\n\tv_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "=&v"(a_), "=&v"(b_) : "v"(m0), "v"(m1));
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(mt) : "v"(a_), "v"(b_));
        return mt;
    };
    auto new_era = [&](float mt, float& m, float& cm, float& mth, float& lsum, float& al, bool& ev) {
        const float mn = fmaxf(m, mt);
        ev = ev || __builtin_amdgcn_ballot_w64(m > -INFINITY) != 0;  // (a block's first tile: O^T and the sum are still zero)
        al = m > -INFINITY ? __builtin_amdgcn_exp2f((m - mn) * scale_log2e) : 1.0f;
        lsum *= al;
        m = mn, cm = -mn * scale_log2e, mth = mn + thr;
    };
    // LDS offsets of the ring buffers this step works on (scalars, rotated per tile; one v_add per fragment address)
    int oV = 0, oK = 2 * TILE, oD = 4 * TILE;  // tile t (its V, read for the next step) | tile t + 2 (K, likewise) | tile t + 4 (the DMA lands there)
    auto phase = [&](int t, auto PH_, f32x16& sc, const bf16x8 (&qf)[KS], f32x16 (&ot)[DT_], bf16x8 (&pf)[2], float& m, float& cm, float& mth,
                     float& lsum, float& al, bool& ev) {
        constexpr int PH = decltype(PH_)::value;
        auto PV = [&](int j) { if (!FA_ABL(16) && !(FA2_EXP & 32)) fa2_mfma_o(ot[j & 3], vv[j >> 2][j & 3], pf[j >> 2]); };
        auto CH = [&](int kk) {
            if (FA_ABL(32) || (FA2_EXP & 64)) return;
            if (kk == 0) fa2_mfma_s0(sn, kf[0], qf[0]);
            else if (kk < 7) fa2_mfma_s(sn, kf[kk], qf[kk]);
            else fa2_mfma_sl(sc, kf[7], qf[7], sn);  // lands in the set the exponentials are done with
        };
        auto piece = [&](int j) {  // one K and one V piece of tile t + 4 (into the buffer of tile t - 2: its readers passed the last barrier)
            if (FA_ABL(4) || (FA2_EXP & 2)) return;
            const int soff = min(t + 4, nt - 1) * 32 * ldk * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr)(smem + oD + (wave * 2 + j) * 1024), 16, dvo[j], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr)(smem + VBASE + oD + (wave * 2 + j) * 1024), 16, dvo[j], soff, 0, 0);
        };
        auto kread = [&](int kk) {  // K(t + 2) -> the next step's chains
            if (FA_ABL(2) || (FA2_EXP & 4)) return;
            const uint32_t ad = ka[kk] + (uint32_t)oK;
            asm volatile("ds_read_b128 %0, %1" : "=v"(kf[kk]) : "v"(ad));
        };
        auto vread = [&](int j) {  // V(t) -> the next step's O^T MFMAs
            if (FA_ABL(2) || (FA2_EXP & 4)) return;
            const int sx = j >> 2, d = j & 3;
            s16x4 lo, hi;
            const uint32_t ad0 = va[d][0] + (uint32_t)oV, ad1 = va[d][1] + (uint32_t)oV;
            if (sx == 0) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(ad0));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(ad1));
            } else {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(lo) : "v"(ad0));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(hi) : "v"(ad1));
            }
            vv[sx][d] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        // The 16 MFMAs of a phase, chain and O^T alternating (a chain link waits for its predecessor's result; the O^T ones are
        // independent).  Phase B re-reads each fragment one slot after the MFMA that used it last: the LDS returns them in the order
        // phase A of the next step consumes them, so that phase waits with falling counts, never for a read it has just issued.
        //   slot:  0    1    2    3    4    5    6    7    8    9    10   11   12   13   14   15
        //          CH0  PV0  CH1  PV1  CH2  PV2  CH3  PV3  PV4  CH4  PV5  CH5  PV6  CH6  PV7  CH7
        auto M = [&](auto SLOT_) {
            constexpr int S = decltype(SLOT_)::value;
            constexpr bool is_ch = S < 8 ? !(S & 1) : (S & 1);
            constexpr int idx = S < 8 ? (S >> 1) : (is_ch ? 4 + ((S - 9) >> 1) : 4 + ((S - 8) >> 1));
            if constexpr (PH == 0) {
                // LDS operations of the last step still allowed out when this slot's fragment is in (K: 1 operation, V: 2; slot order):
                // 24 less those up to and including this slot's
                constexpr int upto = S < 8 ? 3 * (S >> 1) + (is_ch ? 1 : 3) : 12 + (S == 8 ? 2 : 2 + 3 * ((S - 9) >> 1) + (is_ch ? 1 : 3));
                constexpr int after = 24 - upto;
                if constexpr (S == 0) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");  // (the counter's range; covers slots 0 .. 5)
                else if constexpr (S >= 6) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(after) : "memory");
            }
            if constexpr (is_ch) CH(idx);
            else PV(idx);
        };
        auto R = [&](auto SLOT_) {  // (phase B) the re-read of the fragment slot S's MFMA used
            constexpr int S = decltype(SLOT_)::value;
            constexpr bool is_ch = S < 8 ? !(S & 1) : (S & 1);
            constexpr int idx = S < 8 ? (S >> 1) : (is_ch ? 4 + ((S - 9) >> 1) : 4 + ((S - 8) >> 1));
            if constexpr (PH == 1) {
                if constexpr (is_ch) kread(idx);
                else vread(idx);
            }
        };
#define SL(n) std::integral_constant<int, n> {}
        if constexpr (PH == 0) asm volatile("s_nop 1" ::: "memory");  // (the P^T fragments were written by vector instructions)
        SB();
        // slots 0-2: the tile maximum over both lanes of the query; does the reference hold?
        M(SL(0));
        float m0, m1;
        asm volatile("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %0, %0, %8, %9"
                     : "=&v"(m0)
                     : "v"(sc[0]), "v"(sc[1]), "v"(sc[2]), "v"(sc[3]), "v"(sc[4]), "v"(sc[5]), "v"(sc[6]), "v"(sc[7]), "v"(sc[8]));
        SB();
        M(SL(1));
        R(SL(0));
        asm volatile("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7"
                     : "=&v"(m1)
                     : "v"(sc[9]), "v"(sc[10]), "v"(sc[11]), "v"(sc[12]), "v"(sc[13]), "v"(sc[14]), "v"(sc[15]));
        SB();
        M(SL(2));
        R(SL(1));
        // Does the reference hold?  Asked per LANE (the two lanes of a query see different keys of the tile; a ballot covers both), and
        // the answer is only looked at after the exponentials: they are computed against the old reference on speculation - a vector
        // compare feeding a scalar branch right away stalls the wave's one instruction stream for the compare's whole latency.
        float ml;
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(ml) : "v"(m0), "v"(m1));
        const bool jump = !(FA2_EXP & 8) && __builtin_amdgcn_ballot_w64(ml > mth) != 0;
        SB();
        // slots 3-10: p = 2^(c s + cm), two pairs per slot
        float e[16];
        auto EX = [&](int g) {
            const float z0 = __builtin_fmaf(sc[2 * g], scale_log2e, cm), z1 = __builtin_fmaf(sc[2 * g + 1], scale_log2e, cm);
            e[2 * g] = (FA_ABL(1) || (FA2_EXP & 16)) ? z0 : __builtin_amdgcn_exp2f(z0), e[2 * g + 1] = (FA_ABL(1) || (FA2_EXP & 16)) ? z1 : __builtin_amdgcn_exp2f(z1);
            PIN(e[2 * g], e[2 * g + 1]);
        };
        M(SL(3)), R(SL(2)), EX(0), SB();
        M(SL(4)), R(SL(3)), EX(1), SB();
        M(SL(5)), R(SL(4)), EX(2), SB();
        M(SL(6)), R(SL(5)), EX(3), SB();
        M(SL(7)), R(SL(6)), EX(4), SB();
        M(SL(8)), R(SL(7)), EX(5), SB();
        M(SL(9)), R(SL(8)), EX(6), SB();
        M(SL(10)), R(SL(9)), EX(7), SB();
        if (jump) {  // (rare, wave-uniform) a new era for this block: the tile's exponentials again, against the new reference
            new_era(tile_max(m0, m1), m, cm, mth, lsum, al, ev);
#pragma unroll
            for (int g = 0; g < 8; ++g) EX(g);
        }
        // slots 11-15: the row sum (per lane); the K and V pieces of tile t + 4 (phase A: this wave's first pair, phase B: its second)
        M(SL(11));
        R(SL(10));
        float a0 = e[0] + e[1], a1 = e[2] + e[3], a2 = e[4] + e[5], a3 = e[6] + e[7];
        PIN(a0, a1, a2, a3);
        SB();
        M(SL(12));
        R(SL(11));
        float a4 = e[8] + e[9], a5 = e[10] + e[11], a6 = e[12] + e[13], a7 = e[14] + e[15];
        PIN(a4, a5, a6, a7);
        piece(PH);
        SB();
        M(SL(13));
        R(SL(12));
        float b0 = a0 + a1, b1 = a2 + a3, b2 = a4 + a5, b3 = a6 + a7;
        PIN(b0, b1, b2, b3);
        SB();
        M(SL(14));
        R(SL(13));
        lsum += (b0 + b1) + (b2 + b3);
        PIN(lsum);
        SB();
        M(SL(15));  // (the chain's last link lands in the set the exponentials are done with)
        R(SL(14));
        // P^T(t) over P^T(t - 1), whose MFMAs are all issued
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) pf[0][k2] = (__bf16)e[k2];
        PIN(pf[0]);
        SB();
        R(SL(15));
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) pf[1][k2] = (__bf16)e[8 + k2];
        PIN(pf[1]);
        SB();
#undef SL
    };
    auto era = [&]() {  // (no barrier in here: the other waves just see this one arrive at the next one later)
        asm volatile("s_nop 15" ::: "memory");
        // One 16-register tile at a time, each in a basic block of its own (the conditions are opaque always-true scalars): hipcc
        // assigns a register file per value per BLOCK, so a straight-line sweep wants all 128 accumulators in vector registers at once
        // and spills everything that lives across it - the loop's fragment addresses among them, in the loop as well.
        int yes;
        asm volatile("s_mov_b32 %0, 1" : "=s"(yes));
#pragma unroll
        for (int d = 0; d < DT_; ++d) {
            if (yes) {
#pragma unroll
                for (int i = 0; i < 16; ++i) otA[d][i] *= alA;
                asm volatile("" : "+a"(otA[d]));
            }
            asm volatile("s_add_u32 %0, %0, 0" : "+s"(yes));
            if (yes) {
#pragma unroll
                for (int i = 0; i < 16; ++i) otB[d][i] *= alB;
                asm volatile("" : "+a"(otB[d]));
            }
            asm volatile("s_add_u32 %0, %0, 0" : "+s"(yes));
        }
        alA = alB = 1.0f;
        evA = evB = false;
    };
    const int t_end = (nt == ntiles && (Lkv & 31)) ? nt - 1 : nt;  // at most the very last tile is ragged: it is finished off below
    int t = t0;
    while (t < t_end) {
        for (; t < t_end;) {  // the hot loop: O^T is only ever an MFMA operand here
            phase(t, std::integral_constant<int, 0>{}, sA, qfA, otA, pfA, mA, cmA, mthA, lA, alA, evA);
            phase(t, std::integral_constant<int, 1>{}, sB, qfB, otB, pfB, mB, cmB, mthB, lB, alB, evB);
            // one younger tile (4 instructions) may stay in flight: tile t + 2 has landed; then the hand-over barrier
            if (FA_ABL(8) || ((FA2_EXP & 1) && (t & 1))) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            ++t;
            oV = oV == (NST - 1) * TILE ? 0 : oV + TILE, oK = oK == (NST - 1) * TILE ? 0 : oK + TILE, oD = oD == (NST - 1) * TILE ? 0 : oD + TILE;
            if (evA || evB) break;
        }
        if (evA || evB) era();  // (rare; the tile's own O^T MFMAs come with the next step, against the new reference)
    }
    // the O^T MFMAs of the last whole tile
    auto drain = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int d = 0; d < DT_; ++d) asm volatile("" : "+v"(vv[sx][d]));
        asm volatile("s_nop 1" ::: "memory");
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int d = 0; d < DT_; ++d) {
                fa2_mfma_o(otA[d], vv[sx][d], pfA[sx]);
                fa2_mfma_o(otB[d], vv[sx][d], pfB[sx]);
            }
    };
    drain();
    if (t_end < nt) {  // the ragged tile, unpipelined: its S^T came out of the last step's chains (rows past Lkv: zeros from the buffer bounds)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // every wave's pieces of it have landed
        auto soft_plain = [&](f32x16& sc, bf16x8 (&pf)[2], float& m, float& cm, float& mth, float& lsum, float& al, bool& ev) {
            asm volatile("s_nop 7" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (t_end * 32 + acc_row(i, h) >= Lkv) sc[i] = -INFINITY;
            float m0 = sc[0], m1 = sc[8];
#pragma unroll
            for (int i = 1; i < 8; ++i) m0 = fmaxf(m0, sc[i]), m1 = fmaxf(m1, sc[8 + i]);
            const float mt = tile_max(m0, m1);
            if (__builtin_amdgcn_ballot_w64(mt > mth) != 0) new_era(mt, m, cm, mth, lsum, al, ev);
            float ps = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float ei = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[i], scale_log2e, cm));
                ps += ei;
                pf[i >> 3][i & 7] = (__bf16)ei;
            }
            lsum += ps;
        };
        soft_plain(sA, pfA, mA, cmA, mthA, lA, alA, evA);
        soft_plain(sB, pfB, mB, cmB, mthB, lB, alB, evB);
        if (evA || evB) era();
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int d = 0; d < DT_; ++d) {
                s16x4 lo, hi;
                const uint32_t ad0 = va[d][0] + (uint32_t)oV, ad1 = va[d][1] + (uint32_t)oV;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(ad0), "n"(16 * 256 * sx));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(ad1), "n"(16 * 256 * sx));
                vv[sx][d] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        drain();
    }
    {  // the two lanes of a query hold the sums over their halves of each tile's keys
        float a_ = lA, b_ = lA, c_ = lB, d_ = lB;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c_), "+v"(d_));
        lA = a_ + b_, lB = c_ + d_;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // no DMA may land in an LDS allocation this workgroup has given up; the last O^T MFMAs have retired
    // out[query][head * HD + dim]: this lane holds dims 32 d + 8 i4 + 4 h + e of its two queries
    auto finish = [&](int qrow, const f32x16 (&ot)[DT_], float m, float lsum) {
        if (qrow >= Lq) return;
        const float inv = 1.0f / lsum;
        if (nsplit > 1) {
            const size_t row = (size_t)b * Lq + qrow, rows = (size_t)batch * Lq;
            float* orow = po + ((size_t)split * rows + row) * (heads * HD) + head * HD;
#pragma unroll
            for (int d = 0; d < DT_; ++d)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
                    *reinterpret_cast<f32x4*>(orow + 32 * d + 8 * i4 + 4 * h) =
                        f32x4{ot[d][4 * i4] * inv, ot[d][4 * i4 + 1] * inv, ot[d][4 * i4 + 2] * inv, ot[d][4 * i4 + 3] * inv};
            if (h == 0) plse[((size_t)split * rows + row) * heads + head] = m * scale_log2e + __builtin_amdgcn_logf(lsum);
            return;
        }
        __bf16* orow = out + (size_t)b * o_bs + (size_t)qrow * ldo + head * HD;
#pragma unroll
        for (int d = 0; d < DT_; ++d)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const bf16x4 o4 = {(__bf16)(ot[d][4 * i4] * inv), (__bf16)(ot[d][4 * i4 + 1] * inv), (__bf16)(ot[d][4 * i4 + 2] * inv),
                                   (__bf16)(ot[d][4 * i4 + 3] * inv)};
                *reinterpret_cast<bf16x4*>(orow + 32 * d + 8 * i4 + 4 * h) = o4;
            }
    };
    finish(q0 + r, otA, mA, lA);
    finish(q0 + 32 + r, otB, mB, lB);
}

// compile-time loop (the LDS instructions below take their offsets as immediates) and LDS reads the compiler may not move or merge
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int OFF>
__device__ __forceinline__ void lds_read_b128(bf16x8& v, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr(s16x4& v, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait() {  // at most N LDS operations still out (they return in order)
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// ---- attention, head dim 72, at most 256 keys: the whole key sequence in one pass ---------------------------------------------------
// DiT-XL/2's shape (256 tokens, 16 heads x 72: fastgen/networks/DiT/network.py:168, 191).  fa_kernel<72> walks its 8 key tiles through a
// two-buffer ring with a barrier per tile and an online softmax whose per-tile bookkeeping (maximum exchange, rescale factor, ballot)
// costs as much as the tile's exponentials: 16.8 vector instructions per MFMA, 18 % of the matrix pipe (profiles/r03_dit_attn_pmc_*).
// With so few keys none of that is needed: a workgroup brings ALL of K and V of its (sample, head) into LDS with 72 LDS-DMA
// instructions (144-byte rows back to back, exactly as they lie in memory: 2 x 36 KiB, two workgroups per CU), each wave computes its 32
// queries' S^T against all 256 keys (8 accumulator tiles = 128 registers), takes the exact row maximum once, exponentiates, sums, and
// multiplies by V.  One barrier per workgroup; no rescale exists.  The second workgroup of the CU computes while this one loads.
// The third 32-wide tile of output dims (72 .. 95) is computed from the 24 bytes that follow each 144-byte row (the next row's start:
// finite) and never stored; the fifth 16-deep step of q k^T is half empty on the q side (zeros) as in fa_kernel<72>.
__global__ __launch_bounds__(256, 2) void fa72_seq_kernel(const __bf16* __restrict__ q, int ldq, int64_t q_bs, const __bf16* __restrict__ k,
                                                          const __bf16* __restrict__ v, int ldk, int64_t kv_bs, __bf16* __restrict__ out, int ldo,
                                                          int64_t o_bs, int Lq, int Lkv, float scale_log2e, int heads, int batch) {
    constexpr int HD = 72, KS = 5, DT_ = 3, RP = 144, NKT = 8, MAT = 256 * RP;  // bytes of one matrix in LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];                // K | V | 64 bytes of slack behind V's last row
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // all heads of a sample on one XCD (fa_kernel's sample-major mapping: a head's 144-byte rows straddle its neighbours' cache lines)
    const int qtiles = (Lq + 127) / 128;
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int qt = slot % qtiles, u = slot / qtiles;
    const int b = (u / heads) * 8 + xcd, head = u % heads;
    if (b >= batch) return;  // (uniform per workgroup, before the barrier)
    const int q0 = qt * 128 + wave * 32;
    // this lane's query row (clamped: rows past Lq compute on the last row and are not stored)
    const __bf16* qrow = q + (size_t)b * q_bs + (size_t)min(q0 + r, Lq - 1) * ldq + head * HD + 8 * h;
    bf16x8 qf[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        qf[kk] = bf16x8{};
        if (kk * 16 + 8 * h < HD) qf[kk] = *reinterpret_cast<const bf16x8*>(qrow + kk * 16);
    }
    // (the query loads are issued before the DMA below: the counter that is waited on retires in order)
    const __bf16* kb = k + (size_t)b * kv_bs + head * HD;
    const __bf16* vb = v + (size_t)b * kv_bs + head * HD;
    {
        const unsigned bytes = (unsigned)(((size_t)(Lkv - 1) * ldk + HD) * 2);  // rows past Lkv are out of range: they read as zeros
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(kb), 0, bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(vb), 0, bytes, 0x00020000);
        // (a wave instruction fills 1 KiB = 64 consecutive 16-byte chunks of the row-major image; all of K first: the S^T phase
        // starts when K is in, V lands behind it)
        unsigned src[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int g = (wave * 9 + i) * 64 + lane, row = g / 9, c = g - 9 * row;
            src[i] = (unsigned)row * (unsigned)ldk * 2u + 16u * (unsigned)c;
        }
#pragma unroll
        for (int i = 0; i < ((FA2_EXP & 256) ? 0 : 9); ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr)(smem + (wave * 9 + i) * 1024), 16, src[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < ((FA2_EXP & 256) ? 0 : 9); ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr)(smem + MAT + (wave * 9 + i) * 1024), 16, src[i], 0, 0, 0);
    }
    if (tid < 4) *reinterpret_cast<fg_u32x4*>(smem + 2 * MAT + 16 * tid) = fg_u32x4{0, 0, 0, 0};  // (finite bytes behind V's last row)
    // the query fragments were loaded FIRST (below the DMA in program order would put them behind V in the in-order counter)
    // K is in (this wave's part; the 9 V pieces may still be out), then the workgroup's barrier (not __syncthreads(): it drains the counter)
    asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // S^T[key][query], all keys.  A tile's five K fragments are read two tiles ahead of their MFMAs (three register sets, counted waits:
    // left to itself hipcc puts each read right in front of the MFMA that uses it - an LDS round trip per MFMA)
    f32x16 s[NKT];
    const uint32_t ka = (uint32_t)(uintptr_t)smem + (uint32_t)(r * RP + 16 * h);
    bf16x8 kf[3][KS];
    static_for<0, 2>([&](auto T_) {
        constexpr int T = decltype(T_)::value;
        static_for<0, KS>([&](auto K_) { constexpr int K = decltype(K_)::value; lds_read_b128<32 * T * RP + 32 * K>(kf[T][K], ka); });
    });
    static_for<0, ((FA2_EXP & 512) ? 1 : NKT)>([&](auto T_) {
        constexpr int T = decltype(T_)::value;
        if constexpr (T + 2 < NKT)
            static_for<0, KS>([&](auto K_) { constexpr int K = decltype(K_)::value; lds_read_b128<32 * (T + 2) * RP + 32 * K>(kf[(T + 2) % 3][K], ka); });
        lds_wait<(T + 2 < NKT ? 2 : NKT - 1 - T) * KS>();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) asm volatile("" : "+v"(kf[T % 3][kk]));  // (the MFMAs depend on the wait, not only on the reads)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[T][i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) s[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[T % 3][kk], qf[kk], s[T], 0, 0, 0);
    });
    if (Lkv < 32 * NKT) {  // (uniform) keys past Lkv
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (32 * kt + acc_row(i, h) >= Lkv) s[kt][i] = -INFINITY;
    }
    // the exact row maximum (the query's keys lie in this lane and in lane ^ 32), p = 2^(c (s - max)), the row sum
    float m = s[0][0];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) m = fmaxf(m, s[kt][i]);
    {
        float a_ = m, b_ = m;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
        m = fmaxf(a_, b_);
    }
    const float mc = m * scale_log2e;
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
    bf16x8 pf[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][i], scale_log2e, -mc));
            ls[i & 3] += e;
            pf[kt][i >> 3][i & 7] = (__bf16)e;
        }
    float lsum = (ls[0] + ls[1]) + (ls[2] + ls[3]);
    {
        float a_ = lsum, b_ = lsum;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
        lsum = a_ + b_;
    }
    // O^T[dim][query] = V^T[dim][key] P^T[key][query]; transposed-read lane addresses as in fa_kernel: group (h, c = (lane >> 4) & 1),
    // lane 4 q + p of the group supplies row r0 + q, 16-byte chunk c0 + (p >> 1), half p & 1; r0 = 32 kt + 16 sx + 4 h (+ 8), c0 = 4 d + 2 c
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, gc = (lane >> 4) & 1;
    const uint32_t va = (uint32_t)(uintptr_t)smem + (uint32_t)(MAT + (4 * h + gq) * RP + 16 * (2 * gc + (gp >> 1)) + 8 * (gp & 1));
    f32x16 ot[DT_];
#pragma unroll
    for (int d = 0; d < DT_; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[d][i] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // V is in: this wave's part, then every wave's
    // the twelve transposed reads of a key tile (two 16-key steps x three dim tiles x two row blocks) go out a tile ahead of its six MFMAs
    s16x4 vl[2][2][DT_], vh[2][2][DT_];
    auto vread = [&](auto T_) {
        constexpr int T = decltype(T_)::value;
        static_for<0, 2>([&](auto X_) {
            constexpr int X = decltype(X_)::value;
            static_for<0, DT_>([&](auto D_) {
                constexpr int D = decltype(D_)::value;
                lds_read_tr<(32 * T + 16 * X) * RP + 64 * D>(vl[T & 1][X][D], va);
                lds_read_tr<(32 * T + 16 * X + 8) * RP + 64 * D>(vh[T & 1][X][D], va);
            });
        });
    };
    vread(std::integral_constant<int, 0>{});
    static_for<0, ((FA2_EXP & 512) ? 1 : NKT)>([&](auto T_) {
        constexpr int T = decltype(T_)::value;
        if constexpr (T + 1 < NKT) vread(std::integral_constant<int, T + 1>{});
        lds_wait<(T + 1 < NKT) ? 12 : 0>();
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int d = 0; d < DT_; ++d) {
                asm volatile("" : "+v"(vl[T & 1][sx][d]), "+v"(vh[T & 1][sx][d]));
                const s16x4 lo = vl[T & 1][sx][d], hi = vh[T & 1][sx][d];
                const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[T][sx], ot[d], 0, 0, 0);
            }
    });
    // out[query][head * HD + dim]: this lane holds dims 32 d + 8 i4 + 4 h + e of its query
    if (q0 + r < Lq) {
        const float inv = 1.0f / lsum;
        __bf16* orow = out + (size_t)b * o_bs + (size_t)(q0 + r) * ldo + head * HD;
#pragma unroll
        for (int d = 0; d < DT_; ++d)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
                if (32 * d + 8 * i4 + 4 * h < HD) {
                    const bf16x4 o4 = {(__bf16)(ot[d][4 * i4] * inv), (__bf16)(ot[d][4 * i4 + 1] * inv), (__bf16)(ot[d][4 * i4 + 2] * inv),
                                       (__bf16)(ot[d][4 * i4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(orow + 32 * d + 8 * i4 + 4 * h) = o4;
                }
    }
}

// out[row][c] = sum_s w_s po[s][row][c] / sum_s w_s,  w_s = 2^(plse[s][row][head(c)] - max_s): the merge of fa128_kernel's key splits
__global__ void fa128_combine_kernel(const float* __restrict__ po, const float* __restrict__ plse, __bf16* __restrict__ out, int ldo,
                                     int64_t o_bs, int64_t rows, int Lq, int heads, int nsplit, int hd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row, 4 consecutive columns)
    const int D4 = heads * hd / 4;
    if (i >= rows * D4) return;
    const int64_t row = i / D4;
    const int c = (int)(i - row * D4) * 4, head = c / hd;
    float mx = -INFINITY;
    for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, plse[((size_t)s * rows + row) * heads + head]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float ws = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float w = __builtin_amdgcn_exp2f(plse[((size_t)s * rows + row) * heads + head] - mx);
        acc += w * *reinterpret_cast<const f32x4*>(po + ((size_t)s * rows + row) * (heads * hd) + c);
        ws += w;
    }
    const float inv = 1.0f / ws;
    const int64_t b = row / Lq, l = row - b * Lq;
    *reinterpret_cast<bf16x4*>(out + (size_t)b * o_bs + (size_t)l * ldo + c) =
        bf16x4{(__bf16)(acc[0] * inv), (__bf16)(acc[1] * inv), (__bf16)(acc[2] * inv), (__bf16)(acc[3] * inv)};
}

// LayerNorm(eps, no affine) -> (1 + scale) y + shift with mod[tok / rows_per_mod] = {shift, scale} [2][D] -> proj_out -> un-patchify.
// One wave per token; D = 128 NP.  out [B][C][F][H][W] fp32, feature o = (py 2 + px) C + c.
template <int NP>
__global__ __launch_bounds__(256) void wan_final_kernel(const __bf16* __restrict__ x, const float* __restrict__ mod, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out, int ntok, int Fr, int gh, int gw,
                                                        int C, float eps) {
    constexpr int D = 128 * NP;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= ntok) return;
    const int fs = gh * gw;
    const int bf = tok / fs, hw = tok - bf * fs, b = bf / Fr, f = bf - b * Fr, gy = hw / gw, gx = hw - gy * gw;
    float v[2 * NP];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const bf16x2 q = *reinterpret_cast<const bf16x2*>(x + (size_t)tok * D + 2 * (lane + 64 * j));
        v[2 * j] = (float)q[0], v[2 * j + 1] = (float)q[1];
        s += v[2 * j] + v[2 * j + 1];
    }
    const float mean = wsum(s) * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 2 * NP; ++e) {
        v[e] -= mean;
        ss = fmaf(v[e], v[e], ss);
    }
    const float rstd = 1.0f / sqrtf(wsum(ss) * (1.0f / D) + eps);
    const float* mrow = mod + (size_t)bf * 2 * D;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = 2 * (lane + 64 * j);
        const f32x2 sh = *reinterpret_cast<const f32x2*>(mrow + c), sc = *reinterpret_cast<const f32x2*>(mrow + D + c);
        v[2 * j] = fmaf(v[2 * j] * rstd, 1.0f + sc[0], sh[0]);
        v[2 * j + 1] = fmaf(v[2 * j + 1] * rstd, 1.0f + sc[1], sh[1]);
    }
    const int H = 2 * gh, W = 2 * gw;
    for (int o = 0; o < 4 * C; ++o) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const f32x2 q = *reinterpret_cast<const f32x2*>(w + (size_t)o * D + 2 * (lane + 64 * j));
            a = fmaf(v[2 * j], q[0], a);
            a = fmaf(v[2 * j + 1], q[1], a);
        }
        a = wsum(a);
        if (lane == 0) {
            const int c = o % C, pq = o / C, py = pq >> 1, px = pq & 1;
            out[((((size_t)b * C + c) * Fr + f) * H + 2 * gy + py) * W + 2 * gx + px] = a + bias[o];
        }
    }
}

#define WAN_RET() return (int)hipGetLastError()

}  // namespace

int launch_wan_patch_embed(const float* x, const float* w, const float* bias, void* out, int B, int C, int Fr, int H, int W, int D, hipStream_t s) {
    const int64_t total = (int64_t)B * Fr * (H / 2) * (W / 2) * D;
    if (C == 16 && (D % 2) == 0) {
        const int64_t ntok = (int64_t)B * Fr * (H / 2) * (W / 2);
        hipLaunchKernelGGL(wan_patch_embed16_kernel, dim3((unsigned)((ntok + 31) / 32)), dim3(256), 0, s, x, w, bias, (__bf16*)out, B, Fr, H, W, D);
        WAN_RET();
    }
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(wan_patch_embed_kernel, dim3((unsigned)(blocks > 262144 ? 262144 : blocks)), dim3(256), 0, s, x, w, bias, (__bf16*)out, B, C,
                       Fr, H, W, D);
    WAN_RET();
}
int launch_wan_mod(const float* table, const float* tproj, float* mod, int rows, int J, int D, hipStream_t s) {
    const int64_t n = (int64_t)rows * J * D;
    hipLaunchKernelGGL(wan_mod_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, table, tproj, mod, rows, J, D);
    WAN_RET();
}
int launch_wan_outmod(const float* table, const float* temb, float* mod, int rows, int D, hipStream_t s) {
    const int64_t n = (int64_t)rows * 2 * D;
    hipLaunchKernelGGL(wan_outmod_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, table, temb, mod, rows, D);
    WAN_RET();
}
int launch_silu(const float* x, float* y, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
    WAN_RET();
}
int launch_cvt_rows_bf16(const float* x, void* y, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(cvt_rows_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, (__bf16*)y, n);
    WAN_RET();
}
int launch_wan_rope_table(const float2* tab_t, const float2* tab_h, const float2* tab_w, float2* cs, int L, int fs, int gw, int start, int S,
                          int nt, int nh, int nw, hipStream_t s) {
    const int n = L * (nt + nh + nw);
    hipLaunchKernelGGL(wan_rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, s, tab_t, tab_h, tab_w, cs, L, fs, gw, start, S, nt, nh, nw);
    WAN_RET();
}
// D = 128 NP with NP in {2, 3, 12, 16, 40} (256, 384, 1536 = the 1.3B network, 2048, 5120 = the 14B network)
int launch_rms_rope(int D, const void* src, int ld_src, const float* w, float eps, const float2* cs, void* dst, int64_t dst_bs, int dst_row0,
                    int ld_dst, int rows, int L, hipStream_t s) {
    dim3 g((rows + 3) / 4), b(256);
#define RR(NP) hipLaunchKernelGGL(rms_rope_kernel<NP>, g, b, 0, s, (const __bf16*)src, ld_src, w, eps, cs, (__bf16*)dst, dst_bs, dst_row0, ld_dst, rows, L)
    switch (D) {
        case 256: RR(2); break;
        case 384: RR(3); break;
        case 1536: RR(12); break;
        case 2048: RR(16); break;
        case 5120: RR(40); break;
        default: return (int)hipErrorInvalidValue;
    }
#undef RR
    WAN_RET();
}
// Key splits: how many workgroups share one query tile's keys, so that short grids (one sample: ceil(Lq / 128) x heads workgroups)
// still fill 256 CUs x 2 resident workgroups.  scratch (nullable -> no split) holds the splits'
// partial outputs and log-sum-exps: fa128_scratch_bytes(B, heads, Lq).
constexpr int FA_MAX_SPLIT = 8;
size_t fa128_scratch_bytes(int B, int heads, int Lq) { return (size_t)FA_MAX_SPLIT * B * Lq * ((size_t)heads * 128 + heads) * 4 + 256; }
int launch_fa(int hd, const void* q, int ldq, int64_t q_bs, const void* k, const void* v, int ldk, int64_t kv_bs, void* out, int ldo, int64_t o_bs,
              int B, int heads, int Lq, int Lkv, hipStream_t s, void* scratch, int force_split) {
    if ((hd != 128 && hd != 72) || Lq <= 0 || Lkv <= 0 || (ldq % 8) || (ldk % 8) || (ldo % 4)) return (int)hipErrorInvalidValue;
    if (force_split < 0 || force_split > FA_MAX_SPLIT || (force_split > 1 && !scratch)) return (int)hipErrorInvalidValue;
    const int qtiles = (Lq + 127) / 128, ktiles = (Lkv + 31) / 32;
    // key splits: units (sample, head, split) are dealt to the 8 XCDs (fa_kernel's mapping: 32 CUs each, two workgroups resident per CU at
    // half speed each).  Cost in tile iterations (~1.2 us) = workgroups on the fullest CU / 2 x (key tiles per split + ~6 of prologue
    // and epilogue) + the merge pass (fp32 partial outputs read back at ~5 TB/s); every split keeps >= 8 key tiles.
    int nsplit = 1;
    if (force_split > 0) {  // (parity tests: the splits of one launch against another's)
        nsplit = force_split < ktiles ? force_split : ktiles;
    } else if (scratch) {
        double best = -1.0;
        for (int c = 1; c <= FA_MAX_SPLIT && (c == 1 || ktiles / c >= 8); ++c) {
            const long long ux = ((long long)B * heads * c + 7) / 8, per_cu = (ux * qtiles + 31) / 32;
            const double cost = 0.5 * (double)per_cu * ((ktiles + c - 1) / c + 6) + (c > 1 ? 3.0 + 6.7e-7 * c * B * (double)Lq * heads * hd : 0.0);
            if (best < 0.0 || cost < best) best = cost, nsplit = c;
        }
    }
    float* po = (float*)scratch;
    float* plse = po ? po + (size_t)FA_MAX_SPLIT * B * Lq * heads * hd : nullptr;
    const int sample_major = (nsplit == 1 && B >= 8 && Lq <= 1024) ? 1 : 0;
    const long long units = sample_major ? (long long)((B + 7) / 8) * 8 * heads : (long long)B * heads * nsplit;
    dim3 g((unsigned)(((units + 7) / 8) * 8 * qtiles));
    static int use_dma = -1;  // head dim 128: LDS-DMA ring (1, default) | two-deep register prefetch (0): FASTGEN_AMD_FA_DMA
    if (use_dma < 0) {
        const char* e = getenv("FASTGEN_AMD_FA_DMA");
        use_dma = (e && e[0] == '0') ? 0 : 1;
    }
#ifdef FG_TIMING_BUILD
    static int abl = -1;  // FASTGEN_AMD_FA_ABL bit mask, see fa_kernel's pipelined step (libfastgen_amd_timing.so only)
    if (abl < 0) {
        const char* e = getenv("FASTGEN_AMD_FA_ABL");
        abl = e ? atoi(e) : 0;
    }
#endif
    static int minw = -1;  // waves per SIMD the kernel is compiled for (register budget 256 | 168): FASTGEN_AMD_FA_WAVES = 2 (default: room for the tile's fragments) | 3
    if (minw < 0) {
        const char* e = getenv("FASTGEN_AMD_FA_WAVES");
        minw = (e && e[0] == '3') ? 3 : 2;
    }
    static int seq72 = -1;  // head dim 72, <= 256 keys: the whole-sequence kernel (1, default) | fa_kernel<72> (0): FASTGEN_AMD_FA_SEQ72
    if (seq72 < 0) {
        const char* e = getenv("FASTGEN_AMD_FA_SEQ72");
        seq72 = (e && e[0] == '0') ? 0 : 1;
    }
    const float sc = 1.44269504088896341f / sqrtf((float)hd);
#define FA_GO(HD, MW, LDS, DM)                                                                                                              \
    hipLaunchKernelGGL((fa_kernel<HD, MW, DM>), g, dim3(256), LDS, s, (const __bf16*)q, ldq, q_bs, (const __bf16*)k, (const __bf16*)v, ldk, kv_bs, \
                       (__bf16*)out, ldo, o_bs, Lq, Lkv, sc, nsplit, po, plse, heads, B, sample_major FA_ABL_ARG)
    // head dim 128: fa2_kernel (64 queries per wave, one wave per SIMD) where it measures ahead - self-attention over >= 1024 keys:
    // 165 / 596 / 1005 us against fa_kernel's 186 / 608 / 1033 at 4680 queries x 4680 / 18720 / 32760 keys, the 21-frame loop 1.168 s
    // against 1.180 (profiles/r03_fa2_experiments.txt); the text cross-attention (512 keys) stays with fa_kernel (42 us against 50).
    // FASTGEN_AMD_FA_WIDE = 0 | 1 forces one of them for every length.
    static int wide = -2;
    if (wide == -2) {
        const char* e = getenv("FASTGEN_AMD_FA_WIDE");
        wide = !e ? -1 : (e[0] == '1' ? 1 : 0);
    }
    const bool use_wide = wide == 1 || (wide == -1 && Lkv >= 1024);
    if (hd == 128) {
        if ((size_t)Lkv * ldk * 2 >= ((size_t)1 << 31)) return (int)hipErrorInvalidValue;  // (buffer-resource offsets are 32-bit)
        if (use_wide && use_dma && !sample_major) {
            const int qt2 = (Lq + 255) / 256;
            // fewer workgroups than CUs (one per CU: 256): the uneven two-way split (fa2_kernel) - the long pieces last f of the keys with
            // f = P (L + c) / (L CUs), c ~ 6 tiles of prologue / epilogue per piece: then the CUs - P spare CUs get through the P short
            // pieces while the long ones run
            const long long P = (long long)B * heads * qt2;
            int t_cut = 0;
            unsigned grid = (unsigned)(((units + 7) / 8) * 8 * qt2);
            static int cut = -1;  // FASTGEN_AMD_FA_CUT = 0: even splits by the cost model above instead (A / B)
            if (cut < 0) {
                const char* e = getenv("FASTGEN_AMD_FA_CUT");
                cut = (e && e[0] == '0') ? 0 : 1;
            }
            if (cut && force_split == 0 && scratch && P <= 248 && ktiles >= 32) {
                const double f = (double)P * (ktiles + 6) / ((double)ktiles * 256.0);
                if (f < 0.97) {
                    t_cut = (int)(f * ktiles) + 1;
                    if (t_cut < (ktiles + 1) / 2) t_cut = (ktiles + 1) / 2;
                    nsplit = 2;
                    grid = (unsigned)(2 * 8 * ((P + 7) / 8));
                }
            }
            hipLaunchKernelGGL(fa2_kernel, dim3(grid), dim3(256), 6 * 16384, s, (const __bf16*)q, ldq, q_bs, (const __bf16*)k, (const __bf16*)v, ldk,
                               kv_bs, (__bf16*)out, ldo, o_bs, Lq, Lkv, sc, nsplit, po, plse, heads, B, t_cut FA_ABL_ARG);
        } else if (!use_dma) FA_GO(128, 2, 32768, false);
        else if (minw == 3) FA_GO(128, 3, 65536, true);
        else FA_GO(128, 2, 65536, true);
    } else if (Lkv <= 256 && !force_split && seq72) {
        const int qt = (Lq + 127) / 128;
        hipLaunchKernelGGL(fa72_seq_kernel, dim3((unsigned)(((B + 7) / 8) * 8 * heads * qt)), dim3(256), 2 * 256 * 144 + 64, s, (const __bf16*)q, ldq,
                           q_bs, (const __bf16*)k, (const __bf16*)v, ldk, kv_bs, (__bf16*)out, ldo, o_bs, Lq, Lkv, sc, heads, B);
    } else {
        if (minw == 3) FA_GO(72, 3, 2 * (32 * 144 + 32 * 192), false);
        else FA_GO(72, 2, 2 * (32 * 144 + 32 * 192), false);
    }
#undef FA_GO
    if (nsplit > 1) {
        const int64_t rows = (int64_t)B * Lq, n = rows * heads * hd / 4;
        hipLaunchKernelGGL(fa128_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, po, plse, (__bf16*)out, ldo, o_bs, rows, Lq,
                           heads, nsplit, hd);
    }
    WAN_RET();
}
int launch_fa128(const void* q, int ldq, int64_t q_bs, const void* k, const void* v, int ldk, int64_t kv_bs, void* out, int ldo, int64_t o_bs,
                 int B, int heads, int Lq, int Lkv, hipStream_t s, void* scratch) {
    return launch_fa(128, q, ldq, q_bs, k, v, ldk, kv_bs, out, ldo, o_bs, B, heads, Lq, Lkv, s, scratch, 0);
}
int launch_wan_final(int D, const void* x, const float* mod, const float* w, const float* bias, float* out, int ntok, int Fr, int gh, int gw, int C,
                     float eps, hipStream_t s) {
    dim3 g((ntok + 3) / 4), b(256);
#define WF(NP) hipLaunchKernelGGL(wan_final_kernel<NP>, g, b, 0, s, (const __bf16*)x, mod, w, bias, out, ntok, Fr, gh, gw, C, eps)
    switch (D) {
        case 256: WF(2); break;
        case 384: WF(3); break;
        case 1536: WF(12); break;
        case 2048: WF(16); break;
        case 5120: WF(40); break;
        default: return (int)hipErrorInvalidValue;
    }
#undef WF
    WAN_RET();
}
