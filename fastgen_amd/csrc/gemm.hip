// Token GEMM of the transformer blocks (DiT: fastgen/networks/DiT/network.py:153-201 - qkv, attention.proj, feed_forward.fc1 / fc2;
// the causal video DiT's block linears take the same kernel), bf16 compute mode: the arithmetic the reference runs these networks
// in (`precision_amp = "bfloat16"`, configs/experiments/DiT/config_*.py:19-26; Wan: `precision = "bfloat16"`, WanT2V/config_sf.py:19).
//
//   out[tok][n] = epilogue( sum_k A[tok][k] W[n][k] + bias[n] )        A: [M][K] bf16 row-major, W: [N][K] bf16 row-major
//
// Why not conv_fused_kernel's 1x1 token mode (conv.hip OUT_TOK): that kernel is a convolution kernel - 128 x 128 per 256-thread
// workgroup, weights streamed L2 -> registers by every wave, one workgroup barrier per 64-deep step of 16 MFMAs per wave; on the
// DiT-XL shapes it reaches 24 % of the bf16 roof (profiles/r02_dit_*).  A GEMM has no halo and no prologue arithmetic, so both
// operands can be staged once per workgroup and the tile can be twice as large in both directions:
//
//   persistent, one 512-thread workgroup per CU (two waves per SIMD), tile 256 tokens x (64 NT) outputs, NT = 4 | 3
//   (N % 256 == 0 -> 256 wide; DiT-XL's 1152 / 3456 = 6 / 18 x 192 -> 192 wide), K in steps of 64 through a two-stage LDS ring.
//   Per step and CU: 32 KiB of A + 32 KiB of W from L2 for 2 x 256 x 256 x 64 FLOP = 74 GB/s per CU at the MFMA roof - a
//   128 x 256 tile with register-streamed weights would need 112 GB/s against the vector L1's 64 B/clk.
//   Every wave: 128 tokens x 16 NT outputs = 8 x NT accumulator tiles of v_mfma_f32_16x16x32_bf16, TRANSPOSED (A operand =
//   weights [16 n x 32 k], B operand = activations [32 k x 16 tokens]) so that a lane ends with 4 CONSECUTIVE outputs of one
//   token: the epilogue (bias, tanh-GELU, adaLN gate x value + residual, head split) runs on registers and stores 8-byte
//   pieces, four lanes = one 32-byte run, the NT tiles of a wave back to back = 128 contiguous bytes per token.
//   LDS image of a stage: 8 planes per operand (plane o = k-octet o of the step: [row][8 bf16 = 16 B]), planes 16 B more than
//   a multiple of 256 B apart: a fragment read (16 consecutive rows of one plane) is one contiguous 256-byte run, and the
//   staging store of a quarter wave (8 octets of one row, the coalesced 128-byte global read) lands on 8 different bank groups.
//   LDS reads: (8 + NT) KiB per wave and 32-deep half step = 96 B/clk/CU at the MFMA roof, of 256.
//
// Tile order: workgroup b sits on XCD b % 8.  The 8 XCDs are arranged xm x xn over (token tiles, output tiles); inside its
// rectangle an XCD walks output tiles fastest, so its 32 workgroups share the A rows they read at about the same time (one
// HBM / MALL fetch per XCD) and its slice of W (N K 2 / xn bytes) stays in its 4 MiB L2.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "misc.h"

#ifndef W4_EXP
#define W4_EXP 0  // gemm_bf16_w4_kernel timing experiments (wrong results): 1 no LDS-DMA in the K-loop, 2 no fragment reads
#endif

namespace {

constexpr int GM_NTHR = 512;
constexpr int GM_TM = 256;   // tokens per tile
constexpr int GM_KC = 64;    // k per pipeline step
constexpr int GM_PA = GM_TM * 16 + 16;  // bytes between the octet planes of the A image

__device__ __forceinline__ void gm_barrier() {
    // this wave's LDS traffic retired, then the workgroup barrier; vmcnt is NOT drained (epilogue stores stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float gm_gelu_tanh(float x) {
    // torch.nn.GELU(approximate="tanh") (DiT/network.py:176): 0.5 x (1 + tanh(u)) = x sigmoid(2 u), u = sqrt(2/pi) (x + 0.044715 x^3),
    // as x / (1 + 2^(-z)) with z = 2 u log2(e) = x (c0 + c1 x^2): seven VALU operations, two of them transcendental (v_exp / v_rcp, bf16
    // mode) - the epilogue of the fc1 GEMM evaluates 128 of these per lane and tile
    constexpr float c0 = 2.0f * 0.7978845608028654f * 1.44269504088896341f, c1 = c0 * 0.044715f;
    const float z = x * fmaf(x * x, c1, c0);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-z));
}

enum { GM_EPI_TOK = 0, GM_EPI_HEADS = 1, GM_EPI_RAW = 2, GM_EPI_TOK32 = 3, GM_EPI_SPLIT = 4, GM_EPI_HEADS32 = 5 };

template <int NT, int EPI>
__global__ __launch_bounds__(GM_NTHR) void gemm_bf16_kernel(const GemmArgs a) {
    constexpr int TN = 64 * NT;             // outputs per tile
    constexpr int PW = TN * 16 + 16;        // bytes between the octet planes of the W image
    constexpr int STAGE = 8 * (GM_PA + PW);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // token half, output quarter of the tile
    const int col = lane & 15, g = lane >> 4;

    // ---- this workgroup's tiles -------------------------------------------------------------------------------------
    const int Mt = (a.M + GM_TM - 1) / GM_TM, Nt = (a.N + TN - 1) / TN;
    int mlo = 0, nlo = 0, nx = Nt, first, stride, count;
    if ((gridDim.x & 7) == 0 && a.xn > 0) {
        const int x = (int)blockIdx.x & 7, xm = 8 / a.xn, xi = x / a.xn, xj = x - xi * a.xn;
        mlo = (int)((long long)xi * Mt / xm);
        const int mhi = (int)((long long)(xi + 1) * Mt / xm);
        nlo = xj * Nt / a.xn;
        nx = (xj + 1) * Nt / a.xn - nlo;
        first = (int)blockIdx.x >> 3;
        stride = (int)gridDim.x >> 3;
        count = (mhi - mlo) * nx;
    } else {
        first = (int)blockIdx.x;
        stride = (int)gridDim.x;
        count = Mt * Nt;
    }
    const int my_tiles = first < count ? (count - first + stride - 1) / stride : 0;
    if (my_tiles == 0) return;
    auto tile_origin = [&](int i, int& m0, int& n0) {
        const int lt = first + i * stride;
        const int q = lt / nx;
        m0 = (mlo + q) * GM_TM;
        n0 = (nlo + (lt - q * nx)) * TN;
    };
    const int nk = a.K / GM_KC;
    const int S = my_tiles * nk;

    // ---- staging: thread -> (row tid >> 3 + 64 i, octet tid & 7) of both operand tiles ------------------------------
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A), rsW = make_rsrc(a.W);
    const int srow = tid >> 3, soct = tid & 7;
    const int ldsA = soct * GM_PA + srow * 16;
    const int ldsW = 8 * GM_PA + soct * PW + srow * 16;
    unsigned offA[4], offW[NT];
    fg_u32x4 pa[4], pw[NT];
    auto setup = [&](int i) {
        int m0, n0;
        tile_origin(i, m0, n0);
#pragma unroll
        for (int j = 0; j < 4; ++j) offA[j] = (unsigned)min(m0 + srow + 64 * j, a.M - 1) * (unsigned)(a.K * 2) + soct * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) offW[j] = (unsigned)min(n0 + srow + 64 * j, a.N - 1) * (unsigned)(a.K * 2) + soct * 16;
    };
    auto issue = [&](int kidx) {
        const int so = __builtin_amdgcn_readfirstlane(kidx * (GM_KC * 2));  // (uniform; kept off the vector unit: no waterfall loop per load)
#pragma unroll
        for (int j = 0; j < 4; ++j) pa[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA[j], so, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) pw[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, offW[j], so, 0);
    };
    auto park = [&](char* st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<fg_u32x4*>(st + ldsA + j * 1024) = pa[j];
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<fg_u32x4*>(st + ldsW + j * 1024) = pw[j];
    };

    // ---- fragments ----------------------------------------------------------------------------------------------------
    const int fA = g * GM_PA + (wm * 128 + col) * 16;                 // + 4 j PA + m 256
    const int fW = 8 * GM_PA + g * PW + (wn * 16 * NT + col) * 16;    // + 4 j PW + nt 256
    f32x4 acc[8][NT];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const char* st) {
        bf16x8 x0[4], x1[4], w0[NT], w1[NT];
        auto read_x = [&](int j, int half, bf16x8 (&x)[4]) {
#pragma unroll
            for (int m = 0; m < 4; ++m) x[m] = *reinterpret_cast<const bf16x8*>(st + fA + 4 * j * GM_PA + (half * 4 + m) * 256);
        };
        auto read_w = [&](int j, bf16x8 (&w)[NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w[nt] = *reinterpret_cast<const bf16x8*>(st + fW + 4 * j * PW + nt * 256);
        };
        auto mma = [&](int half, const bf16x8 (&x)[4], const bf16x8 (&w)[NT]) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[half * 4 + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nt], x[m], acc[half * 4 + m][nt], 0, 0, 0);
        };
        read_w(0, w0);
        read_x(0, 0, x0);
        __builtin_amdgcn_sched_barrier(0);
        read_x(0, 1, x1);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, x0, w0);
        __builtin_amdgcn_sched_barrier(0);
        read_w(1, w1);
        read_x(1, 0, x0);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, x1, w0);
        __builtin_amdgcn_sched_barrier(0);
        read_x(1, 1, x1);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, x0, w1);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, x1, w1);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- epilogue of one finished tile: registers -> global -----------------------------------------------------------
    auto epilogue = [&](int i) {
        int m0, n0;
        tile_origin(i, m0, n0);
        const int c0 = n0 + wn * 16 * NT + 4 * g;  // + 16 nt: this lane's 4 consecutive outputs
        f32x4 b4[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            b4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias && c0 + 16 * nt < a.N) b4[nt] = *reinterpret_cast<const f32x4*>(a.bias + c0 + 16 * nt);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int row = m0 + wm * 128 + m * 16 + col;
            if (row < a.M) {
                if constexpr (EPI == GM_EPI_TOK) {
                    __bf16* orow = reinterpret_cast<__bf16*>(a.out) + (size_t)row * a.N;
                    const __bf16* rrow = reinterpret_cast<const __bf16*>(a.resid) + (size_t)row * a.N;
                    const float* grow = a.gate ? a.gate + (size_t)((a.row0 + row) / a.gate_rows) * a.gate_stride : nullptr;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int c = c0 + 16 * nt;
                        if (c < a.N) {
                            f32x4 v = acc[m][nt] + b4[nt];
                            if (a.act == 1) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = gm_gelu_tanh(v[e]);
                            }
                            if (grow) v *= *reinterpret_cast<const f32x4*>(grow + c);
                            if (a.resid) v += load4(rrow + c);
                            store4(orow + c, v);
                        }
                    }
                } else {
                    // head-split q | k [B][H][T][hd], v^T [B][H][hd][T] (the operand layouts of dit_attention_kernel)
                    const int D = a.heads * a.head_dim;
                    const int b = (a.row0 + row) / a.T, t = a.row0 + row - b * a.T;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int c = c0 + 16 * nt;
                        if (c < a.N) {
                            const f32x4 v = acc[m][nt] + b4[nt];
                            const int plane = c / D, within = c - plane * D;
                            const int hh = within / a.head_dim, d = within - hh * a.head_dim;
                            const size_t bh = (size_t)b * a.heads + hh;
                            if (plane < 2) {
                                store4(reinterpret_cast<__bf16*>(plane ? a.k : a.q) + (bh * a.T + t) * a.head_dim + d, v);
                            } else {
                                __bf16* vp = reinterpret_cast<__bf16*>(a.vt) + (bh * a.head_dim + d) * a.T + t;
#pragma unroll
                                for (int e = 0; e < 4; ++e) vp[(size_t)e * a.T] = (__bf16)v[e];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- step machine: during step s every thread (1) issues the global loads of step s + 1, (2) computes step s out of stage
    // s & 1, (3) after a tile's last step stores the tile, (4) parks step s + 1 in stage (s + 1) & 1; one barrier per step --------
    int ti = 0, kk = 0;   // tile / k-step being computed
    int tn = 0, kn = 0;   // ... being staged
    setup(0);
    issue(0);
    park(smem);
    if (++kn == nk) kn = 0, ++tn;
    gm_barrier();
    for (int s = 0; s < S; ++s) {
        // (the step behind the last one is staged too, from clamped - valid - addresses into the stage nobody reads any more:
        // unconditional loads let the compiler count them, a conditional issue makes it drain vmcnt before every step's LDS reads)
        if (kn == 0) setup(tn);
        issue(kn);
        compute(smem + (s & 1) * STAGE);
        if (kk + 1 == nk) epilogue(ti);
        park(smem + ((s + 1) & 1) * STAGE);
        if (++kn == nk) kn = 0, ++tn;
        if (++kk == nk) kk = 0, ++ti;
        gm_barrier();
    }
}

// ---- the 8-wave ping-pong schedule ----------------------------------------------------------------------------------------
// Same tile (256 tokens x 256 outputs, 8 waves = 2 token halves x 4 output quarters, 128 x 64 per wave) and the same register
// epilogue, but (1) the operand tiles go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round trip, no
// ds_write - the kernel above spends 830 of every 2048 cycles of LDS time on its 64 KiB of ds_write_b128 per step), and (2) the
// two waves of a SIMD alternate: the K-step of 64 is cut into four phases of 16 MFMAs (one 64 x 32 quadrant of the wave's
// outputs), every phase = {memory part: ds_read the fragments a later phase needs + issue one half-tile of DMA + counted
// vmcnt; barrier; compute part: 16 MFMAs; barrier}, and waves 4-7 run one barrier behind waves 0-3 - at any time one wave of
// each SIMD is in its MFMA block while its partner is in its memory part (MI355X_MICROARCH.md, two waves per SIMD: the matrix
// pipe is per-SIMD and paced; a partner's LDS / DMA segment costs the computing wave 40-50 cycles).
// LDS: 2 buffers x {A rows 0-127, A rows 128-255, W rows 0-127, W rows 128-255} x 16 KiB, each half-tile row-major [128][64 k]
// with the 16-byte k-octets of row r XOR-ed by (r >> 1) & 7 - applied on the DMA's per-lane SOURCE address (the LDS destination
// of one wave-instruction is 1 KiB contiguous) and on the fragment read: the 16 lanes of a ds_read_b128 lane group then cover
// all 64 banks.
// Half-tile schedule (t = flat K-step index, d = 2 phases between a DMA's issue and the counted wait that retires it, so a
// half-tile issued in phase p is readable from p + 3; a region is re-issued at the earliest one phase after its last read,
// whose lgkmcnt(0) sits before that phase's barrier):
//   phase 0: read W(t) quarters nt 0,1    issue A rows 128-255 of t+1     phase 2: read X(t) m 4-7     issue W rows 0-127 of t+2
//   phase 1: read W(t) quarters nt 2,3    issue W rows 128-255 of t+1     phase 3: read X(t+1) m 0-3   issue A rows 0-127 of t+2
// Fragment reads are inline asm: hipcc orders every compiler-visible LDS read behind ALL pending LDS-DMA (vmcnt(0)), which
// would serialise the pipeline; the waits here are counted by hand (vmcnt(4) = two half-tiles in flight).
// Ragged edges: the last token tile / output tile is shifted back to end at M / N (M, N >= 256); what it recomputes of its neighbour
// tile is not stored again.
// act bits 4 / 8 / 16 (no stores / no epilogue / cycle stamps) are timing experiments with wrong results: compiled only into
// libfastgen_amd_timing.so (`make timing`, -DFG_TIMING_BUILD); in the product build the tests fold to false and act is 0 or 1.
#ifdef FG_TIMING_BUILD
#define GM_TIMING(x) (x)
#else
#define GM_TIMING(x) false
#endif
#define GM_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int EPI>
__global__ __launch_bounds__(GM_NTHR) void gemm_bf16_pp_kernel(const GemmArgs a) {
    constexpr int TN = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;  // token half (= ping-pong group), output quarter
    const int col = lane & 15, g = lane >> 4;

    // ---- this workgroup's tiles (as gemm_bf16_kernel) ----------------------------------------------------------------
    const int Mt = (a.M + GM_TM - 1) / GM_TM, Nt = (a.N + TN - 1) / TN;
    int mlo = 0, nlo = 0, nx = Nt, first, stride, count;
    if ((gridDim.x & 7) == 0 && a.xn > 0) {
        const int x = (int)blockIdx.x & 7, xm = 8 / a.xn, xi = x / a.xn, xj = x - xi * a.xn;
        mlo = (int)((long long)xi * Mt / xm);
        const int mhi = (int)((long long)(xi + 1) * Mt / xm);
        nlo = xj * Nt / a.xn;
        nx = (xj + 1) * Nt / a.xn - nlo;
        first = (int)blockIdx.x >> 3;
        stride = (int)gridDim.x >> 3;
        count = (mhi - mlo) * nx;
    } else {
        first = (int)blockIdx.x;
        stride = (int)gridDim.x;
        count = Mt * Nt;
    }
    // split-K (EPI == GM_EPI_RAW): a work item is (tile, split); the ks splits of a tile are neighbours in the enumeration (same
    // XCD: they read the same A rows and W rows at different k) and leave fp32 partial sums for gemm_splitk_finish_kernel
    const int ks = (EPI == GM_EPI_RAW) ? a.ksplit : 1;
    count *= ks;
    const int my_tiles = first < count ? (count - first + stride - 1) / stride : 0;  // work items of this workgroup
    if (my_tiles == 0) return;
    auto tile_origin = [&](int i, int& m0, int& n0) {
        const int lt = (first + min(i, my_tiles - 1) * stride) / ks;  // (cursors running past the end re-read the last tile)
        const int q = lt / nx;
        m0 = min((mlo + q) * GM_TM, a.M - GM_TM);
        n0 = min((nlo + (lt - q * nx)) * TN, a.N - TN);
    };
    auto item_split = [&](int i) { return (first + min(i, my_tiles - 1) * stride) % ks; };
    // where the tile would start without the shift: rows / columns below are the neighbour tile's and are not stored again
    auto tile_keep_from = [&](int i, int& mk, int& nk_) {
        const int lt = (first + min(i, my_tiles - 1) * stride) / ks;
        const int q = lt / nx;
        mk = (mlo + q) * GM_TM;
        nk_ = (nlo + (lt - q * nx)) * TN;
    };
    const int nk = (a.K / GM_KC) / ks;  // K-steps per work item
    const int S = my_tiles * nk;
    // row pitches in bytes (split-bf16 flavour: A rows hold [hi | lo] = 2 K1 elements, W rows [hi | lo | hi] = 3 K1 = K elements)
    const int KA2 = (a.lda ? a.lda : a.K) * 2, KW2 = (a.ldw ? a.ldw : a.K) * 2;
    // ---- DMA: wave w moves the 8-row groups 2 w, 2 w + 1 of every half-tile; lane -> (row lane >> 3 of the group, LDS octet
    // position lane & 7, which holds k-octet position ^ ((row >> 1) & 7)) ------------------------------------------------
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A), rsW = make_rsrc(a.W);
    unsigned voffA[2], voffW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 8 + (lane >> 3);
        const unsigned sw = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
        voffA[j] = (unsigned)row * (unsigned)KA2 + sw;
        voffW[j] = (unsigned)row * (unsigned)KW2 + sw;
    }
    // regions of buffer b: A half h at b * 65536 + h * 16384, W half h at b * 65536 + 32768 + h * 16384
    // (the scalar offset through v_readfirstlane: the cursors are uniform by construction, but hipcc kept the A cursor in a VGPR and
    // wrapped each of its DMA instructions in a waterfall loop - readfirstlane, compare, exec mask, branch - inside the K-loop)
    auto dmaA = [&](int region, int soff) {
        const int so = __builtin_amdgcn_readfirstlane(soff);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(smem + region + (wave * 2 + j) * 1024), 16, voffA[j], so, 0, 0);
    };
    auto dmaW = [&](int region, int soff) {
        const int so = __builtin_amdgcn_readfirstlane(soff);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(smem + region + (wave * 2 + j) * 1024), 16, voffW[j], so, 0, 0);
    };
    // cursors over the flat K-steps t + 1 and t + 2: scalar byte offsets of their A / W tiles' first row at their k
    struct Cur {
        int ti, kk, sa, sw;
    };
    auto cur_set = [&](Cur& c) {
        int m0, n0;
        tile_origin(c.ti, m0, n0);
        const int ks_ = item_split(c.ti) * nk + c.kk;                       // K-step of the (virtual) contraction
        const int ka = (a.a_wrap && ks_ >= a.a_wrap) ? ks_ - a.a_wrap : ks_;  // split-bf16: A' = [hi | hi | lo] read out of [hi | lo]
        c.sa = __builtin_amdgcn_readfirstlane(m0 * KA2 + ka * (GM_KC * 2));  // (uniform by construction; keeps the cursor arithmetic on the scalar unit)
        c.sw = __builtin_amdgcn_readfirstlane(n0 * KW2 + ks_ * (GM_KC * 2));
    };
    auto cur_next = [&](Cur& c) {
        if (++c.kk == nk) c.kk = 0, ++c.ti;
        cur_set(c);
    };

    // ---- fragments ----------------------------------------------------------------------------------------------------
    const int lrow0 = col * 128 + ((g ^ ((col >> 1) & 7)) * 16), lrow1 = lrow0 ^ 64;   // k-substep 0 / 1
    const int fxa = wr * 16384;                                  // this wave's A half
    const int fwb = 32768 + (wc >> 1) * 16384 + (wc & 1) * 8192;  // this wave's 64 W rows
    bf16x8 X0[4][2], X1[4][2], W0[2][2], W1[2][2];  // [m | nt][k-substep]
    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto read_x = [&](bf16x8 (&x)[4][2], int buf, int mh) {
        const int a0 = buf * 65536 + fxa + lrow0, a1 = buf * 65536 + fxa + lrow1;
        if (mh == 0) {
            GM_DSR(x[0][0], a0, 0); GM_DSR(x[0][1], a1, 0); GM_DSR(x[1][0], a0, 2048); GM_DSR(x[1][1], a1, 2048);
            GM_DSR(x[2][0], a0, 4096); GM_DSR(x[2][1], a1, 4096); GM_DSR(x[3][0], a0, 6144); GM_DSR(x[3][1], a1, 6144);
        } else {
            GM_DSR(x[0][0], a0, 8192); GM_DSR(x[0][1], a1, 8192); GM_DSR(x[1][0], a0, 10240); GM_DSR(x[1][1], a1, 10240);
            GM_DSR(x[2][0], a0, 12288); GM_DSR(x[2][1], a1, 12288); GM_DSR(x[3][0], a0, 14336); GM_DSR(x[3][1], a1, 14336);
        }
    };
    auto read_w = [&](bf16x8 (&w)[2][2], int buf, int nh) {
        const int a0 = buf * 65536 + fwb + lrow0, a1 = buf * 65536 + fwb + lrow1;
        if (nh == 0) {
            GM_DSR(w[0][0], a0, 0); GM_DSR(w[0][1], a1, 0); GM_DSR(w[1][0], a0, 2048); GM_DSR(w[1][1], a1, 2048);
        } else {
            GM_DSR(w[0][0], a0, 4096); GM_DSR(w[0][1], a1, 4096); GM_DSR(w[1][0], a0, 6144); GM_DSR(w[1][1], a1, 6144);
        }
    };
    auto mma = [&](int mh, int nh, const bf16x8 (&x)[4][2], const bf16x8 (&w)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mh * 4 + m][nh * 2 + nt] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nt][j], x[m][j], acc[mh * 4 + m][nh * 2 + nt], 0, 0, 0);
    };
    // end of a memory part: two half-tiles of DMA (4 instructions) may stay in flight, this wave's LDS reads have landed; then
    // the hand-over barrier.  (vmcnt retires in issue order, stores included: behind an epilogue the first such wait also waits
    // for the tile's stores.)
    auto mem_done = [&]() {
        asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto cmp_done = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
    };

    // ---- epilogue of one 64-token x 32-output quadrant (MH, NH) of this wave, on registers.  v_permlane16_swap exchanges the
    // odd-numbered output quads of the even 16-lane rows with the even-numbered quads of the odd rows: a lane then holds 8
    // CONSECUTIVE outputs of its token (n0 + 64 wc + 32 NH + 16 (g & 1) + 8 (g >> 1)): one 16-byte store, four lanes = one
    // 64-byte run.  be / bo: this lane's bias quads for the even / odd quad of the pair (loaded a K-step earlier: a load waited
    // for here would wait for the DMA in flight as well).
    // Measured and rejected (profiles/r02_gemm_*): the quadrants leaving one by one in the memory parts of the next four phases
    // (stores spread over a K-step, vmcnt budgets widened by the stores in flight: -7 %), and start offsets between XCDs or
    // between the token-tile groups of an XCD (-3...-8 %).  What the epilogue costs is measured with act bits 4 / 8: at K = 1152
    // the K-loop alone runs at 1450-1530 TFLOP/s, the epilogue arithmetic takes it to 1210, the stores (HBM write rate, exposed
    // by the in-order vmcnt) to 900-980.
    // token epilogue (GM_EPI_TOK): every address is buffer-resource arithmetic - a lane-constant byte offset (its row inside the
    // wave's half, its 8-output column group), everything that depends on the tile, the 16-row block or the quadrant in the scalar
    // offset.  (As 64-bit pointer arithmetic per row and quadrant the epilogue was ~1000 vector instructions per tile and wave -
    // 5 us per tile at two waves per SIMD, a quarter of a K = 1152 tile's time - mostly address computation.)
    int ep_m0 = 0, ep_n0 = 0, ep_mk = 0, ep_nk = 0;
    const int lane_c8 = wc * 64 + 16 * (g & 1) + 8 * (g >> 1);
    const int vo_out = ((wr * 128 + col) * a.N + lane_c8) * 2;
    const __amdgpu_buffer_rsrc_t rsO = make_rsrc(a.out), rsR = make_rsrc(a.resid), rsG = make_rsrc(a.gate);
    // (gate periods >= the tile height: a tile then spans at most two gate rows; launch_gemm_bf16 sends shorter ones to gemm_bf16_kernel)
    auto slice = [&](auto MH_, auto NH_, int tile, f32x4 be, f32x4 bo) {
        constexpr int MH = decltype(MH_)::value, NH = decltype(NH_)::value;
        if constexpr (EPI == GM_EPI_TOK) {
            if (!GM_TIMING(a.act & 8)) {
                const int m0 = ep_m0, n0 = ep_n0, mk = ep_mk, nk_ = ep_nk;  // (tile_origin / tile_keep_from of `tile`, computed once per tile)
                const bool edge = (mk != m0) || (nk_ != n0);  // a shifted last tile: part of it is its neighbour's (uniform)
                const int c8 = n0 + lane_c8 + 32 * NH;
                // gate x value: the gate rows of the tile's first token and of the next gate period (if the tile reaches into it)
                f32x4 g0e = {1.f, 1.f, 1.f, 1.f}, g0o = g0e, g1e = g0e, g1o = g0e;
                int bnd = 0x7fffffff;
                if (a.gate) {
                    const int gi0 = (a.row0 + m0) / a.gate_rows;
                    bnd = (gi0 + 1) * a.gate_rows - a.row0;  // first token (in this launch's rows) of the next gate period
                    const int gso = __builtin_amdgcn_readfirstlane((gi0 * a.gate_stride + n0 + 32 * NH) * 4);
                    g0e = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, lane_c8 * 4, gso, 0));
                    g0o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, lane_c8 * 4, gso + 16, 0));
                    g1e = g0e, g1o = g0o;
                    if (bnd < m0 + GM_TM) {
                        g1e = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, lane_c8 * 4, gso + a.gate_stride * 4, 0));
                        g1o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, lane_c8 * 4, gso + a.gate_stride * 4 + 16, 0));
                    }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int rb = (MH * 4 + m) * 16;  // 16-row block inside the wave's token half
                    const int row = m0 + wr * 128 + rb + col;
                    f32x4 ve = acc[MH * 4 + m][NH * 2] + be, vo = acc[MH * 4 + m][NH * 2 + 1] + bo;
                    acc[MH * 4 + m][NH * 2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc[MH * 4 + m][NH * 2 + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (a.act & 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ve[e] = gm_gelu_tanh(ve[e]), vo[e] = gm_gelu_tanh(vo[e]);
                    }
                    {
                        float e0 = ve[0], e1 = ve[1], e2 = ve[2], e3 = ve[3], o0 = vo[0], o1 = vo[1], o2 = vo[2], o3 = vo[3];
                        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\t"
                                     "v_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                                     : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
                        ve = f32x4{e0, e1, e2, e3};
                        vo = f32x4{o0, o1, o2, o3};
                    }
                    if (edge && !(row >= mk && c8 >= nk_)) continue;
                    if (a.gate) {
                        const bool nx_ = row >= bnd;
                        ve *= nx_ ? g1e : g0e;
                        vo *= nx_ ? g1o : g0o;
                    }
                    const int so = __builtin_amdgcn_readfirstlane(((m0 + rb) * a.N + n0 + 32 * NH) * 2);  // (scalar; < 2 GiB: launch_gemm_bf16 chunks the rows)
                    if (a.resid) {
                        const bf16x8 r8 = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsR, vo_out, so, 0));
#pragma unroll
                        for (int e = 0; e < 4; ++e) ve[e] += (float)r8[e], vo[e] += (float)r8[4 + e];
                    }
                    bf16x8 o8;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o8[e] = (__bf16)ve[e], o8[4 + e] = (__bf16)vo[e];
                    if (GM_TIMING(a.act & 4)) asm volatile("" ::"v"(o8));  // (timing build: everything but the store)
                    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fg_u32x4, o8), rsO, vo_out, so, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                return;
            }
        }
        if constexpr (EPI == GM_EPI_RAW) {
            if (!GM_TIMING(a.act & 8)) {  // split-K partial sums: fp32 [split][M][N], the same buffer-resource addressing as the token epilogue
                const int m0 = ep_m0, n0 = ep_n0, mk = ep_mk, nk_ = ep_nk;
                const bool edge = (mk != m0) || (nk_ != n0);
                const int c8 = n0 + lane_c8 + 32 * NH;
                const int sp = item_split(tile);
                const __amdgpu_buffer_rsrc_t rsS = make_rsrc(a.scratch);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int rb = (MH * 4 + m) * 16;
                    const int row = m0 + wr * 128 + rb + col;
                    f32x4 ve = acc[MH * 4 + m][NH * 2], vo = acc[MH * 4 + m][NH * 2 + 1];
                    acc[MH * 4 + m][NH * 2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc[MH * 4 + m][NH * 2 + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    {
                        float e0 = ve[0], e1 = ve[1], e2 = ve[2], e3 = ve[3], o0 = vo[0], o1 = vo[1], o2 = vo[2], o3 = vo[3];
                        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\t"
                                     "v_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                                     : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
                        ve = f32x4{e0, e1, e2, e3};
                        vo = f32x4{o0, o1, o2, o3};
                    }
                    if (edge && !(row >= mk && c8 >= nk_)) continue;
                    // (the launcher gives split-K only to grids of < 128 tiles: 4 splits x 128 tiles x 256 KiB of fp32 stay far below 2 GiB)
                    const int so = __builtin_amdgcn_readfirstlane((((sp * a.M + m0 + rb) * a.N) + n0 + 32 * NH) * 4);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fg_u32x4, ve), rsS, 2 * vo_out, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fg_u32x4, vo), rsS, 2 * vo_out + 16, so, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                return;
            }
        }
        if (GM_TIMING(a.act & 8)) {  // (timing build: no epilogue at all)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                asm volatile("" ::"v"(acc[MH * 4 + m][NH * 2]), "v"(acc[MH * 4 + m][NH * 2 + 1]));
                acc[MH * 4 + m][NH * 2] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[MH * 4 + m][NH * 2 + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            return;
        }
        int m0, n0, mk, nk_;
        tile_origin(tile, m0, n0);
        tile_keep_from(tile, mk, nk_);
        const int c8 = n0 + wc * 64 + 32 * NH + 16 * (g & 1) + 8 * (g >> 1);      // after the swap: 8 consecutive outputs
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = m0 + wr * 128 + (MH * 4 + m) * 16 + col;
            // a ragged last tile was shifted back to end at M / N: what it recomputes of its neighbour is not stored again (two
            // workgroups, usually on two XCDs, writing the same lines cost 20 % at N = 3456 - and an in-place residual stays exact)
            const bool keep = row >= mk && c8 >= nk_;
            f32x4 ve = acc[MH * 4 + m][NH * 2] + be, vo = acc[MH * 4 + m][NH * 2 + 1] + bo;
            acc[MH * 4 + m][NH * 2] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[MH * 4 + m][NH * 2 + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((EPI == GM_EPI_TOK || EPI == GM_EPI_SPLIT || EPI == GM_EPI_TOK32) && (a.act & 1)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ve[e] = gm_gelu_tanh(ve[e]), vo[e] = gm_gelu_tanh(vo[e]);
            }
            {
                // (inline asm: this hipcc's __builtin_amdgcn_permlane16_swap loses the SECOND result - it copies the first over it -
                // once the call sits in a larger body; scripts/micro/permlane16_swap_map.hip checks the instruction itself.  The
                // s_nops cover the VALU -> swap -> VALU wait states the compiler would otherwise insert.)
                float e0 = ve[0], e1 = ve[1], e2 = ve[2], e3 = ve[3], o0 = vo[0], o1 = vo[1], o2 = vo[2], o3 = vo[3];
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\t"
                             "v_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                             : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
                ve = f32x4{e0, e1, e2, e3};
                vo = f32x4{o0, o1, o2, o3};
            }
            if (!keep) continue;
            if constexpr (EPI == GM_EPI_TOK32 || EPI == GM_EPI_SPLIT) {
                // split-bf16 flavour: fp32 gate / residual arithmetic; out fp32 [M][N] (TOK32) or [hi | lo] bf16 planes [M][2 N], the
                // next GEMM's A operand (SPLIT)
                if (a.gate) {
                    const float* grow = a.gate + (size_t)((a.row0 + row) / a.gate_rows) * a.gate_stride + c8;
                    ve *= *reinterpret_cast<const f32x4*>(grow);
                    vo *= *reinterpret_cast<const f32x4*>(grow + 4);
                }
                if (a.resid) {
                    const float* rr = reinterpret_cast<const float*>(a.resid) + (size_t)row * a.N + c8;
                    ve += *reinterpret_cast<const f32x4*>(rr);
                    vo += *reinterpret_cast<const f32x4*>(rr + 4);
                }
                if constexpr (EPI == GM_EPI_TOK32) {
                    float* orow = reinterpret_cast<float*>(a.out) + (size_t)row * a.N + c8;
                    *reinterpret_cast<f32x4*>(orow) = ve;
                    *reinterpret_cast<f32x4*>(orow + 4) = vo;
                } else {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (__bf16)ve[e], hi[4 + e] = (__bf16)vo[e];
                        lo[e] = (__bf16)(ve[e] - (float)hi[e]), lo[4 + e] = (__bf16)(vo[e] - (float)hi[4 + e]);
                    }
                    __bf16* orow = reinterpret_cast<__bf16*>(a.out) + (size_t)row * 2 * a.N + c8;
                    *reinterpret_cast<bf16x8*>(orow) = hi;
                    *reinterpret_cast<bf16x8*>(orow + a.N) = lo;
                }
            } else if constexpr (EPI == GM_EPI_HEADS32) {
                // head-split q | k [B][H][T][hd] and v^T [B][H][hd][T] as hi / lo bf16 planes lo_off elements apart: the operand layout
                // of dit_attention_kernel<bf16x3>
                const int D = a.heads * a.head_dim;
                const int b = (a.row0 + row) / a.T, t = a.row0 + row - b * a.T;
                const int plane = c8 / D, within = c8 - plane * D;
                const int hh = within / a.head_dim, d = within - hh * a.head_dim;
                const size_t bh = (size_t)b * a.heads + hh;
                bf16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[e] = (__bf16)ve[e], hi[4 + e] = (__bf16)vo[e];
                    lo[e] = (__bf16)(ve[e] - (float)hi[e]), lo[4 + e] = (__bf16)(vo[e] - (float)hi[4 + e]);
                }
                if (plane < 2) {
                    __bf16* dst = reinterpret_cast<__bf16*>(plane ? a.k : a.q) + (bh * a.T + t) * a.head_dim + d;
                    *reinterpret_cast<bf16x8*>(dst) = hi;
                    *reinterpret_cast<bf16x8*>(dst + a.lo_off) = lo;
                } else {
                    __bf16* vp = reinterpret_cast<__bf16*>(a.vt) + (bh * a.head_dim + d) * a.T + t;
#pragma unroll
                    for (int e = 0; e < 8; ++e) vp[(size_t)e * a.T] = hi[e], vp[(size_t)e * a.T + a.lo_off] = lo[e];
                }
            } else {
                const int D = a.heads * a.head_dim;  // % 8 == 0: the 8 outputs lie in one plane and one head
                const int b = (a.row0 + row) / a.T, t = a.row0 + row - b * a.T;
                const int plane = c8 / D, within = c8 - plane * D;
                const int hh = within / a.head_dim, d = within - hh * a.head_dim;
                const size_t bh = (size_t)b * a.heads + hh;
                if (plane < 2) {
                    bf16x8 o8;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o8[e] = (__bf16)ve[e], o8[4 + e] = (__bf16)vo[e];
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(plane ? a.k : a.q) + (bh * a.T + t) * a.head_dim + d) = o8;
                } else {
                    __bf16* vp = reinterpret_cast<__bf16*>(a.vt) + (bh * a.head_dim + d) * a.T + t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) vp[(size_t)e * a.T] = (__bf16)ve[e], vp[(size_t)(4 + e) * a.T] = (__bf16)vo[e];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    // ---- prologue: K-step 0 whole, the two early half-tiles of K-step 1 ------------------------------------------------
    Cur c1{0, 0, 0, 0}, c2{0, 0, 0, 0};
    cur_set(c1);                 // = K-step 0 for now
    dmaA(0, c1.sa);
    dmaA(16384, c1.sa + 128 * KA2);
    dmaW(32768, c1.sw);
    dmaW(49152, c1.sw + 128 * KW2);
    if (S > 1) cur_next(c1);     // K-step 1
    dmaW(65536 + 32768, c1.sw);
    dmaA(65536, c1.sa);
    c2 = c1;
    if (S > 2) cur_next(c2);     // K-step 2
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    read_x(X0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) asm volatile("s_barrier" ::: "memory");  // the second group runs one barrier behind

    // one K-step = four phases
    auto kstep = [&](int t) {
        const int cb = t & 1, nb = cb ^ 1;
        // phase 0: quadrant (0,0)
        read_w(W0, cb, 0);
        dmaA(nb * 65536 + 16384, c1.sa + 128 * KA2);
        mem_done();
        __builtin_amdgcn_s_setprio(1);
        mma(0, 0, X0, W0);
        __builtin_amdgcn_s_setprio(0);
        cmp_done();
        // phase 1: quadrant (0,1)
        read_w(W1, cb, 1);
        dmaW(nb * 65536 + 49152, c1.sw + 128 * KW2);
        mem_done();
        __builtin_amdgcn_s_setprio(1);
        mma(0, 1, X0, W1);
        __builtin_amdgcn_s_setprio(0);
        cmp_done();
        // phase 2: quadrant (1,1)
        read_x(X1, cb, 1);
        dmaW(cb * 65536 + 32768, c2.sw);
        mem_done();
        __builtin_amdgcn_s_setprio(1);
        mma(1, 1, X1, W1);
        __builtin_amdgcn_s_setprio(0);
        cmp_done();
        // phase 3: quadrant (1,0)
        read_x(X0, nb, 0);
        dmaA(cb * 65536, c2.sa);
        mem_done();
        __builtin_amdgcn_s_setprio(1);
        mma(1, 0, X1, W0);
        __builtin_amdgcn_s_setprio(0);
        cmp_done();
    };
    // (the loop nest keeps the K-steps inside a tile - the hot loop - a loop of its own for the register allocator)
    // (timing experiments: act bit 16 = cycle stamps of workgroup 0, waves 0 and 4, into a.scratch: [tile][slot][group])
    int stamp_tile = 0;
    auto stamp = [&](int slot) {
        if (EPI == GM_EPI_TOK && GM_TIMING(a.act & 16) && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0)
            reinterpret_cast<unsigned long long*>(a.scratch)[(stamp_tile * 8 + slot) * 2 + wr] = __builtin_readcyclecounter();
    };
    int t = 0;
    auto advance = [&]() {
        c1 = c2;
        if (t + 3 < S) cur_next(c2);
        ++t;
    };
    for (int ti = 0; ti < my_tiles; ++ti) {
        stamp_tile = ti;
        stamp(0);
        if (ti > 0 && wr == 1) asm volatile("s_barrier" ::: "memory");  // the second group falls one barrier behind again
        for (int kk = 0; kk + 1 < nk; ++kk) {
            kstep(t);
            advance();
        }
        stamp(1);
        // the tile's bias, one K-step ahead of its use: this lane's quads of the four output pairs (own layout, before the swap)
        f32x4 b00 = {0.f, 0.f, 0.f, 0.f}, b01 = b00, b10 = b00, b11 = b00;
        tile_origin(ti, ep_m0, ep_n0);
        tile_keep_from(ti, ep_mk, ep_nk);
        if (a.bias && EPI != GM_EPI_RAW) {
            // (buffer-resource addressing, one lane-constant register + scalar offsets: as a 64-bit per-lane pointer kept across the tile
            // loop this cost spilled registers in the head-split / fp32-output instantiations, and their reload here - a scratch load,
            // waited for with vmcnt(0) - drained the DMA pipeline once per tile)
            const __amdgpu_buffer_rsrc_t rsB = make_rsrc(a.bias);
            const int bo_ = __builtin_amdgcn_readfirstlane((ep_n0 + wc * 64) * 4);
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int bl = (ln >> 4) * 16;
            b00 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bl, bo_, 0));
            b01 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bl, bo_ + 64, 0));
            b10 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bl, bo_ + 128, 0));
            b11 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bl, bo_ + 192, 0));
        }
        kstep(t);
        // the groups meet before the epilogue (the first one waits out the second's last compute part) and run it side by side:
        // one barrier apart, each group's 7 000-cycle epilogue stalled the other at its next barrier (cycle stamps,
        // scripts/gemm_stamps.py: 14 800 of a K = 1152 tile's 70 000 cycles)
        if (wr == 0) asm volatile("s_barrier" ::: "memory");
        stamp(2);
        slice(I0{}, I0{}, ti, b00, b01);
        stamp(4);
        slice(I0{}, I1{}, ti, b10, b11);
        stamp(5);
        slice(I1{}, I1{}, ti, b10, b11);
        stamp(6);
        slice(I1{}, I0{}, ti, b00, b01);
        stamp(3);
        advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may land in an LDS allocation this workgroup has given up
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {  // compile-time loop (instruction offsets / register indices as constants)
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- one wave per SIMD: 4 waves x (128 tokens x 128 outputs) --------------------------------------------------------------------
// hipBLASLt's kernel for these shapes keeps the matrix pipe 70 % busy where the 8-wave ping-pong above keeps it 51 % busy, with the same
// instruction mix per MFMA (profiles/r03_gemm_vs_hipblaslt_pmc.txt): its wave owns a 128 x 128 output tile - all 512 registers of a
// one-wave-per-SIMD kernel, 256 of them accumulators - so a K-step of 64 is 128 MFMAs fed by 32 fragment reads (the 128 x 64 wave
// tile: 64 MFMAs per 24 reads), and there is no partner wave to be paced against with two barriers per phase.  This is that shape on
// this file's LDS image and LDS-DMA staging: tile 256 x 256, waves 2 x 2, K-step = two sub-steps of 32 (64 MFMAs each, fragments of
// the next sub-step read while this one's MFMAs run), ONE barrier per K-step: the DMA of K-step t + 1 is issued at the start of step
// t into the buffer step t - 1 left, waited for (vmcnt(0)) and handed over before the last 16 MFMAs of step t, behind which the first
// fragments of step t + 1 are read.  Token epilogue only (GM_EPI_TOK).
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4_kernel(const GemmArgs a) {
    constexpr int TN = 256, EPI = GM_EPI_TOK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;  // token half, output half
    const int col = lane & 15, g = lane >> 4;
    // ---- this workgroup's tiles (as gemm_bf16_kernel) ----------------------------------------------------------------
    const int Mt = (a.M + GM_TM - 1) / GM_TM, Nt = (a.N + TN - 1) / TN;
    int mlo = 0, nlo = 0, nx = Nt, first, stride, count;
    if ((gridDim.x & 7) == 0 && a.xn > 0) {
        const int x = (int)blockIdx.x & 7, xm = 8 / a.xn, xi = x / a.xn, xj = x - xi * a.xn;
        mlo = (int)((long long)xi * Mt / xm);
        const int mhi = (int)((long long)(xi + 1) * Mt / xm);
        nlo = xj * Nt / a.xn;
        nx = (xj + 1) * Nt / a.xn - nlo;
        first = (int)blockIdx.x >> 3;
        stride = (int)gridDim.x >> 3;
        count = (mhi - mlo) * nx;
    } else {
        first = (int)blockIdx.x;
        stride = (int)gridDim.x;
        count = Mt * Nt;
    }
    // split-K (EPI == GM_EPI_RAW): a work item is (tile, split); the ks splits of a tile are neighbours in the enumeration (same
    // XCD: they read the same A rows and W rows at different k) and leave fp32 partial sums for gemm_splitk_finish_kernel
    const int ks = 1;
    count *= ks;
    const int my_tiles = first < count ? (count - first + stride - 1) / stride : 0;  // work items of this workgroup
    if (my_tiles == 0) return;
    auto tile_origin = [&](int i, int& m0, int& n0) {
        const int lt = (first + min(i, my_tiles - 1) * stride) / ks;  // (cursors running past the end re-read the last tile)
        const int q = lt / nx;
        m0 = min((mlo + q) * GM_TM, a.M - GM_TM);
        n0 = min((nlo + (lt - q * nx)) * TN, a.N - TN);
    };
    auto item_split = [&](int i) { return (first + min(i, my_tiles - 1) * stride) % ks; };
    // where the tile would start without the shift: rows / columns below are the neighbour tile's and are not stored again
    auto tile_keep_from = [&](int i, int& mk, int& nk_) {
        const int lt = (first + min(i, my_tiles - 1) * stride) / ks;
        const int q = lt / nx;
        mk = (mlo + q) * GM_TM;
        nk_ = (nlo + (lt - q * nx)) * TN;
    };
    const int nk = (a.K / GM_KC) / ks;  // K-steps per work item
    const int S = my_tiles * nk;
    // row pitches in bytes (split-bf16 flavour: A rows hold [hi | lo] = 2 K1 elements, W rows [hi | lo | hi] = 3 K1 = K elements)
    const int KA2 = (a.lda ? a.lda : a.K) * 2, KW2 = (a.ldw ? a.ldw : a.K) * 2;
    // ---- DMA: wave w moves the 8-row groups 4 w .. 4 w + 3 of every half-tile; lane -> (row lane >> 3 of the group, LDS octet
    // position lane & 7, which holds k-octet position ^ ((row >> 1) & 7)) ------------------------------------------------
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A), rsW = make_rsrc(a.W);
    unsigned voffA[4], voffW[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        const unsigned sw = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
        voffA[j] = (unsigned)row * (unsigned)KA2 + sw;
        voffW[j] = (unsigned)row * (unsigned)KW2 + sw;
    }
    // regions of buffer b: A half h at b * 65536 + h * 16384, W half h at b * 65536 + 32768 + h * 16384
    // (the scalar offset through v_readfirstlane: the cursors are uniform by construction, but hipcc kept the A cursor in a VGPR and
    // wrapped each of its DMA instructions in a waterfall loop - readfirstlane, compare, exec mask, branch - inside the K-loop)
    // one piece = one wave instruction = 1 KiB = 8 rows of a half-tile; a half-tile is 16 pieces, 4 per wave
    auto pieceA = [&](int region, int so, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(smem + region + (wave * 4 + j) * 1024), 16, voffA[j], so, 0, 0);
    };
    auto pieceW = [&](int region, int so, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(smem + region + (wave * 4 + j) * 1024), 16, voffW[j], so, 0, 0);
    };
    // cursors over the flat K-steps t + 1 and t + 2: scalar byte offsets of their A / W tiles' first row at their k
    struct Cur {
        int ti, kk, sa, sw;
    };
    auto cur_set = [&](Cur& c) {
        int m0, n0;
        tile_origin(c.ti, m0, n0);
        const int ks_ = item_split(c.ti) * nk + c.kk;                       // K-step of the (virtual) contraction
        const int ka = (a.a_wrap && ks_ >= a.a_wrap) ? ks_ - a.a_wrap : ks_;  // split-bf16: A' = [hi | hi | lo] read out of [hi | lo]
        c.sa = __builtin_amdgcn_readfirstlane(m0 * KA2 + ka * (GM_KC * 2));  // (uniform by construction; keeps the cursor arithmetic on the scalar unit)
        c.sw = __builtin_amdgcn_readfirstlane(n0 * KW2 + ks_ * (GM_KC * 2));
    };
    auto cur_next = [&](Cur& c) {
        if (++c.kk == nk) c.kk = 0, ++c.ti;
        cur_set(c);
    };


    // ---- fragments (v_mfma_f32_32x32x16_bf16: a 32-cycle MFMA hides ONE memory instruction completely - a 16x16x32 one, 16 cycles, does
    // not: measured 15-20 cycles of matrix-pipe idle per LDS / VMEM instruction placed behind it).  Fragment (32-row group g, 16-deep
    // sub-step ks): lane (r = lane & 31, kh = lane >> 5) reads the 16 bytes at row g * 32 + r, k-octet 2 ks + kh of the half-tile. ----------
    const int r32 = lane & 31, kh = lane >> 5;
    uint32_t la[4];  // lane address inside a half-tile at 32-row group 0, per sub-step (the octet swizzle depends on the row pair only)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) la[ks] = (uint32_t)(r32 * 128 + (((2 * ks + kh) ^ ((r32 >> 1) & 7)) * 16));
    const int fxa = wr * 16384, fwb = 32768 + wc * 16384;
    bf16x8 X[2][4], W[2][4];  // [register set = sub-step parity][32-row group]
    f32x16 acc[4][4];         // [token group][output group]: D[output][token] of W X^T
#pragma unroll
    for (int mg = 0; mg < 4; ++mg)
#pragma unroll
        for (int ng = 0; ng < 4; ++ng)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mg][ng][e] = 0.f;
    // staging registers: this wave's 16 pieces (16 bytes per lane each) of one K-step, global -> VGPR -> LDS.  (LDS-DMA, which the other
    // kernels of this file stage with, takes 60-100 cycles of the wave's instruction stream per piece: with one wave per SIMD that is
    // 36 % of a K-step - measured, profiles/r03_gemm_w4_experiments.txt - where the 8-wave kernel hides it behind the partner wave.)
    fg_u32x4 G[16];
    const uint32_t wl = (uint32_t)(uintptr_t)smem + (uint32_t)((wave * 4) * 1024 + lane * 16);  // LDS address of this lane's 16 bytes of piece j = 0

    // ---- epilogue of one 32-token x 32-output accumulator tile (mg, ng).  Lane (token r, half h) holds outputs 8 q + 4 h + (0..3), q = 0..3;
    // v_permlane32_swap of group 2 P with group 2 P + 1 leaves lane h = 0 with outputs 16 P .. 16 P + 7 and lane h = 1 with 16 P + 8 .. + 15
    // of its token: one 16-byte store each, bias / GELU / gate / residual on the eight consecutive outputs as in gemm_bf16_pp_kernel. ------
    int ep_m0 = 0, ep_n0 = 0, ep_mk = 0, ep_nk = 0;
    const int lane_c8 = wc * 128 + 8 * kh;
    const int vo_out = ((wr * 128 + r32) * a.N + lane_c8) * 2;
    const __amdgpu_buffer_rsrc_t rsO = make_rsrc(a.out), rsR = make_rsrc(a.resid), rsG = make_rsrc(a.gate), rsB = make_rsrc(a.bias);
    auto tile_out = [&](auto MG_, auto NG_) {
        constexpr int MG = decltype(MG_)::value, NG = decltype(NG_)::value;
        const int m0 = ep_m0, n0 = ep_n0, mk = ep_mk, nk_ = ep_nk;
        const bool edge = (mk != m0) || (nk_ != n0);
        const int row = m0 + wr * 128 + MG * 32 + r32;
        int bnd = 0x7fffffff, gso = 0;
        if (a.gate) {
            const int gi0 = (a.row0 + m0) / a.gate_rows;
            bnd = (gi0 + 1) * a.gate_rows - a.row0;
            gso = __builtin_amdgcn_readfirstlane((gi0 * a.gate_stride + n0 + 32 * NG) * 4);
        }
        if (W4_EXP & 16) {  // (experiment: no epilogue)
            asm volatile("" : "+a"(acc[MG][NG]));
            return;
        }
        // (the tile leaves the accumulator file through explicit reads: as plain vector arithmetic on acc[][] hipcc moves ALL sixteen tiles
        //  into vector registers for the epilogue - 256 of them - and spills the staging registers, in the K-loop as well)
        float tv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(tv[e]) : "a"(acc[MG][NG][e]));
        {
            f32x16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.f;
            asm volatile("s_nop 0" ::: "memory");
            acc[MG][NG] = z;
            asm volatile("" : "+a"(acc[MG][NG]));
        }
#pragma unroll
        for (int P = 0; P < 2; ++P) {
            float e0 = tv[8 * P], e1 = tv[8 * P + 1], e2 = tv[8 * P + 2], e3 = tv[8 * P + 3];
            float o0 = tv[8 * P + 4], o1 = tv[8 * P + 5], o2 = tv[8 * P + 6], o3 = tv[8 * P + 7];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
                         "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
                         : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
            f32x4 ve = {e0, e1, e2, e3}, vo = {o0, o1, o2, o3};
            const int c8 = n0 + lane_c8 + 32 * NG + 16 * P;
            if (a.bias) {
                const int bo_ = __builtin_amdgcn_readfirstlane((n0 + 32 * NG + 16 * P) * 4);
                ve += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, lane_c8 * 4, bo_, 0));
                vo += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, lane_c8 * 4, bo_ + 16, 0));
            }
            if (a.act & 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ve[e] = gm_gelu_tanh(ve[e]), vo[e] = gm_gelu_tanh(vo[e]);
            }
            if (edge && !(row >= mk && c8 >= nk_)) continue;
            if (a.gate) {
                const int go = gso + 64 * P + (row >= bnd ? a.gate_stride * 4 : 0);
                ve *= __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, lane_c8 * 4 + go - gso, gso, 0));
                vo *= __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, lane_c8 * 4 + go - gso + 16, gso, 0));
            }
            const int so = __builtin_amdgcn_readfirstlane(((m0 + MG * 32) * a.N + n0 + 32 * NG + 16 * P) * 2);
            if (a.resid) {
                const bf16x8 r8 = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsR, vo_out, so, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) ve[e] += (float)r8[e], vo[e] += (float)r8[4 + e];
            }
            bf16x8 o8;
#pragma unroll
            for (int e = 0; e < 4; ++e) o8[e] = (__bf16)ve[e], o8[4 + e] = (__bf16)vo[e];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fg_u32x4, o8), rsO, vo_out, so, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- staging: piece p of a K-step = (region p >> 2: A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255, 8-row group p & 3) ----
    Cur c2{0, 0, 0, 0};  // the K-step the next loads fetch
    cur_set(c2);
    int g_sa0 = 0, g_sa1 = 0, g_sw0 = 0, g_sw1 = 0;
    auto g_cur = [&]() {
        if ((W4_EXP & 4) && g_sa1 != 0) return;  // (experiment: every K-step re-reads the first one's operands - cache-resident)
        g_sa0 = c2.sa, g_sa1 = __builtin_amdgcn_readfirstlane(c2.sa + 128 * KA2), g_sw0 = c2.sw, g_sw1 = __builtin_amdgcn_readfirstlane(c2.sw + 128 * KW2);
    };
    auto g_load = [&](auto P_) {
        constexpr int P = decltype(P_)::value, R = P >> 2, J = P & 3;
        if (W4_EXP & 1) return;
        if constexpr (R == 0) G[P] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA[J], g_sa0, 0);
        else if constexpr (R == 1) G[P] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA[J], g_sa1, 0);
        else if constexpr (R == 2) G[P] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voffW[J], g_sw0, 0);
        else G[P] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voffW[J], g_sw1, 0);
    };
    auto g_store = [&](auto P_, uint32_t base) {  // base = wl + buffer * 65536
        constexpr int P = decltype(P_)::value;
        if (W4_EXP & 1) return;
        const fg_u32x4& gp = G[P];
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(base), "v"(gp), "n"((P >> 2) * 16384 + (P & 3) * 1024) : "memory");
    };
    auto SBAR = [] { __builtin_amdgcn_sched_barrier(0); };
    // fragment i of a sub-step: i = 0..3 token groups, 4..7 output groups
    auto rd_f = [&](int set, auto I_, uint32_t xb, uint32_t wb) {  // xb / wb = buffer + half-tile + la[ks]
        constexpr int I = decltype(I_)::value;
        if (W4_EXP & 2) return;
        if constexpr (I < 4) {
            bf16x8& f = X[set][I];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(xb), "n"(I * 4096));
        } else {
            bf16x8& f = W[set][I - 4];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(wb), "n"((I - 4) * 4096));
        }
    };
    const uint32_t s0 = (uint32_t)(uintptr_t)smem;

    // ---- prologue: K-step 0 into buffer 0 through the registers, the loads of K-step 1 in flight, first fragments read --------------------
    g_cur();
    static_for<0, 16>([&](auto P_) { g_load(P_); });
    if (S > 1) cur_next(c2);
    static_for<0, 16>([&](auto P_) { g_store(P_, wl); });  // (the compiler waits for each load in front of its store)
    g_cur();
    static_for<0, 16>([&](auto P_) { g_load(P_); });      // K-step 1
    if (S > 2) cur_next(c2);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    static_for<0, 8>([&](auto I_) { rd_f(0, I_, s0 + fxa + la[0], s0 + fwb + la[0]); });

    // One K-step of 64 = four sub-steps of 16 MFMAs (4 token groups x 4 output groups), ONE memory instruction behind every MFMA:
    //   slots 0-7 of sub-step ks: the eight fragments of sub-step ks + 1 (of the NEXT K-step, other buffer, for ks = 3)
    //   slots 8-15: ks = 0, 1: the staged K-step t + 1 goes into the other LDS buffer (2 x 8 ds_write_b128)
    //               ks = 2, 3: the loads of K-step t + 2 into the same registers (2 x 8)
    //   between ks = 2 and 3: this wave's writes are done (lgkmcnt(0)), then the one barrier of the step
    auto kstep = [&](int t) {
        const int cb = t & 1, nb = cb ^ 1;
        const uint32_t xcb = s0 + cb * 65536 + fxa, wcb = s0 + cb * 65536 + fwb, xnb = s0 + nb * 65536 + fxa, wnb_ = s0 + nb * 65536 + fwb;
        const uint32_t wst = wl + nb * 65536;
        static_for<0, 4>([&](auto KS_) {
            constexpr int KS = decltype(KS_)::value, SET = KS & 1;
            // this sub-step's fragments (read in slots 0-7 of the previous one; behind them at most its 8 LDS writes)
            if constexpr (KS == 1 || KS == 2) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (KS == 3) {
                asm volatile("s_barrier" ::: "memory");  // K-step t + 1 is in LDS, in every wave's part
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(X[SET][i]), "+v"(W[SET][i]));
            SBAR();
            if constexpr (KS == 2) g_cur();
            static_for<0, 16>([&](auto I_) {
                constexpr int I = decltype(I_)::value, MG = I >> 2, NG = I & 3;
                acc[MG][NG] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[SET][NG], X[SET][MG], acc[MG][NG], 0, 0, 0);
                if constexpr (I < 8) {
                    if constexpr (KS < 3) rd_f(SET ^ 1, I_, xcb + la[KS < 3 ? KS + 1 : 0], wcb + la[KS < 3 ? KS + 1 : 0]);
                    else rd_f(SET ^ 1, I_, xnb + la[0], wnb_ + la[0]);
                } else {
                    if constexpr (KS < 2) g_store(std::integral_constant<int, 8 * KS + I - 8>{}, wst);
                    else g_load(std::integral_constant<int, 8 * (KS - 2) + I - 8>{});
                }
                SBAR();
            });
        });
    };
    int t = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        for (int kk = 0; kk < nk; ++kk) {  // (ONE copy of the K-step in the kernel: a second, peeled one gets its own accumulator registers
            kstep(t);                      //  and 240 accumulator-to-accumulator moves to reconcile them)
            if (t + 3 < S) cur_next(c2);
            ++t;
        }
        tile_origin(ti, ep_m0, ep_n0);
        tile_keep_from(ti, ep_mk, ep_nk);
        static_for<0, 16>([&](auto I_) { tile_out(std::integral_constant<int, (decltype(I_)::value >> 2)>{}, std::integral_constant<int, (decltype(I_)::value & 3)>{}); });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// ---- the narrow tile: 256 tokens x 128 outputs, for SHORT token counts ---------------------------------------------------------
// One sample of the causal video DiT hands the block linears M = 4 680 tokens (a chunk of 3 latent frames): 19 token tiles.  With
// 256-wide tiles N = 1 536 gives 114 tiles for 256 CUs; the kernel above then cuts K in two (fp32 partial sums through HBM + a finishing
// pass: 67 us for 22 GFLOP = 330 TFLOP/s for the attention-output and cross-attention projections), and at N = 4 608 it runs two rounds of
// 342 tiles at two thirds occupancy.  With 128-wide tiles the same shapes are 228 / 684 work items and need neither.
// Same ping-pong structure (8 waves = 2 token halves x 4 output eighths of 32, waves 4-7 one barrier behind), but a wave's 128 x 32
// outputs are TWO quadrants of 16 MFMAs, so a K-step is two phases, and the 48 KiB stage (A rows 0-127 | A rows 128-255 | W) sits in a
// ring of THREE: the DMA of K-step t + 2 goes into the buffer K-step t - 1 left.
//   phase 0: read W(t), X(t) m 4-7       issue A (both halves) of t+2      16 MFMAs: m 0-3
//   phase 1: read X(t+1) m 0-3           issue W of t+2                    16 MFMAs: m 4-7
// Every memory part ends with vmcnt(6): behind the half-tile that the NEXT phase reads, a wave has issued exactly six younger pieces
// (A: 4 per wave and K-step, W: 2).  LDS reads per MFMA are 1.7 x the wide tile's (20 fragments per 32 MFMAs): about 60 % of the LDS
// bandwidth at the MFMA roof - the price of the narrow tile, paid only where the wide one leaves CUs idle.
__global__ __launch_bounds__(GM_NTHR) void gemm_bf16_pp2_kernel(const GemmArgs a) {
    constexpr int TN = 128;
    constexpr int BUFB = 49152;  // bytes per stage: A rows 0-127 at 0, A rows 128-255 at 16384, W at 32768
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;  // token half (= ping-pong group), output eighth (32 outputs)
    const int col = lane & 15, g = lane >> 4;

    // ---- this workgroup's tiles (as gemm_bf16_pp_kernel) ----------------------------------------------------------------
    const int Mt = (a.M + GM_TM - 1) / GM_TM, Nt = (a.N + TN - 1) / TN;
    int mlo = 0, nlo = 0, nx = Nt, first, stride, count;
    if ((gridDim.x & 7) == 0 && a.xn > 0) {
        const int x = (int)blockIdx.x & 7, xm = 8 / a.xn, xi = x / a.xn, xj = x - xi * a.xn;
        mlo = (int)((long long)xi * Mt / xm);
        const int mhi = (int)((long long)(xi + 1) * Mt / xm);
        nlo = xj * Nt / a.xn;
        nx = (xj + 1) * Nt / a.xn - nlo;
        first = (int)blockIdx.x >> 3;
        stride = (int)gridDim.x >> 3;
        count = (mhi - mlo) * nx;
    } else {
        first = (int)blockIdx.x;
        stride = (int)gridDim.x;
        count = Mt * Nt;
    }
    const int my_tiles = first < count ? (count - first + stride - 1) / stride : 0;
    if (my_tiles == 0) return;
    auto tile_origin = [&](int i, int& m0, int& n0) {
        const int lt = first + min(i, my_tiles - 1) * stride;  // (cursors running past the end re-read the last tile)
        const int q = lt / nx;
        m0 = min((mlo + q) * GM_TM, a.M - GM_TM);
        n0 = min((nlo + (lt - q * nx)) * TN, a.N - TN);
    };
    auto tile_keep_from = [&](int i, int& mk, int& nk_) {
        const int lt = first + min(i, my_tiles - 1) * stride;
        const int q = lt / nx;
        mk = (mlo + q) * GM_TM;
        nk_ = (nlo + (lt - q * nx)) * TN;
    };
    const int nk = a.K / GM_KC;
    const int S = my_tiles * nk;
    const int KA2 = a.K * 2, KW2 = a.K * 2;
    // ---- DMA: wave w moves the 8-row groups 2 w, 2 w + 1 of every 128-row half-tile (see gemm_bf16_pp_kernel) -------------------
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A), rsW = make_rsrc(a.W);
    unsigned voffA[2], voffW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 8 + (lane >> 3);
        const unsigned sw = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
        voffA[j] = (unsigned)row * (unsigned)KA2 + sw;
        voffW[j] = (unsigned)row * (unsigned)KW2 + sw;
    }
    auto dmaA = [&](int region, int soff) {
        const int so = __builtin_amdgcn_readfirstlane(soff);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(smem + region + (wave * 2 + j) * 1024), 16, voffA[j], so, 0, 0);
    };
    auto dmaW = [&](int region, int soff) {
        const int so = __builtin_amdgcn_readfirstlane(soff);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(smem + region + (wave * 2 + j) * 1024), 16, voffW[j], so, 0, 0);
    };
    struct Cur {
        int ti, kk, sa, sw;
    };
    auto cur_set = [&](Cur& c) {
        int m0, n0;
        tile_origin(c.ti, m0, n0);
        c.sa = __builtin_amdgcn_readfirstlane(m0 * KA2 + c.kk * (GM_KC * 2));
        c.sw = __builtin_amdgcn_readfirstlane(n0 * KW2 + c.kk * (GM_KC * 2));
    };
    auto cur_next = [&](Cur& c) {
        if (++c.kk == nk) c.kk = 0, ++c.ti;
        cur_set(c);
    };
    auto stage = [&](int buf, const Cur& c, int what) {  // what: 1 = A (both halves), 2 = W, 3 = both
        if (what & 1) {
            dmaA(buf * BUFB, c.sa);
            dmaA(buf * BUFB + 16384, c.sa + 128 * KA2);
        }
        if (what & 2) dmaW(buf * BUFB + 32768, c.sw);
    };

    // ---- fragments --------------------------------------------------------------------------------------------------------
    const int lrow0 = col * 128 + ((g ^ ((col >> 1) & 7)) * 16), lrow1 = lrow0 ^ 64;  // k-substep 0 / 1
    const int fxa = wr * 16384;         // this wave's A half
    const int fwb = 32768 + wc * 4096;  // this wave's 32 W rows
    bf16x8 X0[4][2], X1[4][2], W0[2][2];  // [m | nt][k-substep]
    f32x4 acc[8][2];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto read_x = [&](bf16x8 (&x)[4][2], int bufoff, int mh) {
        const int a0 = bufoff + fxa + lrow0, a1 = bufoff + fxa + lrow1;
        if (mh == 0) {
            GM_DSR(x[0][0], a0, 0); GM_DSR(x[0][1], a1, 0); GM_DSR(x[1][0], a0, 2048); GM_DSR(x[1][1], a1, 2048);
            GM_DSR(x[2][0], a0, 4096); GM_DSR(x[2][1], a1, 4096); GM_DSR(x[3][0], a0, 6144); GM_DSR(x[3][1], a1, 6144);
        } else {
            GM_DSR(x[0][0], a0, 8192); GM_DSR(x[0][1], a1, 8192); GM_DSR(x[1][0], a0, 10240); GM_DSR(x[1][1], a1, 10240);
            GM_DSR(x[2][0], a0, 12288); GM_DSR(x[2][1], a1, 12288); GM_DSR(x[3][0], a0, 14336); GM_DSR(x[3][1], a1, 14336);
        }
    };
    auto read_w = [&](bf16x8 (&w)[2][2], int bufoff) {
        const int a0 = bufoff + fwb + lrow0, a1 = bufoff + fwb + lrow1;
        GM_DSR(w[0][0], a0, 0); GM_DSR(w[0][1], a1, 0); GM_DSR(w[1][0], a0, 2048); GM_DSR(w[1][1], a1, 2048);
    };
    auto mma = [&](int mh, const bf16x8 (&x)[4][2], const bf16x8 (&w)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mh * 4 + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nt][j], x[m][j], acc[mh * 4 + m][nt], 0, 0, 0);
    };
    auto mem_done = [&]() {
        asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto cmp_done = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
    };

    // ---- token epilogue of one half (MH: 64 tokens x 32 outputs of this wave), as gemm_bf16_pp_kernel's -----------------------------
    int ep_m0 = 0, ep_n0 = 0, ep_mk = 0, ep_nk = 0;
    const int lane_c8 = wc * 32 + 16 * (g & 1) + 8 * (g >> 1);
    const int vo_out = ((wr * 128 + col) * a.N + lane_c8) * 2;
    const __amdgpu_buffer_rsrc_t rsO = make_rsrc(a.out), rsR = make_rsrc(a.resid), rsG = make_rsrc(a.gate);
    auto slice = [&](auto MH_, f32x4 be, f32x4 bo) {
        constexpr int MH = decltype(MH_)::value;
        const int m0 = ep_m0, n0 = ep_n0, mk = ep_mk, nk_ = ep_nk;
        const bool edge = (mk != m0) || (nk_ != n0);  // a shifted last tile: part of it is its neighbour's (uniform)
        const int c8 = n0 + lane_c8;
        f32x4 g0e = {1.f, 1.f, 1.f, 1.f}, g0o = g0e, g1e = g0e, g1o = g0e;
        int bnd = 0x7fffffff;
        if (a.gate) {
            const int gi0 = (a.row0 + m0) / a.gate_rows;
            bnd = (gi0 + 1) * a.gate_rows - a.row0;  // first token (in this launch's rows) of the next gate period
            const int gso = __builtin_amdgcn_readfirstlane((gi0 * a.gate_stride + n0) * 4);
            g0e = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, lane_c8 * 4, gso, 0));
            g0o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, lane_c8 * 4, gso + 16, 0));
            g1e = g0e, g1o = g0o;
            if (bnd < m0 + GM_TM) {
                g1e = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, lane_c8 * 4, gso + a.gate_stride * 4, 0));
                g1o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, lane_c8 * 4, gso + a.gate_stride * 4 + 16, 0));
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int rb = (MH * 4 + m) * 16;  // 16-row block inside the wave's token half
            const int row = m0 + wr * 128 + rb + col;
            f32x4 ve = acc[MH * 4 + m][0] + be, vo = acc[MH * 4 + m][1] + bo;
            acc[MH * 4 + m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[MH * 4 + m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.act & 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ve[e] = gm_gelu_tanh(ve[e]), vo[e] = gm_gelu_tanh(vo[e]);
            }
            {
                float e0 = ve[0], e1 = ve[1], e2 = ve[2], e3 = ve[3], o0 = vo[0], o1 = vo[1], o2 = vo[2], o3 = vo[3];
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\t"
                             "v_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                             : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
                ve = f32x4{e0, e1, e2, e3};
                vo = f32x4{o0, o1, o2, o3};
            }
            if (edge && !(row >= mk && c8 >= nk_)) continue;
            if (a.gate) {
                const bool nx_ = row >= bnd;
                ve *= nx_ ? g1e : g0e;
                vo *= nx_ ? g1o : g0o;
            }
            const int so = __builtin_amdgcn_readfirstlane(((m0 + rb) * a.N + n0) * 2);  // (scalar; < 2 GiB: launch_gemm_bf16 chunks the rows)
            if (a.resid) {
                const bf16x8 r8 = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsR, vo_out, so, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) ve[e] += (float)r8[e], vo[e] += (float)r8[4 + e];
            }
            bf16x8 o8;
#pragma unroll
            for (int e = 0; e < 4; ++e) o8[e] = (__bf16)ve[e], o8[4 + e] = (__bf16)vo[e];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fg_u32x4, o8), rsO, vo_out, so, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    // ---- prologue: K-steps 0 and 1 whole -----------------------------------------------------------------------------------------
    Cur c2{0, 0, 0, 0};
    cur_set(c2);
    stage(0, c2, 3);
    if (S > 1) cur_next(c2);
    stage(1, c2, 3);
    if (S > 2) cur_next(c2);  // = K-step 2
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    read_x(X0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) asm volatile("s_barrier" ::: "memory");  // the second group runs one barrier behind

    int t = 0, b0 = 0;  // flat K-step index, byte offset of its stage
    auto kstep = [&]() {
        const int b1 = b0 + BUFB >= 3 * BUFB ? 0 : b0 + BUFB;   // stage of K-step t + 1
        const int b2 = b1 + BUFB >= 3 * BUFB ? 0 : b1 + BUFB;   // ... of t + 2 (= the one K-step t - 1 left)
        // phase 0
        read_w(W0, b0);
        read_x(X1, b0, 1);
        stage(b2 / BUFB, c2, 1);
        mem_done();
        __builtin_amdgcn_s_setprio(1);
        mma(0, X0, W0);
        __builtin_amdgcn_s_setprio(0);
        cmp_done();
        // phase 1
        read_x(X0, b1, 0);
        stage(b2 / BUFB, c2, 2);
        mem_done();
        __builtin_amdgcn_s_setprio(1);
        mma(1, X1, W0);
        __builtin_amdgcn_s_setprio(0);
        cmp_done();
        if (t + 3 < S) cur_next(c2);
        ++t;
        b0 = b1;
    };
    for (int ti = 0; ti < my_tiles; ++ti) {
        if (ti > 0 && wr == 1) asm volatile("s_barrier" ::: "memory");  // the second group falls one barrier behind again
        for (int kk = 0; kk + 1 < nk; ++kk) kstep();
        f32x4 b0v = {0.f, 0.f, 0.f, 0.f}, b1v = b0v;
        tile_origin(ti, ep_m0, ep_n0);
        tile_keep_from(ti, ep_mk, ep_nk);
        if (a.bias) {
            const __amdgpu_buffer_rsrc_t rsB = make_rsrc(a.bias);
            const int bo_ = __builtin_amdgcn_readfirstlane((ep_n0 + wc * 32) * 4);
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int bl = (ln >> 4) * 16;
            b0v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bl, bo_, 0));
            b1v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bl, bo_ + 64, 0));
        }
        kstep();
        if (wr == 0) asm volatile("s_barrier" ::: "memory");  // the groups meet before the epilogue and run it side by side
        slice(I0{}, b0v, b1v);
        slice(I1{}, b0v, b1v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may land in an LDS allocation this workgroup has given up
}

// out[row][c .. c + 7] = epilogue(sum over the K splits of their fp32 partial sums): bias, tanh-GELU, gate x value + residual as in the
// kernels' own token epilogue
__global__ void gemm_splitk_finish_kernel(const GemmArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int N8 = a.N / 8;
    if (i >= (int64_t)a.M * N8) return;
    const int row = (int)(i / N8), c = (int)(i - (int64_t)row * N8) * 8;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    for (int s = 0; s < a.ksplit; ++s) {
        const float* p = a.scratch + ((size_t)s * a.M + row) * a.N + c;
        v0 += *reinterpret_cast<const f32x4*>(p);
        v1 += *reinterpret_cast<const f32x4*>(p + 4);
    }
    if (a.bias) v0 += *reinterpret_cast<const f32x4*>(a.bias + c), v1 += *reinterpret_cast<const f32x4*>(a.bias + c + 4);
    if (a.act & 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v0[e] = gm_gelu_tanh(v0[e]), v1[e] = gm_gelu_tanh(v1[e]);
    }
    if (a.gate) {
        const float* gr = a.gate + (size_t)((a.row0 + row) / a.gate_rows) * a.gate_stride + c;
        v0 *= *reinterpret_cast<const f32x4*>(gr);
        v1 *= *reinterpret_cast<const f32x4*>(gr + 4);
    }
    if (a.resid) {
        const bf16x8 r8 = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.resid) + (size_t)row * a.N + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v0[e] += (float)r8[e], v1[e] += (float)r8[4 + e];
    }
    bf16x8 o8;
#pragma unroll
    for (int e = 0; e < 4; ++e) o8[e] = (__bf16)v0[e], o8[4 + e] = (__bf16)v1[e];
    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(a.out) + (size_t)row * a.N + c) = o8;
}

__global__ void cvt_bf16_kernel(const float* __restrict__ in, __bf16* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (__bf16)in[i];
}

// the split-bf16 flavour's operands: W' [N][3 K] = [hi | lo | hi] of the fp32 weight, planes [M][2 K] = [hi | lo] of an fp32 tensor
__global__ void split3_weights_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int N, int K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    const int64_t n = i / K;
    const int k = (int)(i - n * K);
    const float v = w[i];
    const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
    __bf16* o = out + n * 3 * K;
    o[k] = hi, o[K + k] = lo, o[2 * K + k] = hi;
}
__global__ void split_planes_kernel(const float* __restrict__ x, __bf16* __restrict__ out, int64_t M, int K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row, 4 columns)
    const int K4 = K / 4;
    if (i >= M * K4) return;
    const int64_t r = i / K4;
    const int c = (int)(i - r * K4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * K + c);
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) hi[e] = (__bf16)v[e], lo[e] = (__bf16)(v[e] - (float)hi[e]);
    *reinterpret_cast<bf16x4*>(out + r * 2 * K + c) = hi;
    *reinterpret_cast<bf16x4*>(out + r * 2 * K + K + c) = lo;
}

int g_gm_cus[16] = {};

template <int NT, int EPI>
int launch_gm(const GemmArgs& a, hipStream_t s, bool prepare_only) {
    constexpr int TN = 64 * NT;
    constexpr int LDS = 2 * 8 * (GM_PA + TN * 16 + 16);
    auto kern = gemm_bf16_kernel<NT, EPI>;
    const int dev = fg_device_slot();
    if (dev < 0) return (int)hipErrorInvalidDevice;
    static bool attr_done[16] = {};
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[dev] = true;
    }
    if (!g_gm_cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return (int)hipErrorUnknown;
        g_gm_cus[dev] = n;
    }
    if (prepare_only) return 0;
    const int Mt = (a.M + GM_TM - 1) / GM_TM, Nt = (a.N + TN - 1) / TN;
    GemmArgs b = a;
    int grid = g_gm_cus[dev];
    if ((long long)Mt * Nt < grid) grid = Mt * Nt, b.xn = 0;
    if (grid & 7) b.xn = 0;
    if (b.xn > 0) {
        // XCD grid xm x xn: the widest split of the output tiles that divides them and still leaves every XCD >= 4 token tiles;
        // an explicit a.xn (tuning) is taken when it divides
        int xn = 1;
        for (int c = 8; c >= 1; c >>= 1)
            if (Nt % c == 0 && Mt / (8 / c) >= 4) {
                xn = c;
                break;
            }
        if (a.xn <= 8 && (a.xn & (a.xn - 1)) == 0 && Nt % a.xn == 0 && a.xn != 1) xn = a.xn;
        b.xn = xn;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(GM_NTHR), LDS, s, b);
    return (int)hipGetLastError();
}

template <int EPI>
int launch_pp(const GemmArgs& a, hipStream_t s, bool prepare_only) {
    constexpr int LDS = 131072;
    auto kern = gemm_bf16_pp_kernel<EPI>;
    const int dev = fg_device_slot();
    if (dev < 0) return (int)hipErrorInvalidDevice;
    static bool attr_done[16] = {};
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[dev] = true;
    }
    if (!g_gm_cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return (int)hipErrorUnknown;
        g_gm_cus[dev] = n;
    }
    if (prepare_only) return 0;
    const int Mt = (a.M + GM_TM - 1) / GM_TM, Nt = (a.N + 255) / 256;
    GemmArgs b = a;
    int grid = g_gm_cus[dev];
    const long long items = (long long)Mt * Nt * (EPI == GM_EPI_RAW ? a.ksplit : 1);
    if (items < grid) grid = (int)items, b.xn = 0;
    if (grid & 7) b.xn = 0;
    if (b.xn > 0) {
        int xn = 1;
        for (int c = 8; c >= 1; c >>= 1)
            if (Nt % c == 0 && Mt / (8 / c) >= 4) {
                xn = c;
                break;
            }
        if (a.xn <= 8 && (a.xn & (a.xn - 1)) == 0 && Nt % a.xn == 0 && a.xn != 1) xn = a.xn;
        b.xn = xn;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(GM_NTHR), LDS, s, b);
    if (EPI == GM_EPI_RAW) {
        const int64_t n = (int64_t)a.M * (a.N / 8);
        hipLaunchKernelGGL(gemm_splitk_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, b);
    }
    return (int)hipGetLastError();
}

// the narrow-tile kernel (256 x 128): token epilogue only
// the one-wave-per-SIMD kernel (256 x 256 tiles, 4 waves): token epilogue only
int launch_w4(const GemmArgs& a, hipStream_t s, bool prepare_only) {
    constexpr int LDS = 131072;
    const int dev = fg_device_slot();
    if (dev < 0) return (int)hipErrorInvalidDevice;
    static bool attr_done[16] = {};
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_w4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[dev] = true;
    }
    if (!g_gm_cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return (int)hipErrorUnknown;
        g_gm_cus[dev] = n;
    }
    if (prepare_only) return 0;
    const int Mt = (a.M + GM_TM - 1) / GM_TM, Nt = (a.N + 255) / 256;
    GemmArgs b = a;
    int grid = g_gm_cus[dev];
    const long long items = (long long)Mt * Nt;
    if (items < grid) grid = (int)items, b.xn = 0;
    if (grid & 7) b.xn = 0;
    if (b.xn > 0) {
        int xn = 1;
        for (int c = 8; c >= 1; c >>= 1)
            if (Nt % c == 0 && Mt / (8 / c) >= 4) {
                xn = c;
                break;
            }
        if (a.xn <= 8 && (a.xn & (a.xn - 1)) == 0 && Nt % a.xn == 0 && a.xn != 1) xn = a.xn;
        b.xn = xn;
    }
    hipLaunchKernelGGL(gemm_bf16_w4_kernel, dim3(grid), dim3(256), LDS, s, b);
    return (int)hipGetLastError();
}

int launch_pp2(const GemmArgs& a, hipStream_t s, bool prepare_only) {
    constexpr int LDS = 3 * 49152;
    const int dev = fg_device_slot();
    if (dev < 0) return (int)hipErrorInvalidDevice;
    static bool attr_done[16] = {};
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pp2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[dev] = true;
    }
    if (!g_gm_cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return (int)hipErrorUnknown;
        g_gm_cus[dev] = n;
    }
    if (prepare_only) return 0;
    const int Mt = (a.M + GM_TM - 1) / GM_TM, Nt = (a.N + 127) / 128;
    GemmArgs b = a;
    int grid = g_gm_cus[dev];
    const long long items = (long long)Mt * Nt;
    if (items < grid) grid = (int)items, b.xn = 0;
    if (grid & 7) b.xn = 0;
    if (b.xn > 0) {
        int xn = 1;
        for (int c = 8; c >= 1; c >>= 1)
            if (Nt % c == 0 && Mt / (8 / c) >= 4) {
                xn = c;
                break;
            }
        if (a.xn <= 8 && (a.xn & (a.xn - 1)) == 0 && Nt % a.xn == 0 && a.xn != 1) xn = a.xn;
        b.xn = xn;
    }
    hipLaunchKernelGGL(gemm_bf16_pp2_kernel, dim3(grid), dim3(GM_NTHR), LDS, s, b);
    return (int)hipGetLastError();
}

// Which short grids take the narrow tile: the 256-wide tiling would leave CUs idle (fewer than 2.5 tiles per CU) and the 128-wide one
// fills at least 85 % of its last round (or the grid is so short that split-K was the alternative).
bool gm_use_narrow(const GemmArgs& a, int cus) {
    if (a.heads > 0 || a.out_f32 || a.N < 128 || a.M < GM_TM || (a.N % 16)) return false;
    const long long wide = (long long)((a.M + GM_TM - 1) / GM_TM) * ((a.N + 255) / 256);
    const long long narrow = (long long)((a.M + GM_TM - 1) / GM_TM) * ((a.N + 127) / 128);
    if (wide * 2 >= 5LL * cus) return false;
    if (wide * 2 <= cus) return true;  // split-K territory
    const double fill_w = (double)wide / (double)(((wide + cus - 1) / cus) * cus), fill_n = (double)narrow / (double)(((narrow + cus - 1) / cus) * cus);
    return fill_n > fill_w + 0.05;
}

// Split-K for short grids: with fewer 256 x 256 tiles than half the CUs (the video DiT's 4 680-token chunk x 1 536 outputs = 114
// tiles) the K range of every tile is cut into ks pieces (ks | K / 64, tiles x ks <= CUs) that leave fp32 partial sums in the
// caller's scratch; a finishing pass sums them and applies the token epilogue.
int gm_pick_ksplit(const GemmArgs& a, int cus) {
    if (!a.scratch || a.heads > 0 || (a.N % 8)) return 1;
    const long long tiles = (long long)((a.M + GM_TM - 1) / GM_TM) * ((a.N + 255) / 256);
    const int nk = a.K / GM_KC;
    if (tiles * 2 > cus || nk < 16) return 1;
    for (int ks = 4; ks >= 2; --ks)
        if (nk % ks == 0 && tiles * ks <= cus && nk / ks >= 8 && (size_t)ks * a.M * a.N * 4 <= a.scratch_bytes) return ks;
    return 1;
}

int gm_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
// FASTGEN_AMD_GEMM_NARROW=0: never the 256 x 128 tile (A/B measurements)
bool gm_narrow_enabled() {
    static const bool on = [] {
        const char* e = getenv("FASTGEN_AMD_GEMM_NARROW");
        return !(e && e[0] == '0');
    }();
    return on;
}
// 0: gemm_bf16_kernel (register-staged), 1: gemm_bf16_pp_kernel (LDS-DMA, ping-pong) where the shape allows
int gm_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FASTGEN_AMD_GEMM_PP");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v;
}

}  // namespace

bool gemm_bf16_supported(const GemmArgs& a) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % GM_KC) || (a.N % 16)) return false;
    if ((size_t)a.N * a.K * 2 >= (1ull << 31)) return false;  // staging offsets into W stay below the buffer resource's 2 GiB
    if (a.heads > 0 && (a.head_dim % 4 || a.N != 3 * a.heads * a.head_dim || a.T <= 0 || !a.q || !a.k || !a.vt)) return false;
    if (a.heads <= 0 && !a.out) return false;
    if (a.gate && a.gate_rows <= 0) return false;
    return a.A && a.W;
}

// xn: 0 = linear tile order, 1 = choose the XCD grid, 2 / 4 / 8 = force that many output-tile columns of XCDs
int launch_gemm_bf16(const GemmArgs& a, hipStream_t s, bool prepare_only) {
    if (!prepare_only && !gemm_bf16_supported(a)) return (int)hipErrorInvalidValue;
    const bool heads = a.heads > 0;
    const bool n3 = !prepare_only && (a.N % 256) != 0 && (a.N % 192) == 0;
    if (prepare_only) {
        int rc;
        if ((rc = launch_gm<4, GM_EPI_TOK>(a, s, true)) || (rc = launch_gm<3, GM_EPI_TOK>(a, s, true)) ||
            (rc = launch_gm<4, GM_EPI_HEADS>(a, s, true)) || (rc = launch_gm<3, GM_EPI_HEADS>(a, s, true)) ||
            (rc = launch_pp<GM_EPI_TOK>(a, s, true)) || (rc = launch_pp<GM_EPI_HEADS>(a, s, true)) || (rc = launch_pp<GM_EPI_RAW>(a, s, true)) ||
            (rc = launch_pp<GM_EPI_TOK32>(a, s, true)) || (rc = launch_pp<GM_EPI_SPLIT>(a, s, true)) || (rc = launch_pp<GM_EPI_HEADS32>(a, s, true)) ||
            (rc = launch_pp2(a, s, true)))
            return rc;
        return 0;
    }
    // rows per launch: staging offsets into A stay below the buffer resource's 2 GiB
    // (... and the epilogue's offsets into out / resid: N instead of K)
    const int rows_max = (int)(((1ull << 31) - 1) / ((size_t)(a.K > a.N ? a.K : a.N) * 2)) / GM_TM * GM_TM;
    for (int r0 = 0; r0 < a.M; r0 += rows_max) {
        GemmArgs b = a;
        b.row0 = a.row0 + r0;
        b.M = a.M - r0 < rows_max ? a.M - r0 : rows_max;
        b.A = reinterpret_cast<const __bf16*>(a.A) + (size_t)r0 * a.K;
        if (a.out) b.out = reinterpret_cast<__bf16*>(a.out) + (size_t)r0 * a.N;
        if (a.resid) b.resid = reinterpret_cast<const __bf16*>(a.resid) + (size_t)r0 * a.N;
        const bool pp = (a.variant < 0 ? gm_variant() : (a.variant >= 2 ? 1 : a.variant)) == 1 && b.M >= GM_TM && a.N >= 256 && a.K >= 2 * GM_KC &&
                        (!a.gate || a.gate_rows >= GM_TM);  // (the ping-pong kernel's token epilogue: a tile spans at most two gate rows)
        if (pp && !heads && !a.out_f32 && rows_max >= a.M) {
            const int dev = fg_device_slot();
            const int cus = dev >= 0 && g_gm_cus[dev] ? g_gm_cus[dev] : 256;
            if (a.variant == 3) {  // (experimental: one wave per SIMD)
                if (b.M < GM_TM || b.N < 256) return (int)hipErrorInvalidValue;
                const int rc4 = launch_w4(b, s, false);
                if (rc4) return rc4;
                continue;
            }
            if (a.variant == 2 || (a.variant < 0 && gm_narrow_enabled() && gm_use_narrow(b, cus))) {
                const int rc0 = launch_pp2(b, s, false);
                if (rc0) return rc0;
                continue;
            }
            b.ksplit = gm_pick_ksplit(b, cus);
            if (b.ksplit > 1) {
                const int rc2 = launch_pp<GM_EPI_RAW>(b, s, false);
                if (rc2) return rc2;
                continue;
            }
        }
        if (a.out_f32) {
            if (!pp || heads || rows_max < a.M) return (int)hipErrorInvalidValue;
            const int rc3 = launch_pp<GM_EPI_TOK32>(b, s, false);
            if (rc3) return rc3;
            continue;
        }
        const int rc = pp ? (heads ? launch_pp<GM_EPI_HEADS>(b, s, false) : launch_pp<GM_EPI_TOK>(b, s, false))
                          : heads ? (n3 ? launch_gm<3, GM_EPI_HEADS>(b, s, false) : launch_gm<4, GM_EPI_HEADS>(b, s, false))
                                  : (n3 ? launch_gm<3, GM_EPI_TOK>(b, s, false) : launch_gm<4, GM_EPI_TOK>(b, s, false));
        if (rc) return rc;
    }
    return 0;
}

int launch_cvt_bf16(const float* in, void* out, size_t n, hipStream_t s) {
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, s, in, (__bf16*)out, n);
    return (int)hipGetLastError();
}

// ---- split-bf16 ("bf16x3", the fp32-grade mode of common.h) on the same kernel -----------------------------------------------
// sum_k a_k w_k with a = a_hi + a_lo, w = w_hi + w_lo evaluated as a_hi w_hi + a_hi w_lo + a_lo w_hi is ONE bf16 contraction of
// length 3 K: A' = [a_hi | a_hi | a_lo], W' = [w_hi | w_lo | w_hi].  A is stored as [hi | lo] planes ([M][2 K] bf16, the same bytes
// as the fp32 tensor; its hi plane is read twice: a_wrap) by the producing kernel (LayerNorm-modulate, the SPLIT epilogue of the
// previous GEMM, split_planes_kernel), W' once per weight version; fp32 accumulation over all 3 K products, fp32 epilogue.
// a: M, N, K = the real contraction length; A = planes, W = W'; out: fp32 [M][N] (out_mode 0, + fp32 resid / gate), [hi | lo]
// planes [M][2 N] (out_mode 1), or head-split hi / lo planes (heads > 0, lo_off).
int launch_gemm_x3(const GemmArgs& a, int out_mode, hipStream_t s) {
    if (a.M < GM_TM || a.N < 256 || (a.N % 16) || (a.K % GM_KC) || 3 * a.K < 2 * GM_KC || !a.A || !a.W) return (int)hipErrorInvalidValue;
    if ((size_t)a.N * 3 * a.K * 2 >= (1ull << 31)) return (int)hipErrorInvalidValue;
    const int rows_max = (int)(((1ull << 31) - 1) / ((size_t)a.K * 4)) / GM_TM * GM_TM;
    for (int r0 = 0; r0 < a.M; r0 += rows_max) {
        GemmArgs b = a;
        b.row0 = a.row0 + r0;
        b.M = a.M - r0 < rows_max ? a.M - r0 : rows_max;
        if (b.M < GM_TM) return (int)hipErrorInvalidValue;  // (a remainder below one tile: not met by the callers' shapes)
        b.A = reinterpret_cast<const __bf16*>(a.A) + (size_t)r0 * 2 * a.K;
        b.lda = 2 * a.K, b.ldw = 3 * a.K, b.a_wrap = a.K / GM_KC, b.K = 3 * a.K;
        b.ksplit = 1, b.scratch = nullptr;
        if (a.resid) b.resid = reinterpret_cast<const float*>(a.resid) + (size_t)r0 * a.N;
        int rc;
        if (a.heads > 0) {
            rc = launch_pp<GM_EPI_HEADS32>(b, s, false);
        } else if (out_mode == 1) {
            b.out = reinterpret_cast<__bf16*>(a.out) + (size_t)r0 * 2 * a.N;
            rc = launch_pp<GM_EPI_SPLIT>(b, s, false);
        } else {
            b.out = reinterpret_cast<float*>(a.out) + (size_t)r0 * a.N;
            rc = launch_pp<GM_EPI_TOK32>(b, s, false);
        }
        if (rc) return rc;
    }
    return 0;
}
int launch_split3_weights(const float* w, void* out, int N, int K, hipStream_t s) {
    const int64_t n = (int64_t)N * K;
    hipLaunchKernelGGL(split3_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, (__bf16*)out, N, K);
    return (int)hipGetLastError();
}
int launch_split_planes(const float* x, void* out, int64_t M, int K, hipStream_t s) {
    const int64_t n = M * (K / 4);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, (__bf16*)out, M, K);
    return (int)hipGetLastError();
}
