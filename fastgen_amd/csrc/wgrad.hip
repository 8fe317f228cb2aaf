// Weight gradient of the U-Net's convolutions (first kernel of SURVEY §8(f)1, the DMD2 training step):
//
//   dW[co][ci][ky][kx] = sum over (n, y, x) of  dY[n, y, x, co] * A[n, y + ky - pad, x + kx - pad, ci]
//
// i.e. what autograd computes for `Conv2d.forward` (fastgen/networks/EDM/network.py:93-126) in the student / fake-score
// updates (fastgen/methods/distribution_matching/dmd2.py).  A is the conv's input operand (after GroupNorm + SiLU and any
// resampling), dY the gradient of its output; both NHWC bf16, zero padding outside the image.  dW is fp32 in the
// parameter's own OIHW layout.
//
// This is a GEMM whose contraction index is the PIXEL: M = co, N = ci (x taps), K = B*H*W.  Both operands are stored
// pixel-major, so an MFMA lane, which needs 8 consecutive K values of one row, would have to gather 2-byte elements 2*C
// bytes apart.  gfx950's transposing LDS read does that for free: tiles are parked in LDS exactly as they lie in memory
// ([pixel][channel], 16-byte copies) and `ds_read_b64_tr_b16` hands every lane four consecutive pixels of its channel
// (guide T10: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3; lane i receives column i of the 4 rows).
//
// Work split: one 256-thread workgroup owns the output block [128 co] x [32 ci] x [taps] (wave w: co 32w..32w+31, nine
// 32x32 accumulators = 144 registers) and walks a contiguous range of 64-pixel tiles (4 rows x 16 pixels; one 8x8 image);
// grid = (Cin/32) x KSPLIT x (Cout/128).  Each split writes its partial sums to a workspace and a second kernel adds the
// KSPLIT partials in a fixed order — deterministic, no atomics.  Two workgroups fit a CU, so one's loads overlap the
// other's MFMAs; the tile loop itself is the plain load -> barrier -> MFMA -> barrier form (first correct version).
#include <type_traits>

#include "common.h"
#include "misc.h"

namespace {

constexpr int WG_THREADS = 256;
constexpr int WG_CO = 128;       // output channels per workgroup (4 waves x 32)
constexpr int WG_CI = 32;        // input channels per workgroup, 3x3 kernel (nine 32x32 accumulators per wave)
constexpr int WG_CI1 = 128;      // ... 1x1 kernel: one tap, so four 32-channel tiles share every staged dY tile (4x the MFMAs per byte)
__host__ __device__ constexpr int wg_ci(int ks) { return ks == 1 ? WG_CI1 : WG_CI; }
constexpr int WG_TILE = 64;      // pixels per tile = 4 MFMA k-steps of 16 (128 = 8 rows for the 3x3 kernel at width >= 16)
constexpr int WG_DYP = 2 * WG_CO + 64;  // dY tile row pitch in bytes: 4 consecutive rows tile the 64 banks (320 = 64 mod 256)

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ s16x4 tr_read(const char* lds_addr) {
    // generic -> LDS address space: the low 32 bits of a generic LDS pointer are the LDS offset
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(uint32_t)(uintptr_t)lds_addr);
}

// One MFMA operand (8 consecutive K of this lane's row/column) = two transposed reads of 4 pixels each.
__device__ __forceinline__ Frag8<__bf16> tr_frag(const char* p0, const char* p1) {
    const s16x4 lo = tr_read(p0), hi = tr_read(p1);
    s16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = lo[j], v[4 + j] = hi[j];
    Frag8<__bf16> f;
    f.v = __builtin_bit_cast(bf16x8, v);
    return f;
}

// Operand pair of one MFMA side: bf16 mode = one fragment; split-bf16 mode (X3) = the hi and lo planes of the fp32 tensor.
template <bool X3>
struct WFrag {
    Frag8<__bf16> hi, lo;
};
template <bool X3>
__device__ __forceinline__ void wg_mma(f32x16& acc, const WFrag<X3>& a, const WFrag<X3>& b) {
    if constexpr (X3) {
        mma16(acc, a.lo, b.hi);
        mma16(acc, a.hi, b.lo);
    }
    mma16(acc, a.hi, b.hi);
}

// LOGW: log2 of the image width (5, 4, 3).  KS: 3 or 1.  X3: act / dy are fp32 (the split-bf16 compute mode): every 8-channel
// piece is split into hi = bf16(x), lo = bf16(x - hi) on its way into LDS (two images of each tile), and a product is
// dy_lo*a_hi + dy_hi*a_lo + dy_hi*a_hi (common.h bf16x3).  Its LDS (2 x the bf16 tiles) is dynamic: one workgroup per CU for
// the 3x3 kernel at width >= 16, two otherwise.
template <int LOGW, int KS, bool X3>
__global__ __launch_bounds__(WG_THREADS, X3 ? 1 : 2) void conv_wgrad_kernel(const void* __restrict__ act_v, const void* __restrict__ dy_v,
                                                                            float* __restrict__ partial, int B, int Cin, int Cout,
                                                                            int tiles_per_split) {
    typedef typename std::conditional<X3, float, __bf16>::type ST;
    const ST* __restrict__ act = reinterpret_cast<const ST*>(act_v);
    const ST* __restrict__ dy = reinterpret_cast<const ST*>(dy_v);
    constexpr int W = 1 << LOGW;
    constexpr int TW = (W >= 16) ? 16 : 8;    // tile width; a k-step is 16 pixels = one tile row (two rows at 8x8)
    constexpr bool ROLL = (KS == 3 && TW == 16);  // 8-row tiles with a rolling window of halo-row fragments
    constexpr int TPX = ROLL ? 128 : WG_TILE;     // pixels per tile
    constexpr int TH = TPX / TW;                  // 8, 4 (1x1 at width >= 16) or 8 (8x8 images)
    constexpr int PAD = KS / 2;
    constexpr int HW_ = TW + 2 * PAD, HH_ = TH + 2 * PAD;
    constexpr int HALO = HW_ * HH_;           // 108 (3x3 at W >= 16), 100 (3x3 at 8x8), 64 (1x1)
    constexpr int TAPS = KS * KS;
    constexpr int TCOLS = W / TW, TPI = (W / TH) * TCOLS;
    constexpr int CI = wg_ci(KS), NCI = CI / 32;  // input channels per workgroup, 32-channel accumulator tiles per tap
    constexpr int WG_AP = CI == 32 ? 64 : 2 * CI + 64;  // A tile bytes per (halo) pixel: 4 consecutive pixels tile the 64 banks

    constexpr int SZ_DY = TPX * WG_DYP, SZ_A = HALO * WG_AP;
    extern __shared__ __attribute__((aligned(16))) char wg_smem[];  // [dy hi][a hi] ([dy lo][a lo])
    char* const s_dy = wg_smem;
    char* const s_a = wg_smem + SZ_DY;
    constexpr int LO = SZ_DY + ((SZ_A + 255) & ~255);  // byte distance from a hi image to its lo image

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroup L (dispatch order, x fastest) runs on XCD L % 8, each with its own L2.  All (ci block, co block) workgroups of one
    // split read the same pixels — dY is shared by every ci block, the activations by every co block — so a split's workgroups
    // are placed next to each other on ONE XCD (measured before: 684 MB fetched per launch for 160 MB of operands, the dY tile
    // arriving once per XCD).  v enumerates XCD 0's workgroups first, then XCD 1's ...: a bijection for any grid size.
    const int gx = (int)gridDim.x, nb = gx * (int)gridDim.z, T = nb * (int)gridDim.y;
    const int L = (int)blockIdx.x + gx * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z);
    const int v = (L & 7) * (T >> 3) + min(L & 7, T & 7) + (L >> 3);
    const int split = v / nb, blk = v - split * nb;
    const int ci0 = (blk % gx) * CI, cob = (blk / gx) * WG_CO;
    const int ntiles = B * TPI;
    const int t_begin = split * tiles_per_split, t_end = min(ntiles, t_begin + tiles_per_split);

    // lane roles of the transposed reads: group g = lane >> 4, row q, column quad p
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    // dY^T operand (M = co): k = 16 s + 8 (g >> 1) + 4 r + q, co = 32 wave + 16 (g & 1) + 4 p ...
    const char* const dy_lane = s_dy + (8 * (g >> 1) + q) * WG_DYP + (32 * wave + 16 * (g & 1) + 4 * p) * 2;
    // A operand (N = ci): halo pixel of k plus the tap shift; ci = 16 (g & 1) + 4 p ...
    const int a_pix = (TW == 16) ? 8 * (g >> 1) + q : (g >> 1) * HW_ + q;
    const char* const a_lane = s_a + a_pix * WG_AP + (16 * (g & 1) + 4 * p) * 2;

    // one operand of an MFMA from its LDS image(s): the hi image at p0 / p1, the lo image LO bytes behind it
    auto frag2 = [&](const char* p0, const char* p1) __attribute__((always_inline)) -> WFrag<X3> {
        WFrag<X3> f;
        f.hi = tr_frag(p0, p1);
        if constexpr (X3) f.lo = tr_frag(p0 + LO, p1 + LO);
        return f;
    };
    f32x16 acc[TAPS * NCI];
#pragma unroll
    for (int t = 0; t < TAPS * NCI; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // ---- staging through registers, one tile ahead: the global loads of tile t + 1 are issued before tile t's MFMAs and parked in
    // LDS behind them (the split-bf16 mode runs ONE workgroup per CU: without this its load -> barrier -> MFMA phases ran
    // back to back with the matrix pipe idle during every load, profiles/r01_wgrad_pmc.txt: 59 % of the cycles in waits)
    constexpr int NDY = (TPX * 16) / WG_THREADS;                          // 8-channel pieces of the dY tile per thread
    constexpr int NA = (HALO * (CI / 8) + WG_THREADS - 1) / WG_THREADS;   // ... of the activation halo (the last one ragged)
    Frag8<ST> rdy[NDY], ra[NA];
    bool ain[NA];
    auto issue = [&](int t) __attribute__((always_inline)) {
        const int n = t / TPI, slot = t - n * TPI;
        const int row0 = (slot / TCOLS) * TH, col0 = (slot % TCOLS) * TW;
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int c = tid + WG_THREADS * i;
            const int k = c >> 4, ch = c & 15;  // 16 chunks of 8 channels per pixel
            const int y = row0 + k / TW, x = col0 + k % TW;
            rdy[i] = load_frag(dy + ((size_t)(n * W + y) * W + x) * Cout + cob + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = min(tid + WG_THREADS * i, HALO * (CI / 8) - 1);
            const int hp = c / (CI / 8), ch = c % (CI / 8);
            const int y = row0 + hp / HW_ - PAD, x = col0 + hp % HW_ - PAD;
            ain[i] = y >= 0 && y < W && x >= 0 && x < W;
            ra[i] = load_frag(act + ((size_t)(n * W + (ain[i] ? y : 0)) * W + (ain[i] ? x : 0)) * Cin + ci0 + ch * 8);
        }
    };
    auto park = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int c = tid + WG_THREADS * i;
            const int k = c >> 4, ch = c & 15;
            if constexpr (X3) {
                float v[8];
                widen8(rdy[i], v);
                bf16x8 hi, lo;
                split8(v, hi, lo);
                *reinterpret_cast<bf16x8*>(s_dy + k * WG_DYP + ch * 16) = hi;
                *reinterpret_cast<bf16x8*>(s_dy + LO + k * WG_DYP + ch * 16) = lo;
            } else {
                *reinterpret_cast<bf16x8*>(s_dy + k * WG_DYP + ch * 16) = rdy[i].v;
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + WG_THREADS * i;
            if (c < HALO * (CI / 8)) {
                const int hp = c / (CI / 8), ch = c % (CI / 8);
                if constexpr (X3) {
                    float v[8];
                    widen8(ra[i], v);
                    bf16x8 hi, lo;
                    split8(v, hi, lo);
                    if (!ain[i]) hi = lo = bf16x8{};
                    *reinterpret_cast<bf16x8*>(s_a + hp * WG_AP + ch * 16) = hi;
                    *reinterpret_cast<bf16x8*>(s_a + LO + hp * WG_AP + ch * 16) = lo;
                } else {
                    *reinterpret_cast<bf16x8*>(s_a + hp * WG_AP + ch * 16) = ain[i] ? ra[i].v : bf16x8{};
                }
            }
        }
    };
    // (the bf16 3x3 kernel runs two workgroups per CU inside a 256-register budget: no room for a tile in flight - and its second
    // workgroup already covers the loads)
    constexpr bool PREF = X3 || KS == 1;
    if (PREF && t_begin < t_end) issue(t_begin);
    // the unpipelined form: every piece goes global -> LDS on its own (no tile's worth of registers held)
    auto stage_direct = [&](int t) __attribute__((always_inline)) {
        const int n = t / TPI, slot = t - n * TPI;
        const int row0 = (slot / TCOLS) * TH, col0 = (slot % TCOLS) * TW;
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int c = tid + WG_THREADS * i;
            const int k = c >> 4, ch = c & 15;
            const int y = row0 + k / TW, x = col0 + k % TW;
            *reinterpret_cast<uint4*>(s_dy + k * WG_DYP + ch * 16) =
                *reinterpret_cast<const uint4*>(dy + ((size_t)(n * W + y) * W + x) * Cout + cob + ch * 8);
        }
        for (int c = tid; c < HALO * (CI / 8); c += WG_THREADS) {
            const int hp = c / (CI / 8), ch = c % (CI / 8);
            const int y = row0 + hp / HW_ - PAD, x = col0 + hp % HW_ - PAD;
            const bool in = y >= 0 && y < W && x >= 0 && x < W;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (in) v = *reinterpret_cast<const uint4*>(act + ((size_t)(n * W + y) * W + x) * Cin + ci0 + ch * 8);
            *reinterpret_cast<uint4*>(s_a + hp * WG_AP + ch * 16) = v;
        }
    };
    for (int t = t_begin; t < t_end; ++t) {
        if constexpr (PREF) park();
        else stage_direct(t);
        __syncthreads();
        if (PREF && t + 1 < t_end) issue(t + 1);
        // ---- k-steps x taps MFMAs ------------------------------------------------------------------------------------
        if constexpr (ROLL) {
            // Tap (ky, kx) of tile row s reads halo row s + ky shifted by kx: the same fragment serves (s, ky), (s+1, ky-1) and
            // (s+2, ky-2).  A ring of FOUR halo rows x three shifts stays in registers (slot = halo row % 4): step s computes on
            // rows s .. s+2 while row s+3 and the next step's dY fragment are on their way - 8 transposed reads per 9 MFMAs
            // instead of 20, issued a whole step (27 or 9 MFMAs) before their use.  (Inline asm + one hand-placed wait per step:
            // hipcc sank every compiler-visible read to just before its MFMA, an LDS round trip every two or three MFMAs -
            // 54 full waits per 216 MFMAs with one wave per SIMD and nothing else to issue.)
            struct FR {
                s16x8 h, l;  // hi image: pixels k .. k+7 (two transposed reads of 4); lo image (split-bf16 mode only)
            };
            const uint32_t a_u = (uint32_t)(uintptr_t)a_lane, d_u = (uint32_t)(uintptr_t)dy_lane;
            const uint32_t a_ul = a_u + LO, d_ul = d_u + LO;
            FR F[4][3], fa[2];
            // (the two 4-pixel halves of an operand are joined right at the read: the register allocator gives the two reads the halves
            // of one 4-register tuple - joined only in front of the MFMA they cost two to four v_mov per fragment)
            auto rd = [&](FR& f, uint32_t base, uint32_t base_lo, auto OFF0_, auto OFF1_) __attribute__((always_inline)) {
                constexpr int O0 = decltype(OFF0_)::value, O1 = decltype(OFF1_)::value;
                s16x4 p0, p1;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(p0) : "v"(base), "n"(O0));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(p1) : "v"(base), "n"(O1));
                f.h = s16x8{p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
                if constexpr (X3) {
                    s16x4 q0, q1;
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(q0) : "v"(base_lo), "n"(O0));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(q1) : "v"(base_lo), "n"(O1));
                    f.l = s16x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
                }
            };
            auto landed = [&](FR& f) __attribute__((always_inline)) {  // (ties the fragment to the wait that precedes this call)
                asm volatile("" : "+v"(f.h));
                if constexpr (X3) asm volatile("" : "+v"(f.l));
            };
            auto as_frag = [&](const FR& f) __attribute__((always_inline)) -> WFrag<X3> {
                WFrag<X3> w;
                w.hi.v = __builtin_bit_cast(bf16x8, f.h);
                if constexpr (X3) w.lo.v = __builtin_bit_cast(bf16x8, f.l);
                return w;
            };
            auto load_row = [&](auto HR_) __attribute__((always_inline)) {
                constexpr int hr = decltype(HR_)::value;
                rd(F[hr % 4][0], a_u, a_ul, std::integral_constant<int, (hr * HW_ + 0) * WG_AP>{}, std::integral_constant<int, (hr * HW_ + 4) * WG_AP>{});
                rd(F[hr % 4][1], a_u, a_ul, std::integral_constant<int, (hr * HW_ + 1) * WG_AP>{}, std::integral_constant<int, (hr * HW_ + 5) * WG_AP>{});
                rd(F[hr % 4][2], a_u, a_ul, std::integral_constant<int, (hr * HW_ + 2) * WG_AP>{}, std::integral_constant<int, (hr * HW_ + 6) * WG_AP>{});
            };
            auto load_dy = [&](auto S_) __attribute__((always_inline)) {
                constexpr int s_ = decltype(S_)::value;
                rd(fa[s_ & 1], d_u, d_ul, std::integral_constant<int, (16 * s_) * WG_DYP>{}, std::integral_constant<int, (16 * s_ + 4) * WG_DYP>{});
            };
            static_assert((9 * HW_ + 6) * WG_AP < 65536 && (16 * 7 + 4) * WG_DYP < 65536, "ds_read offsets are 16-bit");
            load_row(std::integral_constant<int, 0>{});
            load_row(std::integral_constant<int, 1>{});
            load_row(std::integral_constant<int, 2>{});
            load_dy(std::integral_constant<int, 0>{});
            auto step = [&](auto S_) __attribute__((always_inline)) {
                constexpr int s_ = decltype(S_)::value;
                // what this step computes on was issued a step ago (or in the prologue): one wait, covered by the last step's MFMAs
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) landed(F[(s_ + r3) % 4][kx]);
                landed(fa[s_ & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s_ + 1 < TH) {
                    load_row(std::integral_constant<int, s_ + 3>{});
                    load_dy(std::integral_constant<int, s_ + 1>{});
                }
                const WFrag<X3> fd = as_frag(fa[s_ & 1]);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) wg_mma<X3>(acc[tap], fd, as_frag(F[(s_ + tap / 3) % 4][tap % 3]));
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            if constexpr (TH > 4) {
                step(std::integral_constant<int, 4>{});
                step(std::integral_constant<int, 5>{});
                step(std::integral_constant<int, 6>{});
                step(std::integral_constant<int, 7>{});
            }
        } else
#pragma unroll
        for (int s = 0; s < TPX / 16; ++s) {
            const WFrag<X3> fa = frag2(dy_lane + (16 * s) * WG_DYP, dy_lane + (16 * s + 4) * WG_DYP);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int ky = tap / KS, kx = tap % KS;
                // first halo pixel of the k-step for this tap (compile-time), rows of the tile are HW_ halo pixels apart
                const int hp0 = (TW == 16) ? (s + ky) * HW_ + kx : (2 * s + ky) * HW_ + kx;
#pragma unroll
                for (int nt = 0; nt < NCI; ++nt) {
                    const WFrag<X3> fb = frag2(a_lane + hp0 * WG_AP + nt * 64, a_lane + (hp0 + 4) * WG_AP + nt * 64);
                    wg_mma<X3>(acc[tap * NCI + nt], fa, fb);
                }
            }
        }
        __syncthreads();
    }
    // ---- partial[split][tap][co][ci] (ci contiguous: a lane group writes 128 consecutive bytes): lane holds column
    // ci = lane & 31, rows co = acc_row(i, lane >> 5) ---------------------------------------------------------------------
    float* dst = partial + (size_t)split * Cout * Cin * TAPS;
    const int ci = ci0 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int nt = 0; nt < NCI; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = cob + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
                dst[((size_t)tap * Cout + co) * Cin + ci + 32 * nt] = acc[tap * NCI + nt][i];
            }
}

// dW[co][ci][tap] (+)= sum over splits of partial[split][tap][co][ci], fixed order; thread i walks the partial layout
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, size_t n,
                                                           int splits, int accumulate, int CC, int taps, float scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += partial[(size_t)k * n + i];
    s *= scale;
    const size_t tap = i / CC, cc = i - tap * CC;  // cc = co * Cin + ci
    float* o = dw + cc * taps + tap;
    *o = accumulate ? *o + s : s;
}

// tiles are 8 x 16 pixels for the 3x3 kernel at width >= 16, 4 x 16 for the 1x1 kernel, one image at 8x8
int tiles_per_image(int res, int ks) { return res >= 16 ? (res / (ks == 3 ? 8 : 4)) * (res / 16) : 1; }

}  // namespace

int conv_wgrad_splits(int B, int res, int cin, int cout, int ks) {
    const int blocks = (cin / wg_ci(ks)) * (cout / WG_CO);
    const int ntiles = B * tiles_per_image(res, ks);
    int splits = 512 / blocks;  // at most two workgroups per CU in ONE round (528 workgroups would run a second, almost empty round) ...
    if (splits > ntiles / 8) splits = ntiles / 8;  // ... but at least 8 tiles per split: the partials cost HBM traffic
    if (splits < 1) splits = 1;
    return splits;
}

size_t conv_wgrad_workspace_bytes(int B, int res, int cin, int cout, int ks) {
    return (size_t)conv_wgrad_splits(B, res, cin, cout, ks) * cout * cin * ks * ks * sizeof(float);
}

int conv_wgrad_supported(int res, int cin, int cout, int ks) {
    return (res == 8 || res == 16 || res == 32) && cin > 0 && (ks == 1 || ks == 3) && cin % wg_ci(ks) == 0 && cout > 0 && cout % WG_CO == 0;
}

// LDS bytes of one instantiation (mirrors the kernel's constants)
template <int LOGW, int KS, bool X3>
constexpr int wgrad_lds_bytes() {
    constexpr int W = 1 << LOGW, TW = (W >= 16) ? 16 : 8;
    constexpr bool ROLL = (KS == 3 && TW == 16);
    constexpr int TPX = ROLL ? 128 : WG_TILE, TH = TPX / TW, PAD = KS / 2;
    constexpr int HALO = (TW + 2 * PAD) * (TH + 2 * PAD);
    constexpr int CI = wg_ci(KS);
    constexpr int AP = CI == 32 ? 64 : 2 * CI + 64;
    constexpr int SZ_DY = TPX * WG_DYP, SZ_A = HALO * AP;
    return X3 ? 2 * (SZ_DY + ((SZ_A + 255) & ~255)) : SZ_DY + SZ_A;
}
template <int LOGW, int KS, bool X3>
int wgrad_launch(dim3 grid, hipStream_t s, const void* a, const void* d, float* part, int B, int cin, int cout, int tps) {
    auto kern = conv_wgrad_kernel<LOGW, KS, X3>;
    constexpr int lds = wgrad_lds_bytes<LOGW, KS, X3>();
    if (lds > 64 * 1024) {  // above the default cap: raise it once per device
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
        static bool done[16] = {};
        if (!done[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return (int)e;
            done[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(WG_THREADS), lds, s, a, d, part, B, cin, cout, tps);
    return (int)hipGetLastError();
}

// mode: FG_DTYPE_BF16 (1): act / dy bf16; FG_DTYPE_BF16X3 (2): act / dy fp32, split-bf16 products
int launch_conv_wgrad(int mode, const void* act, const void* dy, float* dw, int B, int res, int cin, int cout, int ks, int accumulate,
                      void* workspace, hipStream_t s, float scale) {
    if (!conv_wgrad_supported(res, cin, cout, ks) || B <= 0 || (mode != 1 && mode != 2)) return (int)hipErrorInvalidValue;
    const int splits = conv_wgrad_splits(B, res, cin, cout, ks);
    const int ntiles = B * tiles_per_image(res, ks);
    const int tps = (ntiles + splits - 1) / splits;
    dim3 grid(cin / wg_ci(ks), splits, cout / WG_CO);
    float* part = (float*)workspace;
    int rc;
#define WG_LAUNCH(LW, KS) \
    rc = (mode == 2) ? wgrad_launch<LW, KS, true>(grid, s, act, dy, part, B, cin, cout, tps) : wgrad_launch<LW, KS, false>(grid, s, act, dy, part, B, cin, cout, tps)
    if (ks == 3) {
        if (res == 32) WG_LAUNCH(5, 3);
        else if (res == 16) WG_LAUNCH(4, 3);
        else WG_LAUNCH(3, 3);
    } else {
        if (res == 32) WG_LAUNCH(5, 1);
        else if (res == 16) WG_LAUNCH(4, 1);
        else WG_LAUNCH(3, 1);
    }
#undef WG_LAUNCH
    if (rc) return rc;
    const size_t n = (size_t)cout * cin * ks * ks;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, dw, n, splits, accumulate,
                       cout * cin, ks * ks, scale);
    return (int)hipGetLastError();
}
