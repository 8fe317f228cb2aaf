// Fused GroupNorm-apply + SiLU -> Conv2d(3x3 | 1x1) -> bias (+temb) (+residual) * scale (+ GroupNorm partial statistics
// of the result), as an implicit GEMM on the CDNA4 matrix cores.  Replaces, per UNetBlock (reference
// fastgen/networks/EDM/network.py:274-299):
//     conv0(silu(norm0(x))) + affine(emb)        -> PRO_GN_SILU, temb epilogue
//     (conv1(silu(norm1(h))) + skip) * sqrt(.5)  -> PRO_GN_SILU, residual epilogue
//     skip / qkv / proj 1x1 convs                -> KS = 1
// and the resampling of Conv2d.forward (:114-121) folded into the operand load (RES_DOWN = 2x2 mean of the
// *transformed* input, RES_UP = nearest replication), and torch.cat (:560) as two source pointers.
//
// Tiling — one workgroup = 256 threads = 4 waves; TWO workgroups are resident per CU (<= 66 KB LDS, 256 VGPRs), so one
// workgroup's memory-bound prologue/epilogue overlaps the other's MFMA main loop:
//   M = 128 output pixels (4 rows x 32 | 8 x 16 | 2 images x 8 x 8), N = 256 output channels, K = taps x Cin.
//   wave w owns output channels [64w, 64w+64) for all 128 pixels: 4 x 2 accumulator tiles of 32x32 (128 VGPRs).
//   A (activations): per K-chunk of KC input channels the (rows+2) x (W+2) halo of the pixel tile is transformed
//     ONCE (GN affine + SiLU, cast to the compute dtype) and parked in LDS; all 9 taps read shifted windows of it.
//     144-byte pixel pitch (+ a per-row pad at W = 16 / 8) keeps every ds_read_b128 of a fragment conflict-free.
//     Double-buffered: chunk c+1 is staged while chunk c is multiplied, global loads issued before the MFMA block and
//     consumed after it.  One barrier per chunk.
//   B (weights): pre-packed in MFMA fragment order, read straight from global/L2 into registers and refilled in place
//     one (chunk, tap) step ahead — each wave reads only its own 64 output channels: no LDS, no barrier for B.
//   Epilogue: besides the store, each wave reduces sum / sum-of-squares of its outputs per (image, 4-channel quad) and
//     writes them to a small side buffer, so the NEXT GroupNorm never re-reads the tensor (misc.hip gn_finalize).
#include "common.h"
#include "conv.h"
#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int PITCH = 144;  // bytes per halo pixel in LDS: 128 B of channels + 16 B pad (odd multiple of 16 B)
constexpr int NTHR = 256;

template <int KS, int LOGW>
struct Geom {
    static constexpr int W = 1 << LOGW;              // image width (= height)
    // Pixel tile of 128 outputs: 8 rows x 16 columns at 32x32 and 16x16 (halo 10 x 18 = 1.41x the tile; the former 4 x 32
    // strip had 1.59x), two whole images at 8x8.
    static constexpr int LOGTW = (LOGW == 5) ? 4 : LOGW;
    static constexpr int TW = 1 << LOGTW;            // tile columns
    static constexpr int LOGTH = 3;
    static constexpr int TH = 1 << LOGTH;            // tile rows per image
    static constexpr int IMGS = 128 / (TH * TW);     // images per tile (2 at 8x8, else 1)
    static constexpr int TCOLS = W / TW;             // tiles across the image width
    static constexpr int TPI = (IMGS > 1) ? 1 : (W / TH) * TCOLS;  // tiles (= statistics slots) per image
    static constexpr int PAD = KS / 2;
    static constexpr int HW_ = TW + 2 * PAD;
    static constexpr int HH_ = TH + 2 * PAD;
    static constexpr int HALO_PIX = IMGS * HH_ * HW_;
    static constexpr int TAPS = KS * KS;
    // LDS row stride.  A 32-pixel MFMA row tile spans 2 / 4 halo rows at tile width 16 / 8; the row stride in 16-byte slots
    // must be 0 (width 16) or 8 (width 8) mod 16 for the 16-lane ds_read_b128 groups to hit 16 distinct slots.
    static constexpr int ROWPAD = 16 * ((((LOGTW == 4) ? 0 : 8) - (HW_ * 9) % 16 + 32) % 16);
    static constexpr int RS = HW_ * PITCH + ROWPAD;
    static constexpr int ABUF = IMGS * HH_ * RS;
    // LDS byte offset (tap 0,0) of tile pixel p in [0,128); additive in (p & ~31) and (p & 31)
    static __host__ __device__ constexpr int off0(int p) {
        return (((p >> (LOGTW + LOGTH)) * HH_) + ((p >> LOGTW) & (TH - 1))) * RS + (p & (TW - 1)) * PITCH;
    }
};

// Output-channel tiles per wave.  The 8x8 layers have only B/2 pixel tiles: splitting N over two workgroups (128
// channels each) doubles the grid so that two workgroups are resident per CU there as well.
__host__ __device__ constexpr int conv_nt(int ks, int logw, int outmode) {
    return (logw == 3 && ks == 3 && outmode == OUT_NHWC) ? 1 : 2;
}

// GELU with the tanh approximation (torch.nn.GELU(approximate="tanh"), the DiT feed-forward: DiT/network.py:176)
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float u = 0.7978845608028654f * fmaf(0.044715f * x, x * x, x);
    // tanh(u) = 1 - 2 / (1 + e^{2u}); accurate expf keeps the exact-fp32 mode exact, the MFMA-bound GEMM hides its cost
    const float t = 1.0f - 2.0f / (1.0f + expf(2.0f * u));
    return 0.5f * x * (1.0f + t);
}

template <int PRO, bool FAST>
__device__ __forceinline__ float pro_apply(float x, float2 ab) {
    if (PRO == PRO_NONE) return x;
    float y = fmaf(x, ab.x, ab.y);
    if (PRO == PRO_GN_SILU) y = silu_f<FAST>(y);
    return y;
}

__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
}
__device__ __forceinline__ void load8(const __bf16* p, float (&v)[8]) {
    const bf16x8 q = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)q[j];
}

// Activations (sources, residual, output) are stored in the compute dtype T: fp32 in fp32 mode, bf16 in bf16 mode (the
// reference's own bf16 autocast keeps conv outputs — hence the residual stream — in bf16, EDM/network.py:285-287).
// DBG: compile-time switch for the ablation hooks (scripts/conv_ablate.py); production instantiations use DBG = false so
// that no runtime flag splits the MFMA basic block (a join makes the compiler's s_waitcnt placement conservative).
template <typename T, int KS, int PRO, int RES, int LOGW, int OUTMODE, int ABL = 0>
__global__ __launch_bounds__(NTHR, 2) void conv_fused_kernel(const ConvArgs a) {
    constexpr bool DBG = false;  // (runtime ablation flags retired: they perturbed the code they measured)
    using G = Geom<KS, LOGW>;
    using ST = typename DT<T>::ST;  // storage type of the activation tensors
    using WT = typename DT<T>::WT;  // element type of the packed weights
    constexpr int FB = DT<T>::FRAG_BYTES;
    constexpr int WP = DT<T>::WPARTS;
    constexpr bool SAME = (sizeof(ST) == FB / 8);  // LDS holds the storage type itself (no split, no conversion)
    constexpr int dbg = ABL;  // compile-time ablation mask: 1 no staging, 2 no weight refill, 4 no epilogue, 8 no MFMA
    constexpr int KC = DT<T>::KC;
    constexpr bool FAST = DT<T>::FAST;
    constexpr int KK = KC / 16;   // 16-deep MFMA steps per chunk
    constexpr int OPP = KC / 8;   // 8-channel octets per pixel and chunk
    constexpr int LOG_OPP = (OPP == 8) ? 3 : 2;
    constexpr int PSTRIDE = NTHR / OPP;                               // halo pixels covered per staging item
    constexpr int NITEMS = (G::HALO_PIX * OPP + NTHR - 1) / NTHR;     // staging items per thread and chunk
    constexpr int IPS = (NITEMS + G::TAPS - 1) / G::TAPS;             // items staged per (chunk, tap) step
    constexpr int ND = (RES == RES_DOWN) ? 4 : 1;                     // source pixels per staged pixel (2x2 mean when down-sampling)
    // split load / transform+write around the MFMAs; with down-sampling an item holds 4 raw fragments: bf16 3x3 only
    constexpr bool DEFER = (RES != RES_DOWN || (KS == 3 && sizeof(ST) == 2));
    constexpr bool AB_REGS = (PRO != PRO_NONE) && (G::IMGS == 1);     // per-chunk GN coefficients live in registers
    constexpr bool PIPE_A = (sizeof(ST) == 2);                        // two A-fragment register sets (bf16 only; bf16x3 holds hi+lo)
    constexpr int NT = conv_nt(KS, LOGW, OUTMODE);                    // 32-channel tiles per wave: 2, or 1 (N split over 2 workgroups)

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, nblk = blockIdx.y;
    const int Cin = a.C1 + a.C2;
    const int nchunk = Cin / KC;
    const int nsteps = nchunk * G::TAPS;
    const int H = a.H;  // == W == G::W

    const int n_base = (G::IMGS > 1) ? tile * G::IMGS : tile / G::TPI;
    const int slot = (G::IMGS > 1) ? 0 : tile % G::TPI;
    const int row0 = (slot / G::TCOLS) * G::TH, col0 = (slot % G::TCOLS) * G::TW;

    // this wave's packed weights: [cout/32][step][kk][lane][8], two consecutive 32-channel groups
    const size_t wstride = (size_t)nsteps * (KK * 512 * WP);
    // weights through a buffer resource: SGPR offsets, no vector address arithmetic (common.h load_frag_rsrc)
    // Token GEMMs (OUT_TOK / OUT_HEADS) take any Cout % 64 == 0: in the last 256-column tile the waves past Cout re-do the last valid
    // 64 columns (loads stay inside the packed weights) and skip the epilogue.
    constexpr bool TOKM = (OUTMODE == OUT_TOK || OUTMODE == OUT_HEADS);
    const int wcol0 = nblk * (128 * NT) + wave * (32 * NT);  // first output column of this wave
    const bool wactive = !TOKM || wcol0 < a.Cout;
    const int wgroup = wactive ? nblk * (4 * NT) + wave * NT : a.Cout / 32 - NT;
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(reinterpret_cast<const WT*>(a.wpack) + (size_t)wgroup * wstride);
    const int wlane = lane * 8;
    const int wst = (int)wstride;

    const ST* src1 = reinterpret_cast<const ST*>(a.src1);
    const ST* src2 = reinterpret_cast<const ST*>(a.src2);

    // staging role of this thread: fixed channel octet, halo pixels hq0 + i*PSTRIDE
    const int oct = tid & (OPP - 1);
    const int hq0 = tid >> LOG_OPP;

    // ---- staging helpers ------------------------------------------------------------------------------
    auto decode = [&](int hq, int& n, int& y, int& x, int& lds_off) -> bool {
        const int hx = hq % G::HW_;
        const int t = hq / G::HW_;
        const int hy = t % G::HH_;
        const int img = t / G::HH_;
        y = row0 + hy - G::PAD;
        x = col0 + hx - G::PAD;
        n = n_base + img;
        lds_off = (img * G::HH_ + hy) * G::RS + hx * PITCH + oct * FB;
        return (hq < G::HALO_PIX) && (y >= 0) && (y < H) && (x >= 0) && (x < G::W) && (n < a.B);
    };
    auto src_ptr = [&](int chunk, int n, int sy, int sx) -> const ST* {
        const int c0 = chunk * KC + oct * 8;
        const size_t sp = ((size_t)n * a.Hs + sy) * a.Ws + sx;
        return (c0 < a.C1) ? src1 + sp * a.C1 + c0 : src2 + sp * a.C2 + (c0 - a.C1);
    };
    auto load_ab = [&](int chunk, int n, float2 (&ab)[8]) {
        const float2* p = a.ab + (size_t)n * Cin + chunk * KC + oct * 8;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p + j);
            ab[j] = make_float2(q[0], q[1]);
            ab[j + 1] = make_float2(q[2], q[3]);
        }
    };

    float2 abr[8];  // AB_REGS: coefficients of the chunk currently being staged
#pragma unroll
    for (int j = 0; j < 8; ++j) abr[j] = make_float2(1.f, 0.f);

    // phase 1 of an item: issue the global loads (3x3, RES_NONE / RES_UP only)
    auto item_load = [&](int chunk, int i, Frag8<ST> (&raw)[ND], bool& valid) {
        int n, y, x, lo;
        valid = decode(hq0 + i * PSTRIDE, n, y, x, lo);
#pragma unroll
        for (int d = 0; d < ND; ++d) raw[d] = Frag8<ST>{};
        if (valid) {
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int sy = (RES == RES_DOWN) ? 2 * y + (d >> 1) : ((RES == RES_UP) ? (y >> 1) : y);
                const int sx = (RES == RES_DOWN) ? 2 * x + (d & 1) : ((RES == RES_UP) ? (x >> 1) : x);
                raw[d] = load_frag(src_ptr(chunk, n, sy, sx));
            }
        }
    };
    // phase 2: transform and park in LDS
    auto item_finish = [&](int chunk, int i, char* abuf, const Frag8<ST> (&rawp)[ND], bool valid) {
        const int hq = hq0 + i * PSTRIDE;
        if (hq >= G::HALO_PIX) return;
        int n, y, x, lo;
        const bool ok = decode(hq, n, y, x, lo);
        if constexpr (DEFER && PRO == PRO_NONE && ND == 1 && SAME) {
            // nothing to transform: park the loaded fragment as it is (zeros outside the image) — no widening, no
            // re-rounding, no vector arithmetic beyond the select
            Frag8<ST> z = rawp[0];
            if (!valid) z = Frag8<ST>{};
            *reinterpret_cast<Frag8<ST>*>(abuf + lo) = z;
            return;
        }
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
        if (DEFER) {
            if (valid) {
                float2 ab[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) ab[j] = abr[j];
                if (PRO != PRO_NONE && !AB_REGS) load_ab(chunk, n, ab);
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    float raw[8];
                    widen8(rawp[d], raw);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += pro_apply<PRO, FAST>(raw[j], ab[j]);
                }
                if (ND == 4) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] *= 0.25f;
                }
            }
        } else if (ok) {  // synchronous path (qkv, fp32 / 1x1 down-sampling)
            float2 ab[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ab[j] = AB_REGS ? abr[j] : make_float2(1.f, 0.f);
            if (PRO != PRO_NONE && !AB_REGS) load_ab(chunk, n, ab);
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                float v[8];
                const int sy = (RES == RES_DOWN) ? 2 * y + (d >> 1) : ((RES == RES_UP) ? (y >> 1) : y);
                const int sx = (RES == RES_DOWN) ? 2 * x + (d & 1) : ((RES == RES_UP) ? (x >> 1) : x);
                load8(src_ptr(chunk, n, sy, sx), v);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += pro_apply<PRO, FAST>(v[j], ab[j]);
            }
            if (ND == 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] *= 0.25f;
            }
        }
        lds_store_a<T>(abuf + lo, o);
    };

    // ---- prologue: chunk 0 of A, step 0 of B ------------------------------------------------------
    if (AB_REGS) load_ab(0, n_base, abr);
    if (DEFER && ND == 1) {
        // all loads of chunk 0 first, branch-free (out-of-image pixels read a clamped address and are zeroed by `valid`),
        // then the transforms: one global-load latency per workgroup instead of one per item
        Frag8<ST> raw0[NITEMS][ND];
        bool valid0[NITEMS];
#pragma unroll
        for (int i = 0; i < NITEMS; ++i) {
            int n, y, x, lo;
            valid0[i] = decode(min(hq0 + i * PSTRIDE, G::HALO_PIX - 1), n, y, x, lo);
            const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), G::W - 1), nc = min(n, a.B - 1);
            raw0[i][0] = load_frag(src_ptr(0, nc, (RES == RES_UP) ? (yc >> 1) : yc, (RES == RES_UP) ? (xc >> 1) : xc));
        }
#pragma unroll
        for (int i = 0; i < NITEMS; ++i) item_finish(0, i, smem, raw0[i], valid0[i]);
    } else {
#pragma unroll
        for (int i = 0; i < NITEMS; ++i) {
            Frag8<ST> raw[ND] = {};
            bool valid = false;
            if (DEFER) item_load(0, i, raw, valid);
            item_finish(0, i, smem, raw, valid);
        }
    }
    Frag8<T> bfr[NT][KK];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) load_frag_rsrc(bfr[nt][kk], wrs, wlane, nt * wst + kk * (512 * WP));

    f32x16 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int lane_off = G::off0(r) + h * FB;
    __syncthreads();

    // ---- main loop ------------------------------------------------------------------------------------
    int step = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const char* abuf = smem + (chunk & 1) * G::ABUF;
        char* anext = smem + ((chunk + 1) & 1) * G::ABUF;
        const bool stage_next = (chunk + 1 < nchunk);
#pragma unroll 1
        for (int tap = 0; tap < G::TAPS; ++tap, ++step) {
            // (1) weights: each fragment is refilled in place for the NEXT step right after its last use below
            //     (clamped: the last step re-reads itself), so B needs 2*KK fragments with one step of prefetch distance
            const int pnext = ((step + 1 < nsteps) ? step + 1 : step) * (KK * 512 * WP);
            // (2) issue this step's share of the next chunk's activation loads
            Frag8<ST> raw[IPS][ND];
            bool valid[IPS];
            const bool do_stage = stage_next && (tap * IPS < NITEMS) && !(dbg & 1);
            if (do_stage) {
                if (AB_REGS && tap == 0) load_ab(chunk + 1, n_base, abr);
                if (DEFER) {
#pragma unroll
                    for (int q = 0; q < IPS; ++q) item_load(chunk + 1, tap * IPS + q, raw[q], valid[q]);
                }
            }
            // (3) multiply: 4 pixel tiles x 2 channel tiles x KK k-steps.  The 4 A fragments of a k-step are read from LDS
            //     as one group; with bf16 operands the next k-step's group is issued before this k-step's 8 MFMAs.
            const int tap_off = (tap / KS) * G::RS + (tap % KS) * PITCH;
            const char* abase = abuf + lane_off + tap_off;
            auto read_a = [&](int kk, Frag8<T> (&af)[4]) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    af[mt] = lds_read_a<T>(abase + G::off0(mt * 32), kk);
            };
            auto mma8 = [&](int kk, const Frag8<T> (&af)[4]) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma16(acc[mt][nt], af[mt], bfr[nt][kk]);
                if (!(dbg & 2)) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) load_frag_rsrc(bfr[nt][kk], wrs, wlane, pnext + nt * wst + kk * (512 * WP));
                }
            };
            if (dbg & 8) {
                // ablation: no LDS reads, no MFMAs
            } else if (PIPE_A) {
                Frag8<T> a0[4], a1[4];
                read_a(0, a0);
#pragma unroll
                for (int kk = 0; kk < KK; kk += 2) {
                    read_a(kk + 1, a1);
                    __builtin_amdgcn_sched_barrier(0);
                    mma8(kk, a0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kk + 2 < KK) read_a(kk + 2, a0);
                    __builtin_amdgcn_sched_barrier(0);
                    mma8(kk + 1, a1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    Frag8<T> af[4];
                    read_a(kk, af);
                    __builtin_amdgcn_sched_barrier(0);
                    mma8(kk, af);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // (4) transform + park the staged items
            if (do_stage) {
#pragma unroll
                for (int q = 0; q < IPS; ++q)
                    if (tap * IPS + q < NITEMS) item_finish(chunk + 1, tap * IPS + q, anext, raw[q], valid[q]);
            }
        }
        __syncthreads();
    }

    // ---- epilogue -------------------------------------------------------------------------------------
    const int HWo = H * G::W;
    if (dbg & 4) {  // ablation: keep the accumulators live, store (almost) nothing
        float t = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) t += acc[mt][nt][i];
        if (t == 1234.5678f) reinterpret_cast<float*>(a.out)[0] = t;
        return;
    }
    if (!wactive) return;  // (no workgroup barrier below this point)
    // which plane of a head-split qkv projection this wave's 64 columns lie in (planes are multiples of 64 wide)
    const int hplane = (OUTMODE == OUT_HEADS) ? wcol0 / (a.heads * a.head_dim) : 0;
    if (OUTMODE == OUT_NHWC || OUTMODE == OUT_TOK || (OUTMODE == OUT_QKV && nblk < 2) || (OUTMODE == OUT_HEADS && hplane < 2)) {
        // NHWC tensor, or the q / k plane of a qkv projection ([B][HW][256]; OUT_HEADS: [B][heads][T][head_dim])
        // Transpose each 32-pixel x 64-channel accumulator slab through this wave's private LDS scratch (the A buffers
        // are dead after the last barrier) so that global traffic is row-contiguous: a lane owns one channel QUAD of one
        // pixel (16 B fp32 / 8 B bf16 per access), 16 lanes cover a pixel's 64 channels, 4 pixels per wave instruction.
        constexpr int QPW = 8 * NT;              // channel quads per pixel row of this wave's slab
        constexpr int RP = 64 / QPW;             // pixel rows per wave instruction
        constexpr int NP = 32 / RP;              // passes per 32-pixel tile
        constexpr int EP_PITCH = 32 * NT * 4 + 16;
        constexpr int MT_PER_IMG = 4 / G::IMGS;
        static_assert(4 * 32 * EP_PITCH <= 2 * G::ABUF, "epilogue scratch must fit in the A buffers");
        char* ep = smem + wave * (32 * EP_PITCH);
        const int c4 = lane % QPW, prow = lane / QPW;
        const int co0 = nblk * (128 * NT) + wave * (32 * NT) + c4 * 4;
        const bool qk = (OUTMODE == OUT_QKV);
        const bool hd = (OUTMODE == OUT_HEADS);
        ST* out = reinterpret_cast<ST*>(qk ? (nblk == 0 ? a.q_out : a.k_out) : (hd ? (hplane == 0 ? a.q_out : a.k_out) : a.out));
        const ST* resid = (qk || hd) ? nullptr : reinterpret_cast<const ST*>(a.resid);
        const int ostride = qk ? 256 : (hd ? a.head_dim : a.Cout);  // channels per pixel of the tensor written
        // channel within that tensor (co0 indexes bias / temb / stats); OUT_HEADS: this lane's quad lies in ONE head (head_dim % 4 == 0)
        const int hcol = hd ? co0 - hplane * a.heads * a.head_dim : 0;
        const int hhead = hd ? hcol / a.head_dim : 0;
        const int co_out = qk ? co0 - nblk * 256 : (hd ? hcol - hhead * a.head_dim : co0);
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + co0);
        float ssum[G::IMGS], ssq[G::IMGS];
#pragma unroll
        for (int im = 0; im < G::IMGS; ++im) ssum[im] = ssq[im] = 0.f;
        // global element offset of this lane's quad for pixel row pl of pixel tile mt (or -1 past the batch)
        auto goff = [&](int mt, int j, bool& ok) -> size_t {
            const int p = mt * 32 + j * RP + prow;
            const int x = col0 + (p & (G::TW - 1));
            const int y = row0 + ((p >> G::LOGTW) & (G::TH - 1));
            const int n = n_base + mt / MT_PER_IMG;
            ok = n < a.B;
            if (hd) return (((size_t)n * a.heads + hhead) * HWo + (size_t)y * G::W + x) * ostride + co_out;
            return (((size_t)n * H + y) * G::W + x) * ostride + co_out;
        };
        typedef typename Raw4<ST>::type R4;
        R4 rcur[NP], rnext[NP];  // residual quads: the loads for tile mt+1 are in flight while tile mt is processed
        auto issue_resid = [&](int mt, R4 (&rr)[NP]) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                bool ok;
                const size_t go = goff(mt, j, ok);
                rr[j] = R4{};
                if (ok) rr[j] = raw_load4(resid + go);
            }
        };
        if (resid) issue_resid(0, rcur);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int im = mt / MT_PER_IMG;
            const int n = n_base + im;
            if (resid && mt + 1 < 4) issue_resid(mt + 1, rnext);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    *reinterpret_cast<float*>(ep + acc_row(i, h) * EP_PITCH + (nt * 32 + r) * 4) = acc[mt][nt][i];
            f32x4 add = bias4;
            if (!qk && a.temb && n < a.B) add += *reinterpret_cast<const f32x4*>(a.temb + (size_t)n * a.temb_stride + co0);
            f32x4 gate4 = {1.f, 1.f, 1.f, 1.f};
            if (OUTMODE == OUT_TOK && a.gate && n < a.B) gate4 = *reinterpret_cast<const f32x4*>(a.gate + (size_t)n * a.gate_stride + co0);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int pl = j * RP + prow;
                f32x4 v = *reinterpret_cast<const f32x4*>(ep + pl * EP_PITCH + c4 * 16);
                bool ok;
                const size_t go = goff(mt, j, ok);
                if (ok) {
                    v += add;
                    if (OUTMODE == OUT_TOK) {
                        if (a.act == 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
                        }
                        v *= gate4;
                    }
                    if (resid) v += widen4(rcur[j]);
                    v *= a.scale;
                    f32x4 vr;  // the values as the consumer will read them
                    if constexpr (OUTMODE == OUT_HEADS && std::is_same<T, bf16x3>::value) {
                        vr = store4_split(reinterpret_cast<__bf16*>(out) + go, a.heads_lo_off, v);  // q | k as hi and lo bf16 planes
                    } else {
                        vr = store4(out + go, v);
                    }
                    ssum[im] += (vr[0] + vr[1]) + (vr[2] + vr[3]);
                    ssq[im] += (vr[0] * vr[0] + vr[1] * vr[1]) + (vr[2] * vr[2] + vr[3] * vr[3]);
                }
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) rcur[j] = rnext[j];
        }
        if (!qk && a.stats) {
#pragma unroll
            for (int im = 0; im < G::IMGS; ++im) {
                float sv = ssum[im], qv = ssq[im];
#pragma unroll
                for (int sh = QPW; sh < 64; sh <<= 1) {
                    sv += __shfl_xor(sv, sh);
                    qv += __shfl_xor(qv, sh);
                }
                const int n = n_base + im;
                if (prow == 0 && n < a.B)
                    a.stats[((size_t)n * G::TPI + slot) * (a.Cout >> 2) + (co0 >> 2)] = make_float2(sv, qv);
            }
        }
    } else {  // the v^T plane of the qkv projection: [B][256][HW], four consecutive pixels per 8/16-byte store
        static_assert(OUTMODE != OUT_QKV || NT == 2, "qkv epilogue assumes 64 channels per wave");
        ST* vt = reinterpret_cast<ST*>(a.vt_out);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // channel of this lane: row of the [256][HW] plane (OUT_QKV), or of head h's [head_dim][T] plane (OUT_HEADS)
            const int cg = nblk * (128 * NT) + wave * (32 * NT) + nt * 32 + r;  // global output channel
            const int cl = (OUTMODE == OUT_HEADS) ? cg - 2 * a.heads * a.head_dim : wave * 64 + nt * 32 + r;
            const float bias = a.bias ? a.bias[OUTMODE == OUT_HEADS ? cg : nblk * 256 + cl] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int i4 = 0; i4 < 16; i4 += 4) {
                    const int p = mt * 32 + acc_row(i4, h);  // 4 consecutive pixels p..p+3 (same row)
                    const int x = col0 + (p & (G::TW - 1));
                    const int y = row0 + ((p >> G::LOGTW) & (G::TH - 1));
                    const int n = n_base + (p >> (G::LOGTW + G::LOGTH));
                    if (n < a.B)
                    {
                        const size_t vo = ((size_t)n * (OUTMODE == OUT_HEADS ? a.heads * a.head_dim : 256) + cl) * HWo + y * G::W + x;
                        const f32x4 vv = {acc[mt][nt][i4] + bias, acc[mt][nt][i4 + 1] + bias, acc[mt][nt][i4 + 2] + bias,
                                          acc[mt][nt][i4 + 3] + bias};
                        if constexpr (OUTMODE == OUT_HEADS && std::is_same<T, bf16x3>::value)
                            store4_split(reinterpret_cast<__bf16*>(vt) + vo, a.heads_lo_off, vv);  // v^T as hi and lo bf16 planes
                        else
                            store4(vt + vo, vv);
                    }
                }
            }
        }
    }
}

// ---- weight packing ---------------------------------------------------------------------------------
// packed[nt][chunk][tap][kk][lane][j] = W[cout = nt*32 + (lane&31)][cin = chunk*KC + kk*16 + 8*(lane>>5) + j][tap]
// bf16x3: each fragment is followed by its lo plane, packed[...][kk][part][lane][j], part 0 = bf16(w), part 1 = bf16(w - hi)
template <typename T>
__global__ void pack_conv_weights_kernel(const float* __restrict__ w, typename DT<T>::WT* __restrict__ out, int cout, int cin, int ks,
                                         int qkv_perm) {
    constexpr int KC = DT<T>::KC, KK = KC / 16, WP = DT<T>::WPARTS;
    const int taps = ks * ks;
    const size_t total = (size_t)cout * cin * taps;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t t = idx;
        const int j = t % 8; t /= 8;
        const int lane = t % 64; t /= 64;
        const int kk = t % KK; t /= KK;
        const int tap = t % taps; t /= taps;
        const int nchunk = cin / KC;
        const int chunk = t % nchunk; t /= nchunk;
        const int nt = (int)t;
        int co = nt * 32 + (lane & 31);
        if (qkv_perm) {  // packed channel o' = plane*C + c  <-  reference channel c*3 + plane
            const int C = cout / 3;
            co = (co % C) * 3 + co / C;
        }
        const int ci = chunk * KC + kk * 16 + 8 * (lane >> 5) + j;
        const float v = w[((size_t)co * cin + ci) * taps + tap];
        if constexpr (WP == 1) {
            out[idx] = (typename DT<T>::WT)v;
        } else {
            const size_t o = (idx / 512) * 1024 + (idx % 512);
            const __bf16 hi = (__bf16)v;
            out[o] = hi;
            out[o + 512] = (__bf16)(v - (float)hi);
        }
    }
}

size_t g_debug_extra_lds = 0;  // ablation only: pad the dynamic LDS request to force one workgroup per CU
bool g_prepare_only = false;  // conv_prepare_all(): walk the dispatch tables, set attributes, launch nothing

template <typename T, int KS, int PRO, int RES, int LOGW, int OUTMODE, int ABL = 0>
int launch_one(const ConvArgs& a, hipStream_t stream) {
    using G = Geom<KS, LOGW>;
    const int tiles = (G::IMGS > 1) ? (a.B + G::IMGS - 1) / G::IMGS : a.B * G::TPI;
    const size_t lds = 2 * (size_t)G::ABUF + (ABL ? g_debug_extra_lds : 0);
    auto kern = conv_fused_kernel<T, KS, PRO, RES, LOGW, OUTMODE, ABL>;
    // raise the dynamic-LDS cap once per instantiation and device (never inside stream capture: conv_prepare_all does it up front)
    static bool attr_done[16] = {};
    const int dev = fg_device_slot();
    if (dev < 0) return (int)hipErrorInvalidDevice;
    if (!attr_done[dev] || (ABL && g_debug_extra_lds)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done[dev] = true;
    }
    if (g_prepare_only) return 0;
    constexpr int NCOL = 128 * conv_nt(KS, LOGW, OUTMODE);
    dim3 grid(tiles, (a.Cout + NCOL - 1) / NCOL);  // exact for the convolutions (Cout % 256 == 0); token GEMMs round up
    hipLaunchKernelGGL(kern, grid, dim3(NTHR), lds, stream, a);
    return (int)hipGetLastError();
}

template <typename T, int KS, int PRO, int RES, int OUTMODE>
int launch_w(const ConvArgs& a, hipStream_t s) {
    switch (a.W) {
        case 32: return launch_one<T, KS, PRO, RES, 5, OUTMODE>(a, s);
        case 16: return launch_one<T, KS, PRO, RES, 4, OUTMODE>(a, s);
        case 8: return launch_one<T, KS, PRO, RES, 3, OUTMODE>(a, s);
    }
    return (int)hipErrorInvalidValue;
}

template <typename T>
int launch_t(int ks, int pro, int res, int outmode, const ConvArgs& a, hipStream_t s) {
    // the instantiations the EDM U-Net needs (EDM/network.py:237-272)
    if (outmode == OUT_QKV) {
        if (ks == 1 && pro == PRO_GN && res == RES_NONE) return launch_w<T, 1, PRO_GN, RES_NONE, OUT_QKV>(a, s);
        return (int)hipErrorInvalidValue;
    }
    if (outmode == OUT_TOK || outmode == OUT_HEADS) {  // transformer GEMMs: 256 tokens per image as a 16x16 "image", 1x1 "conv"
        if (ks != 1 || pro != PRO_NONE || res != RES_NONE || a.W != 16) return (int)hipErrorInvalidValue;
        return outmode == OUT_TOK ? launch_one<T, 1, PRO_NONE, RES_NONE, 4, OUT_TOK>(a, s) : launch_one<T, 1, PRO_NONE, RES_NONE, 4, OUT_HEADS>(a, s);
    }
    if (ks == 3 && pro == PRO_GN_SILU) {
        if (res == RES_NONE) return launch_w<T, 3, PRO_GN_SILU, RES_NONE, OUT_NHWC>(a, s);
        if (res == RES_UP) return launch_w<T, 3, PRO_GN_SILU, RES_UP, OUT_NHWC>(a, s);
    }
    // down-sampling blocks: the operand was normalised, activated and pooled by gn_silu_pool (misc.hip)
    if (ks == 3 && pro == PRO_NONE && res == RES_NONE) return launch_w<T, 3, PRO_NONE, RES_NONE, OUT_NHWC>(a, s);
    if (ks == 1 && pro == PRO_NONE) {
        if (res == RES_NONE) return launch_w<T, 1, PRO_NONE, RES_NONE, OUT_NHWC>(a, s);
        if (res == RES_DOWN) return launch_w<T, 1, PRO_NONE, RES_DOWN, OUT_NHWC>(a, s);
        if (res == RES_UP) return launch_w<T, 1, PRO_NONE, RES_UP, OUT_NHWC>(a, s);
    }
    return (int)hipErrorInvalidValue;
}

int dispatch_t(int dtype, int ks, int pro, int res, int outmode, const ConvArgs& a, hipStream_t s) {
    switch (dtype) {
        case 0: return launch_t<float>(ks, pro, res, outmode, a, s);
        case 1: return launch_t<__bf16>(ks, pro, res, outmode, a, s);
        case 2: return launch_t<bf16x3>(ks, pro, res, outmode, a, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace

// compile-time ablation builds of the dominant shape (bf16, 3x3, no resample, 32x32): scripts/conv_ablate.py only
int launch_conv_debug(int dtype, const ConvArgs& a, hipStream_t stream) {
    if (dtype != 1 || a.W != 32) return (int)hipErrorInvalidValue;
    g_debug_extra_lds = (a.dbg & 16) ? 48 * 1024 : 0;
    switch (a.dbg & 15) {
        case 0: return launch_one<__bf16, 3, PRO_GN_SILU, RES_NONE, 5, OUT_NHWC, 0>(a, stream);
        case 1: return launch_one<__bf16, 3, PRO_GN_SILU, RES_NONE, 5, OUT_NHWC, 1>(a, stream);
        case 2: return launch_one<__bf16, 3, PRO_GN_SILU, RES_NONE, 5, OUT_NHWC, 2>(a, stream);
        case 3: return launch_one<__bf16, 3, PRO_GN_SILU, RES_NONE, 5, OUT_NHWC, 3>(a, stream);
        case 4: return launch_one<__bf16, 3, PRO_GN_SILU, RES_NONE, 5, OUT_NHWC, 4>(a, stream);
        case 5: return launch_one<__bf16, 3, PRO_GN_SILU, RES_NONE, 5, OUT_NHWC, 5>(a, stream);
        case 6: return launch_one<__bf16, 3, PRO_GN_SILU, RES_NONE, 5, OUT_NHWC, 6>(a, stream);
        case 7: return launch_one<__bf16, 3, PRO_GN_SILU, RES_NONE, 5, OUT_NHWC, 7>(a, stream);
    }
    return (int)hipErrorInvalidValue;
}

int launch_conv_fused(int dtype, int ks, int pro, int res, int outmode, const ConvArgs& a, hipStream_t stream) {
    const int kc = dtype == 1 ? DT<__bf16>::KC : DT<float>::KC;  // 32 for fp32 and bf16x3
    if (a.H != a.W || (a.W != 8 && a.W != 16 && a.W != 32)) return (int)hipErrorInvalidValue;
    const bool tok = outmode == OUT_TOK || outmode == OUT_HEADS;
    if ((a.C1 % kc) || (a.C2 % kc) || (a.Cout % (tok ? 64 : 256)) || a.B <= 0) return (int)hipErrorInvalidValue;
    if (outmode == OUT_HEADS && (a.heads <= 0 || (a.head_dim % 4) || a.Cout != 3 * a.heads * a.head_dim || ((a.heads * a.head_dim) % 64) ||
                                 !a.q_out || !a.k_out || !a.vt_out))
        return (int)hipErrorInvalidValue;
    if (res == RES_NONE && (a.Hs != a.H || a.Ws != a.W)) return (int)hipErrorInvalidValue;
    if (res == RES_DOWN && (a.Hs != 2 * a.H || a.Ws != 2 * a.W)) return (int)hipErrorInvalidValue;
    if (res == RES_UP && (2 * a.Hs != a.H || 2 * a.Ws != a.W)) return (int)hipErrorInvalidValue;
    if (pro != PRO_NONE && !a.ab) return (int)hipErrorInvalidValue;
    if (outmode == OUT_QKV && (a.Cout != 768 || a.W > 16)) return (int)hipErrorInvalidValue;
    if (conv_ws_enabled() && conv_ws_supported(dtype, ks, pro, res, outmode, a)) return launch_conv_ws(res, a, stream, false, pro);
    if (dtype == 2 && conv_ws_enabled() && conv_x3ws_supported(ks, pro, res, outmode, a)) return launch_conv_x3ws(res, a, stream, false, pro);
    return dispatch_t(dtype, ks, pro, res, outmode, a, stream);
}

// FASTGEN_AMD_CONV_WS=0 keeps every conv on conv_fused_kernel (A/B measurements); read once.
bool conv_ws_enabled() {
    static const bool on = [] {
        const char* e = getenv("FASTGEN_AMD_CONV_WS");
        return !(e && e[0] == '0');
    }();
    return on;
}

int conv_launch_stat_slots(int dtype, int ks, int pro, int res, int outmode, const ConvArgs& a) {
    if (dtype == 2 && conv_ws_enabled() && conv_x3ws_supported(ks, pro, res, outmode, a)) return conv_x3ws_stat_slots(a.W);
    return conv_stat_slots(a.W);
}

int conv_stat_slots(int W) { return W == 32 ? Geom<3, 5>::TPI : (W == 16 ? Geom<3, 4>::TPI : Geom<3, 3>::TPI); }

int conv_prepare_all(int dtype) {
    g_prepare_only = true;
    int rc = 0;
    ConvArgs a{};
    a.B = 1;
    a.Cout = 256;
    const int ws[3] = {8, 16, 32};
    for (int wi = 0; wi < 3 && !rc; ++wi) {
        a.W = a.H = ws[wi];
        for (int res = 0; res < 3 && !rc; ++res) {
            if (res != RES_DOWN)
                rc = dispatch_t(dtype, 3, PRO_GN_SILU, res, OUT_NHWC, a, nullptr);
            if (!rc) rc = dispatch_t(dtype, 1, PRO_NONE, res, OUT_NHWC, a, nullptr);
        }
        if (!rc) rc = dispatch_t(dtype, 3, PRO_NONE, RES_NONE, OUT_NHWC, a, nullptr);
        if (!rc && a.W <= 16) rc = dispatch_t(dtype, 1, PRO_GN, RES_NONE, OUT_QKV, a, nullptr);
        if (!rc && a.W == 16) rc = dispatch_t(dtype, 1, PRO_NONE, RES_NONE, OUT_TOK, a, nullptr);
        if (!rc && a.W == 16) rc = dispatch_t(dtype, 1, PRO_NONE, RES_NONE, OUT_HEADS, a, nullptr);
    }
    g_prepare_only = false;
    if (!rc && dtype == 2) {
        ConvArgs w{};
        for (int res : {RES_NONE, RES_UP})
            for (int ww : {32, 16}) {
                w.W = w.H = ww;
                if (!rc) rc = launch_conv_x3ws(res, w, nullptr, true, PRO_GN_SILU);
                if (!rc && res == RES_NONE) rc = launch_conv_x3ws(res, w, nullptr, true, PRO_NONE);
            }
    }
    if (!rc && dtype == 1) {
        ConvArgs w{};
        for (int res : {RES_NONE, RES_UP}) {
            w.W = w.H = 32;
            if (!rc) rc = launch_conv_ws(res, w, nullptr, true);
            if (res == RES_NONE) {  // the 128-input-channel instantiation
                w.C1 = -128;
                if (!rc) rc = launch_conv_ws(res, w, nullptr, true);
                w.C1 = 0;
            }
            w.W = w.H = 16;
            if (!rc) rc = launch_conv_ws(res, w, nullptr, true);
        }
    }
    return rc;
}

size_t conv_pack_elems(int cout, int cin, int ks) { return (size_t)cout * cin * ks * ks; }

int launch_pack_conv_weights(int dtype, const float* w, void* out, int cout, int cin, int ks, int qkv_perm,
                             hipStream_t stream) {
    const size_t total = conv_pack_elems(cout, cin, ks);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == 1)
        hipLaunchKernelGGL(pack_conv_weights_kernel<__bf16>, dim3(grid), dim3(256), 0, stream, w, (__bf16*)out, cout, cin, ks, qkv_perm);
    else if (dtype == 2)  // 2 x bf16 per weight: the same bytes as the fp32 packing
        hipLaunchKernelGGL(pack_conv_weights_kernel<bf16x3>, dim3(grid), dim3(256), 0, stream, w, (__bf16*)out, cout, cin, ks, qkv_perm);
    else
        hipLaunchKernelGGL(pack_conv_weights_kernel<float>, dim3(grid), dim3(256), 0, stream, w, (float*)out, cout, cin, ks, qkv_perm);
    return (int)hipGetLastError();
}
