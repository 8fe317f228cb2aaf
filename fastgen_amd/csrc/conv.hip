// Fused GroupNorm-apply + SiLU -> Conv2d(3x3 | 1x1) -> bias (+temb) (+residual) * scale, as an implicit GEMM on
// the CDNA4 matrix cores.  Replaces, per UNetBlock (reference fastgen/networks/EDM/network.py:274-299):
//     conv0(silu(norm0(x))) + affine(emb)        -> PRO_GN_SILU, temb epilogue
//     (conv1(silu(norm1(h))) + skip) * sqrt(.5)  -> PRO_GN_SILU, residual epilogue
//     skip / qkv / proj 1x1 convs                -> KS = 1
// and the resampling of Conv2d.forward (:114-121) folded into the operand load (RES_DOWN = 2x2 mean of the
// *transformed* input, RES_UP = nearest replication), and torch.cat (:560) as two source pointers.
//
// Tiling (one workgroup = 512 threads = 8 waves, one workgroup per CU):
//   M = 256 output pixels (8 rows x 32 | 16 x 16 | 4 images x 8 x 8), N = 256 output channels, K = taps x Cin.
//   wave w owns output channels [32w, 32w+32) for all 256 pixels: 8 accumulator tiles of 32x32 (128 VGPRs).
//   A (activations): per K-chunk of KC input channels the (rows+2) x (W+2) halo of the pixel tile is transformed
//     ONCE (GN affine + SiLU, cast to the compute dtype) and parked in LDS (144-byte pixel pitch: conflict-free
//     ds_read_b128); all 9 taps read shifted windows of it.  Double-buffered: chunk c+1 is staged while chunk c
//     is multiplied, global loads issued before the MFMA block and consumed after it.  One barrier per chunk.
//   B (weights): pre-packed in MFMA fragment order, read straight from global/L2 into registers one (chunk, tap)
//     step ahead — each wave reads only its own 32 output channels, so no LDS and no barrier for B.
#include "common.h"
#include "conv.h"

namespace {

constexpr int PITCH = 144;  // bytes per halo pixel in LDS: 128 B of channels + 16 B pad (odd multiple of 16 B)

template <int KS, int LOGW>
struct Geom {
    static constexpr int W = 1 << LOGW;
    static constexpr int LOGTH = (LOGW == 4) ? 4 : 3;
    static constexpr int TH = 1 << LOGTH;           // tile rows per image
    static constexpr int IMGS = 256 / (TH * W);     // images per tile (4 at 8x8, else 1)
    static constexpr int TPI = (W * W) / (TH * W) > 0 ? (W * W) / (TH * W) : 1;  // tiles per image (square images)
    static constexpr int PAD = KS / 2;
    static constexpr int HW_ = W + 2 * PAD;
    static constexpr int HH_ = TH + 2 * PAD;
    static constexpr int HALO_PIX = IMGS * HH_ * HW_;
    static constexpr int TAPS = KS * KS;
    // halo index (tap 0,0) of tile pixel p in [0,256); additive in (p & ~31) and (p & 31)
    static __host__ __device__ constexpr int hp0(int p) {
        return (((p >> (LOGW + LOGTH)) * HH_) + ((p >> LOGW) & (TH - 1))) * HW_ + (p & (W - 1));
    }
};

template <int PRO, bool FAST>
__device__ __forceinline__ float pro_apply(float x, float2 ab) {
    if (PRO == PRO_NONE) return x;
    float y = fmaf(x, ab.x, ab.y);
    if (PRO == PRO_GN_SILU) y = silu_f<FAST>(y);
    return y;
}

template <typename T, int KS, int PRO, int RES, int LOGW, int OUTMODE>
__global__ __launch_bounds__(512, 2) void conv_fused_kernel(const ConvArgs a) {
    using G = Geom<KS, LOGW>;
    constexpr int KC = DT<T>::KC;
    constexpr bool FAST = DT<T>::FAST;
    constexpr int KK = KC / 16;   // 16-deep MFMA steps per chunk
    constexpr int OPP = KC / 8;   // 8-channel octets per pixel and chunk
    constexpr int LOG_OPP = (OPP == 8) ? 3 : 2;
    constexpr int NITEMS = (G::HALO_PIX * OPP + 511) / 512;      // staging items per thread and chunk
    constexpr int IPS = (NITEMS + G::TAPS - 1) / G::TAPS;        // items staged per (chunk, tap) step
    constexpr int ABUF = G::HALO_PIX * PITCH;
    constexpr bool DEFER = (RES != RES_DOWN) && (KS == 3);       // split load / transform+write around the MFMAs
    constexpr bool AB_REGS = (PRO != PRO_NONE) && (G::IMGS == 1);  // per-chunk GN coefficients live in registers
    constexpr bool PIPE_A = (sizeof(T) == 2);                      // two A-fragment register sets (bf16 only)

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
    const int row0 = (G::IMGS > 1) ? 0 : (tile % G::TPI) * G::TH;

    // this wave's packed weights: [cout/32][step][kk][lane][8]
    const T* wp = reinterpret_cast<const T*>(a.wpack) + (size_t)(nblk * 8 + wave) * nsteps * (KK * 512) + lane * 8;

    // staging role of this thread: fixed channel octet, halo pixels hq0 + i*(512/OPP)
    const int oct = tid & (OPP - 1);
    const int hq0 = tid >> LOG_OPP;

    // ---- staging helpers ------------------------------------------------------------------------------
    auto decode = [&](int hq, int& n, int& y, int& x) -> bool {
        const int hx = hq % G::HW_;
        const int t = hq / G::HW_;
        const int hy = t % G::HH_;
        const int img = t / G::HH_;
        y = row0 + hy - G::PAD;
        x = hx - G::PAD;
        n = n_base + img;
        return (hq < G::HALO_PIX) && (y >= 0) && (y < H) && (x >= 0) && (x < G::W) && (n < a.B);
    };
    auto src_ptr = [&](int chunk, int n, int sy, int sx) -> const float* {
        const int c0 = chunk * KC + oct * 8;
        const size_t sp = ((size_t)n * a.Hs + sy) * a.Ws + sx;
        return (c0 < a.C1) ? a.src1 + sp * a.C1 + c0 : a.src2 + sp * a.C2 + (c0 - a.C1);
    };
    auto load8 = [&](const float* p, float (&v)[8]) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
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

    // phase 1 of an item: issue the global loads (RES_NONE / RES_UP only)
    auto item_load = [&](int chunk, int i, float (&raw)[8], bool& valid) {
        int n, y, x;
        valid = decode(hq0 + i * (512 / OPP), n, y, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = 0.f;
        if (valid) {
            const int sy = (RES == RES_UP) ? (y >> 1) : y;
            const int sx = (RES == RES_UP) ? (x >> 1) : x;
            load8(src_ptr(chunk, n, sy, sx), raw);
        }
    };
    // phase 2: transform and park in LDS
    auto item_finish = [&](int chunk, int i, char* abuf, float (&raw)[8], bool valid) {
        const int hq = hq0 + i * (512 / OPP);
        if (hq >= G::HALO_PIX) return;
        float o[8];
        if (DEFER) {
            if (valid) {
                if (PRO != PRO_NONE && !AB_REGS) {
                    int n, y, x;
                    decode(hq, n, y, x);
                    float2 ab[8];
                    load_ab(chunk, n, ab);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = pro_apply<PRO, FAST>(raw[j], ab[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = pro_apply<PRO, FAST>(raw[j], abr[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = 0.f;
            }
        } else {  // synchronous path: RES_DOWN (mean of the four transformed source pixels) and all 1x1 convs
            constexpr int ND = (RES == RES_DOWN) ? 4 : 1;
            int n, y, x;
            const bool ok = decode(hq, n, y, x);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
            if (ok) {
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
        }
        store_frag(reinterpret_cast<T*>(abuf + hq * PITCH) + oct * 8, o);
    };

    // ---- prologue: chunk 0 of A, step 0 of B ------------------------------------------------------
    if (AB_REGS) load_ab(0, n_base, abr);
#pragma unroll
    for (int i = 0; i < NITEMS; ++i) {
        float raw[8];
        bool valid = false;
        if (DEFER) item_load(0, i, raw, valid);
        item_finish(0, i, smem, raw, valid);
    }
    Frag8<T> bcur[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) bcur[kk] = load_frag(wp + kk * 512);

    f32x16 acc[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    const int lane_off = G::hp0(r) * PITCH + h * (8 * (int)sizeof(T));
    __syncthreads();

    // ---- main loop ------------------------------------------------------------------------------------
    int step = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const char* abuf = smem + (chunk & 1) * ABUF;
        char* anext = smem + ((chunk + 1) & 1) * ABUF;
        const bool stage_next = (chunk + 1 < nchunk);
#pragma unroll 1
        for (int tap = 0; tap < G::TAPS; ++tap, ++step) {
            // (1) weights: each fragment is refilled in place for the NEXT step right after its last use below
            //     (clamped: the last step re-reads itself), so B needs KK fragments, one step of prefetch distance
            const T* pnext = wp + (size_t)((step + 1 < nsteps) ? step + 1 : step) * (KK * 512);
            // (2) issue this step's share of the next chunk's activation loads
            float raw[IPS][8];
            bool valid[IPS];
            const bool do_stage = stage_next && (tap * IPS < NITEMS);
            if (do_stage) {
                if (AB_REGS && tap == 0) load_ab(chunk + 1, n_base, abr);
                if (DEFER) {
#pragma unroll
                    for (int q = 0; q < IPS; ++q) item_load(chunk + 1, tap * IPS + q, raw[q], valid[q]);
                }
            }
            // (3) multiply: 8 pixel tiles x KK k-steps against this wave's 32 output channels.  The 8 A fragments of
            //     a k-step are read from LDS as one group (latency paid once per group, not per MFMA); with bf16
            //     operands the next k-step's group is issued before this k-step's 8 back-to-back MFMAs.
            const int tap_off = ((tap / KS) * G::HW_ + (tap % KS)) * PITCH;
            const char* abase = abuf + lane_off + tap_off;
            auto read_a = [&](int kk, Frag8<T> (&af)[8]) {
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
                    af[mt] = load_frag(reinterpret_cast<const T*>(abase + G::hp0(mt * 32) * PITCH) + kk * 16);
            };
            if (PIPE_A) {
                Frag8<T> a0[8], a1[8];
                read_a(0, a0);
#pragma unroll
                for (int kk = 0; kk < KK; kk += 2) {
                    read_a(kk + 1, a1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt) mma16(acc[mt], a0[mt], bcur[kk]);
                    bcur[kk] = load_frag(pnext + kk * 512);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kk + 2 < KK) read_a(kk + 2, a0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt) mma16(acc[mt], a1[mt], bcur[kk + 1]);
                    bcur[kk + 1] = load_frag(pnext + (kk + 1) * 512);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    Frag8<T> af[8];
                    read_a(kk, af);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt) mma16(acc[mt], af[mt], bcur[kk]);
                    bcur[kk] = load_frag(pnext + kk * 512);
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
    const int cl = wave * 32 + r;  // channel within this 256-wide block
    const int HWo = H * G::W;
    if (OUTMODE == OUT_NHWC) {
        const int co = nblk * 256 + cl;
        const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int p = mt * 32 + acc_row(i, h);
                const int x = p & (G::W - 1);
                const int y = row0 + ((p >> LOGW) & (G::TH - 1));
                const int n = n_base + (p >> (LOGW + G::LOGTH));
                if (n < a.B) {
                    const size_t gp = ((size_t)n * H + y) * G::W + x;
                    float v = acc[mt][i] + bias;
                    if (a.temb) v += a.temb[(size_t)n * a.temb_stride + co];
                    if (a.resid) v += a.resid[gp * a.Cout + co];
                    a.out[gp * a.Cout + co] = v * a.scale;
                }
            }
        }
    } else {  // OUT_QKV: plane nblk of {q, k, v^T} in the compute dtype
        const float bias = a.bias ? a.bias[nblk * 256 + cl] : 0.f;
        T* qk = reinterpret_cast<T*>(nblk == 0 ? a.q_out : a.k_out);
        T* vt = reinterpret_cast<T*>(a.vt_out);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int i4 = 0; i4 < 16; i4 += 4) {
                const int p = mt * 32 + acc_row(i4, h);  // 4 consecutive pixels p..p+3 (same row)
                const int x = p & (G::W - 1);
                const int y = row0 + ((p >> LOGW) & (G::TH - 1));
                const int n = n_base + (p >> (LOGW + G::LOGTH));
                if (n < a.B) {
                    const int pix = y * G::W + x;
                    if (nblk < 2) {
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            qk[((size_t)n * HWo + pix + d) * 256 + cl] = (T)(acc[mt][i4 + d] + bias);
                    } else {
                        T* dst = vt + ((size_t)n * 256 + cl) * HWo + pix;
#pragma unroll
                        for (int d = 0; d < 4; ++d) dst[d] = (T)(acc[mt][i4 + d] + bias);
                    }
                }
            }
        }
    }
}

// ---- weight packing ---------------------------------------------------------------------------------
// packed[nt][chunk][tap][kk][lane][j] = W[cout = nt*32 + (lane&31)][cin = chunk*KC + kk*16 + 8*(lane>>5) + j][tap]
template <typename T>
__global__ void pack_conv_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int cin, int ks,
                                         int qkv_perm) {
    constexpr int KC = DT<T>::KC, KK = KC / 16;
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
        out[idx] = (T)w[((size_t)co * cin + ci) * taps + tap];
    }
}

bool g_prepare_only = false;  // conv_prepare_all(): walk the dispatch tables, set attributes, launch nothing

template <typename T, int KS, int PRO, int RES, int LOGW, int OUTMODE>
int launch_one(const ConvArgs& a, hipStream_t stream) {
    using G = Geom<KS, LOGW>;
    const int tiles = (G::IMGS > 1) ? (a.B + G::IMGS - 1) / G::IMGS : a.B * G::TPI;
    const size_t lds = 2 * (size_t)G::HALO_PIX * PITCH;
    auto kern = conv_fused_kernel<T, KS, PRO, RES, LOGW, OUTMODE>;
    static bool attr_done = false;  // raise the dynamic-LDS cap once per instantiation (never inside stream capture)
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (g_prepare_only) return 0;
    dim3 grid(tiles, a.Cout / 256);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, a);
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
    if (ks == 3 && pro == PRO_GN_SILU) {
        if (res == RES_NONE) return launch_w<T, 3, PRO_GN_SILU, RES_NONE, OUT_NHWC>(a, s);
        if (res == RES_DOWN) return launch_w<T, 3, PRO_GN_SILU, RES_DOWN, OUT_NHWC>(a, s);
        if (res == RES_UP) return launch_w<T, 3, PRO_GN_SILU, RES_UP, OUT_NHWC>(a, s);
    }
    if (ks == 1 && pro == PRO_NONE) {
        if (res == RES_NONE) return launch_w<T, 1, PRO_NONE, RES_NONE, OUT_NHWC>(a, s);
        if (res == RES_DOWN) return launch_w<T, 1, PRO_NONE, RES_DOWN, OUT_NHWC>(a, s);
        if (res == RES_UP) return launch_w<T, 1, PRO_NONE, RES_UP, OUT_NHWC>(a, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace

int launch_conv_fused(int dtype, int ks, int pro, int res, int outmode, const ConvArgs& a, hipStream_t stream) {
    const int kc = dtype ? DT<__bf16>::KC : DT<float>::KC;
    if (a.H != a.W || (a.W != 8 && a.W != 16 && a.W != 32)) return (int)hipErrorInvalidValue;
    if ((a.C1 % kc) || (a.C2 % kc) || (a.Cout % 256) || a.B <= 0) return (int)hipErrorInvalidValue;
    if (res == RES_NONE && (a.Hs != a.H || a.Ws != a.W)) return (int)hipErrorInvalidValue;
    if (res == RES_DOWN && (a.Hs != 2 * a.H || a.Ws != 2 * a.W)) return (int)hipErrorInvalidValue;
    if (res == RES_UP && (2 * a.Hs != a.H || 2 * a.Ws != a.W)) return (int)hipErrorInvalidValue;
    if (pro != PRO_NONE && !a.ab) return (int)hipErrorInvalidValue;
    if (outmode == OUT_QKV && (a.Cout != 768 || a.W > 16)) return (int)hipErrorInvalidValue;
    return dtype ? launch_t<__bf16>(ks, pro, res, outmode, a, stream) : launch_t<float>(ks, pro, res, outmode, a, stream);
}

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
            if (dtype) {
                rc = launch_t<__bf16>(3, PRO_GN_SILU, res, OUT_NHWC, a, nullptr);
                if (!rc) rc = launch_t<__bf16>(1, PRO_NONE, res, OUT_NHWC, a, nullptr);
            } else {
                rc = launch_t<float>(3, PRO_GN_SILU, res, OUT_NHWC, a, nullptr);
                if (!rc) rc = launch_t<float>(1, PRO_NONE, res, OUT_NHWC, a, nullptr);
            }
        }
        if (!rc && a.W <= 16) rc = dtype ? launch_t<__bf16>(1, PRO_GN, RES_NONE, OUT_QKV, a, nullptr) : launch_t<float>(1, PRO_GN, RES_NONE, OUT_QKV, a, nullptr);
    }
    g_prepare_only = false;
    return rc;
}

size_t conv_pack_elems(int cout, int cin, int ks) { return (size_t)cout * cin * ks * ks; }

int launch_pack_conv_weights(int dtype, const float* w, void* out, int cout, int cin, int ks, int qkv_perm,
                             hipStream_t stream) {
    const size_t total = conv_pack_elems(cout, cin, ks);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype)
        hipLaunchKernelGGL(pack_conv_weights_kernel<__bf16>, dim3(grid), dim3(256), 0, stream, w, (__bf16*)out, cout, cin, ks, qkv_perm);
    else
        hipLaunchKernelGGL(pack_conv_weights_kernel<float>, dim3(grid), dim3(256), 0, stream, w, (float*)out, cout, cin, ks, qkv_perm);
    return (int)hipGetLastError();
}
