// Wave-specialised, persistent variant of the fused GroupNorm+SiLU -> Conv2d 3x3 -> bias/temb/residual/scale (+ GroupNorm
// partial statistics) kernel of conv.hip, for the shapes that dominate the EDM U-Net: bf16, 3x3, 256 output channels,
// 32x32 and 16x16 outputs, with or without the nearest 2x up-sampling folded into the load.  Same arithmetic and same
// operands (NHWC bf16 activations, MFMA-fragment-packed weights, [B][slot][C/4] statistics) as conv_fused_kernel.
//
// Why a second kernel: conv_fused_kernel runs its staging prologue, MFMA loop and epilogue back to back in every wave
// and relies on a second resident workgroup to fill the gaps; measured (profiles/r01_conv_ablation*.txt) the two
// workgroups fall into lockstep and the phases add up (MFMA + weight stream 331 us, + staging 80, + epilogue 60..120).
// Here the phases run on DIFFERENT waves of one 512-thread workgroup (one per CU, persistent over pixel tiles):
//   waves 0-3  "consumers": nothing but ds_read(A) -> v_mfma <- global weights; 4 x 2 accumulator tiles of 32x32 each
//              (128 pixels x 64 output channels per wave), weights streamed from L2 through a 6-deep register ring.
//   waves 4-7  "producers": stage the next step's (8+2) x (16+2) halo of 32 input channels into LDS (GroupNorm affine + SiLU
//              applied once per element), and retire the PREVIOUS tile: read its fp32 accumulators from the LDS hand-off
//              buffer, add bias / temb / residual, scale, round to bf16, store, and reduce the GroupNorm partial sums.
// One s_barrier per step (32 input channels x 9 taps = 144 MFMAs per consumer wave); a consumer wave and a producer wave
// share each SIMD, so the producers' VALU / memory work issues in the shadow of the consumers' MFMAs.
//
// LDS (one workgroup owns the CU's 160 KB):  [0, 128 KB) fp32 hand-off tile [128 px][256 ch], 16-byte quads XOR-swizzled by
// pixel-row bit 2 (the two lane halves of an accumulator column land in disjoint banks); then two 15,360-byte halo buffers
// (80-byte pixel pitch = 64 B of channels + 16 B pad, rows padded to 1536 B: every ds_read_b128 of an A fragment is
// conflict-free).
#include "common.h"
#include "conv.h"

namespace {

constexpr int WS_NTHR = 512;
constexpr int WS_KC = 32;               // input channels per pipeline step
constexpr int WS_PA = 80;               // LDS bytes per halo pixel
constexpr int WS_DBYTES = 128 * 1024;   // hand-off tile
constexpr int WS_RING = 6;              // weight ring depth in k-steps (16 input channels x 1 tap each); 18 k-steps per step
constexpr int WS_KSTEPS = 18;
constexpr int WS_NQ = 4;                // a finished tile is retired in WS_NQ parts, one per step of the next tile
constexpr int WS_QJ = 32 / WS_NQ;       // quads per lane and part

template <int LOGW>
struct WsGeom {
    static constexpr int W = 1 << LOGW;
    static constexpr int TW = 16, TH = 8;
    static constexpr int TCOLS = W / TW;
    static constexpr int TPI = (W / TH) * TCOLS;  // tiles (= statistics slots) per image; equals Geom<3, LOGW>::TPI
    static constexpr int HW_ = TW + 2, HH_ = TH + 2;
    static constexpr int HALO_PIX = HW_ * HH_;    // 180
    // Row stride: a multiple of 256 B.  ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
    // (MI355X_MICROARCH.md, LDS table): a group mixes columns {0-3,12-15} of one pixel row with columns {4-11} of the next, and
    // with the 80-byte pixel pitch their sixteen 16-byte slots tile the 256-byte LDS line exactly when the rows are congruent
    // mod 256 B.  (The unpadded 1440-byte stride cost 3.5 conflict cycles per LDS instruction, SQ_LDS_BANK_CONFLICT.)
    static constexpr int RS = ((HW_ * WS_PA + 255) / 256) * 256;  // 1536
    static constexpr int ABUF = HH_ * RS;                          // 15360
    static __host__ __device__ constexpr int off0(int p) { return ((p >> 4) & 7) * RS + (p & 15) * WS_PA; }
};

constexpr size_t ws_lds_bytes() { return (size_t)WS_DBYTES + 2 * (size_t)WsGeom<5>::ABUF; }

__device__ __forceinline__ void ws_barrier() {
    // LDS traffic of this wave retired, then the workgroup barrier.  Deliberately NOT __syncthreads(): its fence would
    // also drain vmcnt, i.e. the consumers' weight prefetch ring and the producers' stores, at every step.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// element offset of k-step j (tap = j >> 1, 16-channel half kk = j & 1) inside one 32-channel step of the packed weights
__host__ __device__ constexpr int ws_koff(int j) { return ((j >> 1) * 4 + (j & 1)) * 512; }

// ABL: compile-time ablation mask for scripts/conv_ablate.py (0 in production): 1 no staging, 2 no retire (drain),
// 4 no accumulator hand-off, 8 no weight refill, 16 no MFMA / A reads
template <int RES, int LOGW, int ABL = 0>
__global__ __launch_bounds__(WS_NTHR) void conv3_ws_kernel(const ConvArgs a, const int ntiles) {
    typedef __bf16 T;
    using G = WsGeom<LOGW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const dbuf = smem;
    char* const abuf0 = smem + WS_DBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = a.C1 + a.C2;
    const int nchunk = Cin / WS_KC;  // pipeline steps per tile (> WS_NQ and even, checked by the launcher)
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;  // >= 1: grid <= ntiles
    const int S = my_tiles * nchunk;
    const int H = a.H;

    if (wave < 4) {
        // =============================================== consumers ===============================================
        const int r = lane & 31, h = lane >> 5;
        const size_t wstride = (size_t)(Cin / 64) * 9 * 4 * 512;  // packed elements per 32-output-channel group
        const T* wp = reinterpret_cast<const T*>(a.wpack) + (size_t)(wave * 2) * wstride + lane * 8;
        auto wbase = [&](int c32) -> size_t { return ((size_t)(c32 >> 1) * 36 + (c32 & 1) * 2) * 512; };

        Frag8<T> bq[WS_RING][2];
#pragma unroll
        for (int j = 0; j < WS_RING; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bq[j][nt] = load_frag(wp + nt * wstride + ws_koff(j));

        f32x16 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

        const int lane_off = G::off0(r) + h * 16;
        // hand-off address of this lane's accumulator column: channel wave*64 + nt*32 + r, pixel rows with bit 2 == h
        int dl[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int ch = wave * 64 + nt * 32 + r;
            dl[nt] = h * 4096 + (((ch >> 2) ^ (h << 3)) << 4) + (ch & 3) * 4;
        }

        ws_barrier();  // step 0 is staged
        int c = 0;
        for (int s = 0; s < S; ++s) {
            const char* abase = abuf0 + (s & 1) * G::ABUF + lane_off;
            const int cn = (c + 1 == nchunk) ? 0 : c + 1;
            const T* wcur = wp + wbase(c);
            const T* wnxt = (s + 1 < S) ? wp + wbase(cn) : wcur;  // the very last refills re-read this step (never used)

            auto read_a = [&](int j, Frag8<T> (&af)[4]) {
                const int tap = j >> 1, kk = j & 1;
                const int off = (tap / 3) * G::RS + (tap % 3) * WS_PA + kk * 32;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    af[mt] = load_frag(reinterpret_cast<const T*>(abase + off + G::off0(mt * 32)));
            };
            auto mma8 = [&](int j, const Frag8<T> (&af)[4]) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        if (!(ABL & 16)) mma16(acc[mt][nt], af[mt], bq[j % WS_RING][nt]);
                if (ABL & 8) return;
                const T* pn = (j + WS_RING < WS_KSTEPS) ? wcur + ws_koff(j + WS_RING) : wnxt + ws_koff(j + WS_RING - WS_KSTEPS);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bq[j % WS_RING][nt] = load_frag(pn + nt * wstride);
            };
            Frag8<T> a0[4], a1[4];
            if (!(ABL & 16)) read_a(0, a0);
#pragma unroll
            for (int j = 0; j < WS_KSTEPS && !(ABL & 16); j += 2) {
                read_a(j + 1, a1);
                __builtin_amdgcn_sched_barrier(0);
                mma8(j, a0);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 2 < WS_KSTEPS) read_a(j + 2, a0);
                __builtin_amdgcn_sched_barrier(0);
                mma8(j + 1, a1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c + 1 == nchunk && !(ABL & 4)) {
                // tile finished: hand the fp32 accumulators to the producers and start the next tile from zero
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            *reinterpret_cast<float*>(dbuf + (mt * 32 + (i & 3) + 8 * (i >> 2)) * 1024 + dl[nt]) = acc[mt][nt][i];
                            acc[mt][nt][i] = 0.f;
                        }
            }
            c = cn;
            ws_barrier();
        }
        return;
    }

    // ================================================= producers =================================================
    const int ptid = tid - 256;
    const int oct = ptid & 3;  // 8-channel octet of the 32-channel step
    const T* src1 = reinterpret_cast<const T*>(a.src1);
    const T* src2 = reinterpret_cast<const T*>(a.src2);
    // this thread's (up to) three halo pixels: relative position and LDS offset are tile-independent
    int hdy[3], hdx[3], hlds[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int hq = min((ptid >> 2) + 64 * i, G::HALO_PIX - 1);
        const int hy = hq / G::HW_, hx = hq - hy * G::HW_;
        hdy[i] = hy - 1;
        hdx[i] = hx - 1;
        hlds[i] = hy * G::RS + hx * WS_PA + oct * 16;
    }
    const bool third = (ptid >> 2) + 128 < G::HALO_PIX;  // item 2 exists for 52 of the 64 pixel slots

    auto tile_coord = [&](int t, int& n, int& slot, int& row0, int& col0) {
        n = t / G::TPI;
        slot = t - n * G::TPI;
        row0 = (slot / G::TCOLS) * G::TH;
        col0 = (slot % G::TCOLS) * G::TW;
    };

    // operands of one staged step
    struct Staged {
        Frag8<T> raw[3];
        bool valid[3];
        float2 ab[8];
    };
    auto stage_load = [&](Staged& st, int t, int c32) __attribute__((always_inline)) {
        int n, slot, row0, col0;
        tile_coord(t, n, slot, row0, col0);
        const int c0 = c32 * WS_KC + oct * 8;
        const bool first = c0 < a.C1;
        const T* sp = first ? src1 + c0 : src2 + (c0 - a.C1);
        const int C = first ? a.C1 : a.C2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int y = row0 + hdy[i], x = col0 + hdx[i];
            st.valid[i] = (y >= 0) && (y < H) && (x >= 0) && (x < G::W) && (i < 2 || third);
            const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), G::W - 1);
            const int sy = (RES == RES_UP) ? (yc >> 1) : yc, sx = (RES == RES_UP) ? (xc >> 1) : xc;
            st.raw[i] = load_frag(sp + (((size_t)n * a.Hs + sy) * a.Ws + sx) * C);
        }
        const float2* p = a.ab + (size_t)n * Cin + c32 * WS_KC + oct * 8;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p + j);
            st.ab[j] = make_float2(q[0], q[1]);
            st.ab[j + 1] = make_float2(q[2], q[3]);
        }
    };
    auto stage_store = [&](const Staged& st, char* abuf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float v[8];
            widen8(st.raw[i], v);
            bf16x8 pk;
#pragma unroll
            for (int j = 0; j < 8; ++j) pk[j] = (__bf16)silu_f<true>(fmaf(v[j], st.ab[j].x, st.ab[j].y));
            // out-of-image halo pixels are zero: select on the four packed dwords, not on the eight floats
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            u32x4 w = __builtin_bit_cast(u32x4, pk);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = st.valid[i] ? w[j] : 0u;
            if (i < 2 || third) *reinterpret_cast<u32x4*>(abuf + hlds[i]) = w;
        }
    };

    // retire part `part` (128 / WS_NQ pixels) of tile t: this wave owns channel quads [16 pw, 16 pw + 16) of every pixel.
    // The residual quads and the per-image additive term are fetched one step ahead (retire_prefetch).
    const int pw = wave - 4;
    const int q = pw * 16 + (lane & 15), psub = lane >> 4;
    const int co = q * 4;
    T* const out = reinterpret_cast<T*>(a.out);
    const T* const resid = reinterpret_cast<const T*>(a.resid);
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + co);
    float ssum = 0.f, ssq = 0.f;
    typedef typename Raw4<T>::type R4;
    R4 rr[WS_QJ];
#pragma unroll
    for (int j = 0; j < WS_QJ; ++j) rr[j] = R4{};
    const bool has_resid = __builtin_amdgcn_readfirstlane(resid != nullptr);
    f32x4 radd = bias4;
    // element offset of this lane's quad of pixel p = part*(128/NQ) + j*4 + psub (row p >> 4, column p & 15)
    constexpr int ROWS_PER_PART = 8 / WS_NQ;
    auto tile_base = [&](int t, int part, int& n, int& slot) -> size_t {
        int row0, col0;
        tile_coord(t, n, slot, row0, col0);
        return (((size_t)n * H + row0 + part * ROWS_PER_PART) * G::W + col0 + psub) * 256 + co;
    };
    auto retire_prefetch = [&](int t, int part) __attribute__((always_inline)) {
        int n, slot;
        const size_t tb = tile_base(t, part, n, slot);
        radd = bias4;
        if (a.temb) radd += *reinterpret_cast<const f32x4*>(a.temb + (size_t)n * a.temb_stride + co);
        if (has_resid) {
#pragma unroll
            for (int j = 0; j < WS_QJ; ++j) rr[j] = raw_load4(resid + tb + ((j >> 2) * G::W + (j & 3) * 4) * 256);
        }
    };
    auto retire_part = [&](int t, int part) __attribute__((always_inline)) {
        int n, slot;
        const size_t tb = tile_base(t, part, n, slot);
        const char* dsrc = dbuf + (part * (128 / WS_NQ) + psub) * 1024;
        f32x4 dv[WS_QJ];
#pragma unroll
        for (int j = 0; j < WS_QJ; ++j) {
            dv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(ABL & 64)) dv[j] = *reinterpret_cast<const f32x4*>(dsrc + j * 4096 + ((q ^ ((j & 1) << 3)) << 4));  // bit 2 of p is j & 1
        }
#pragma unroll
        for (int j = 0; j < WS_QJ; ++j) {
            f32x4 v = dv[j] + radd;
            if (has_resid) v += widen4(rr[j]);
            v *= a.scale;
            f32x4 vr = v;
            if (!(ABL & 32)) vr = store4(out + tb + ((j >> 2) * G::W + (j & 3) * 4) * 256, v);  // as the next layer reads them
            ssum += (vr[0] + vr[1]) + (vr[2] + vr[3]);
            ssq += (vr[0] * vr[0] + vr[1] * vr[1]) + (vr[2] * vr[2] + vr[3] * vr[3]);
        }
        if (part == WS_NQ - 1) {
            float sv = ssum, qv = ssq;
            sv += __shfl_xor(sv, 16);
            qv += __shfl_xor(qv, 16);
            sv += __shfl_xor(sv, 32);
            qv += __shfl_xor(qv, 32);
            if (a.stats && psub == 0) a.stats[((size_t)n * G::TPI + slot) * 64 + q] = make_float2(sv, qv);
            ssum = ssq = 0.f;
        }
    };

    // ---- step machine ---------------------------------------------------------------------------------------------------
    // During step s (tile t_cur, chunk c) the consumers multiply halo buffer s & 1; the producers
    //   1. issue the global loads of step s+1,
    //   2. retire one part of the previous tile (steps 0..WS_NQ-1 of a tile; its residual was fetched during step s-1),
    //   3. transform + park step s+1 in the other halo buffer, and fetch the next retire's residual.
    // Fetching a whole step ahead instead (loads before the barrier, transform right after it) measured 4 % SLOWER end to end
    // (5343 vs 5574 img/s, same box): the wait for those loads then sits at the head of the step, in front of the retire
    // work that otherwise covers it.
    const int gstride = (int)gridDim.x;
    auto advance = [&](int& t, int& cc) {
        if (++cc == nchunk) {
            cc = 0;
            t += gstride;
        }
    };
    Staged st;
    int t_cur = blockIdx.x, c = 0;   // step s
    int t1 = t_cur, c1 = 0;          // step s+1
    stage_load(st, t1, c1);
    stage_store(st, abuf0);
    advance(t1, c1);
    ws_barrier();
    for (int s = 0; s < S; ++s) {
        const bool more = s + 1 < S && !(ABL & 1);
        const bool retiring = !(ABL & 2) && c < WS_NQ && s >= nchunk;
        if (more) stage_load(st, t1, c1);
        if (retiring) retire_part(t_cur - gstride, c);
        if (more) stage_store(st, abuf0 + ((s + 1) & 1) * G::ABUF);
        if (!(ABL & 2)) {
            if (retiring && c + 1 < WS_NQ) retire_prefetch(t_cur - gstride, c + 1);
            if (c + 1 == nchunk) retire_prefetch(t_cur, 0);  // its accumulators arrive with this step's barrier
        }
        advance(t_cur, c);
        advance(t1, c1);
        ws_barrier();
    }
    if (!(ABL & 2)) {
        const int t_last = (int)blockIdx.x + (my_tiles - 1) * gstride;
        for (int part = 0; part < WS_NQ; ++part) {
            if (part) retire_prefetch(t_last, part);
            retire_part(t_last, part);
        }
    }
}

int g_ws_cus = 0;

template <int RES, int LOGW, int ABL = 0>
int launch_ws_one(const ConvArgs& a, hipStream_t stream, bool prepare_only) {
    using G = WsGeom<LOGW>;
    auto kern = conv3_ws_kernel<RES, LOGW, ABL>;
    const size_t lds = ws_lds_bytes();
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (!g_ws_cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            return (int)hipErrorUnknown;
        g_ws_cus = n;
    }
    if (prepare_only) return 0;
    const int ntiles = a.B * G::TPI;
    const int grid = ntiles < g_ws_cus ? ntiles : g_ws_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WS_NTHR), lds, stream, a, ntiles);
    return (int)hipGetLastError();
}

}  // namespace

// bf16 3x3 GroupNorm+SiLU convs with 256 output channels at 32x32 / 16x16 (RES_NONE or RES_UP)
bool conv_ws_supported(int dtype, int ks, int pro, int res, int outmode, const ConvArgs& a) {
    const int cin = a.C1 + a.C2;
    return dtype == 1 && ks == 3 && pro == PRO_GN_SILU && (res == RES_NONE || res == RES_UP) && outmode == OUT_NHWC &&
           (a.W == 32 || a.W == 16) && a.H == a.W && a.Cout == 256 && (cin % 64) == 0 && cin / WS_KC > WS_NQ &&
           (a.C1 % WS_KC) == 0 && a.ab != nullptr;
}

int launch_conv_ws(int res, const ConvArgs& a, hipStream_t stream, bool prepare_only) {
    if (a.W == 32) return res == RES_UP ? launch_ws_one<RES_UP, 5>(a, stream, prepare_only) : launch_ws_one<RES_NONE, 5>(a, stream, prepare_only);
    return res == RES_UP ? launch_ws_one<RES_UP, 4>(a, stream, prepare_only) : launch_ws_one<RES_NONE, 4>(a, stream, prepare_only);
}

// ablation builds of the dominant shape (32x32, no resampling): scripts/conv_ablate.py only
int launch_conv_ws_debug(const ConvArgs& a, int abl, hipStream_t stream) {
    if (a.W != 32) return (int)hipErrorInvalidValue;
    switch (abl) {
        case 0: return launch_ws_one<RES_NONE, 5, 0>(a, stream, false);
        case 1: return launch_ws_one<RES_NONE, 5, 1>(a, stream, false);
        case 2: return launch_ws_one<RES_NONE, 5, 2>(a, stream, false);
        case 3: return launch_ws_one<RES_NONE, 5, 3>(a, stream, false);
        case 7: return launch_ws_one<RES_NONE, 5, 7>(a, stream, false);
        case 15: return launch_ws_one<RES_NONE, 5, 15>(a, stream, false);
        case 16: return launch_ws_one<RES_NONE, 5, 16>(a, stream, false);
        case 24: return launch_ws_one<RES_NONE, 5, 24>(a, stream, false);
        case 8: return launch_ws_one<RES_NONE, 5, 8>(a, stream, false);
        case 32: return launch_ws_one<RES_NONE, 5, 32>(a, stream, false);
        case 64: return launch_ws_one<RES_NONE, 5, 64>(a, stream, false);
        case 96: return launch_ws_one<RES_NONE, 5, 96>(a, stream, false);
    }
    return (int)hipErrorInvalidValue;
}
