// Wave-specialised, persistent variant of the fused GroupNorm+SiLU -> Conv2d 3x3 -> bias/temb/residual/scale (+ GroupNorm
// partial statistics) kernel of conv.hip, for the shapes that dominate the EDM U-Net: bf16, 3x3, 256 output channels,
// 32x32 and 16x16 outputs, with or without the nearest 2x up-sampling folded into the load.  Same arithmetic and same
// operands (NHWC bf16 activations, MFMA-fragment-packed weights, [B][slot][C/4] statistics) as conv_fused_kernel.
//
// Why a second kernel: conv_fused_kernel runs its staging prologue, MFMA loop and epilogue back to back in every wave
// and relies on a second resident workgroup to fill the gaps; measured (profiles/r01_conv_ablation*.txt) the two
// workgroups fall into lockstep and the phases add up (MFMA + weight stream 331 us, + staging 80, + epilogue 60..120).
// Here the phases run on DIFFERENT waves of one 512-thread workgroup (one per CU, persistent over pixel tiles):
//   waves 0-3  "consumers": nothing but ds_read(A) -> v_mfma <- global weights; 8 x 4 accumulator tiles of 16x16 each
//              (128 pixels x 64 output channels per wave), weights streamed from L2 through a 3-tap-deep register ring.
//   waves 4-7  "producers": stage the next step's (8+2) x (16+2) halo of 32 input channels into LDS (GroupNorm affine + SiLU
//              applied once per element), and retire the PREVIOUS tile: read its fp32 accumulators from the LDS hand-off
//              buffer, add bias / temb / residual, scale, round to bf16, store, and reduce the GroupNorm partial sums.
// One s_barrier per step (32 input channels x 9 taps = 288 MFMAs of 16 cycles per consumer wave).  Vector work of another
// wave is NOT hidden behind a wave's MFMAs on gfx950 (profiles/r01_micro_mfma_valu_overlap.txt); what the split buys is
// that no wave ever waits for memory with the matrix core idle.
//
// MFMA shape: v_mfma_f32_16x16x32_bf16.  Same FLOPs per cycle as 32x32x16, but on random operands the chip sustains a 12 %
// higher clock with it (profiles/r01_micro_mfma_peak_vs_data.txt: 1.96 vs 1.76 PFLOP/s bare; the kernel is power-limited).
//
// LDS (one workgroup owns the CU's 160 KB):  [0, 128 KB) fp32 hand-off tile [128 px][256 ch]; then two 12,288-byte halo
// buffers laid out as 4 planes (one per 8-channel group g) x 10 rows x 18 pixels x 16 B.  An A fragment of the 16x16x32 MFMA
// is lane (pixel column c = lane & 15, group g = lane >> 4); ds_read_b128 is served in the lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS table), i.e. columns {0-3,12-15} of plane g with columns {4-11} of plane
// g+1: with a 16-byte pixel pitch inside a plane and planes 3072 B (= 0 mod 256 B) apart, every group covers one 256-byte
// LDS line exactly, for every tap shift.
#include "common.h"
#include "conv.h"

namespace {

constexpr int WS_NTHR = 512;
constexpr int WS_KC = 32;               // input channels per pipeline step
constexpr int WS_PA = 16;               // LDS bytes per halo pixel inside one channel-group plane
constexpr int WS_DBYTES = 128 * 1024;   // hand-off tile
constexpr int WS_RING = 3;              // weight ring depth in taps (4 fragments of 16 output channels each)
constexpr int WS_HSTEPS = 18;           // half-taps per step: (tap, pixel rows 0-3 | 4-7), 16 MFMAs each
// A finished tile is retired in NQ parts, one per step of the next tile (template parameter): 4 normally, 2 for a 128-channel
// input whose tiles have only 4 steps (the parts have to be done before the tile's own accumulators arrive).

template <int LOGW>
struct WsGeom {
    static constexpr int W = 1 << LOGW;
    static constexpr int TW = 16, TH = 8;
    static constexpr int TCOLS = W / TW;
    static constexpr int TPI = (W / TH) * TCOLS;  // tiles (= statistics slots) per image; equals Geom<3, LOGW>::TPI
    static constexpr int HW_ = TW + 2, HH_ = TH + 2;
    static constexpr int HALO_PIX = HW_ * HH_;    // 180
    static constexpr int RS = HW_ * WS_PA;        // 288: row stride inside a plane
    static constexpr int PLANE = 3072;            // >= HH_ * RS = 2880, multiple of 256 B
    static constexpr int ABUF = 4 * PLANE;        // 12288
};

constexpr size_t ws_lds_bytes() { return (size_t)WS_DBYTES + 2 * (size_t)WsGeom<5>::ABUF; }

__device__ __forceinline__ void ws_barrier() {
    // LDS traffic of this wave retired, then the workgroup barrier.  Deliberately NOT __syncthreads(): its fence would
    // also drain vmcnt, i.e. the consumers' weight prefetch ring and the producers' stores, at every step.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// TRANSPOSED product (A operand = weights [16 ch x 32 k], B operand = activations [32 k x 16 px]; both operands have the same lane
// layout, so this is just the argument order): a lane ends with 4 CONSECUTIVE channels of one pixel and the hand-off is 32
// ds_write_b128 per lane and tile instead of 128 ds_write_b32 (the structure proven in conv_ws3.hip)
__device__ __forceinline__ void mma32(f32x4& acc, const Frag8<__bf16>& a, const Frag8<__bf16>& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.v, a.v, acc, 0, 0, 0);
}

// ABL: compile-time ablation mask for scripts/conv_ablate.py (0 in production): 1 no staging, 2 no retire (drain),
// 4 no accumulator hand-off, 8 no weight refill, 16 no MFMA / A reads
// PRO: PRO_GN_SILU, or PRO_NONE — the operand is staged as it lies (data gradients: the same conv on dY with the transposed,
// flipped weights; tangent convolutions of the forward-mode pass)
template <int RES, int LOGW, int ABL = 0, int WS_NQ = 4, int PRO = PRO_GN_SILU>
__global__ __launch_bounds__(WS_NTHR) void conv3_ws_kernel(const ConvArgs a, const int ntiles) {
    constexpr int WS_QJ = 32 / WS_NQ;  // quads per lane and part
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
    // Workgroup b runs on XCD b % 8 (round-robin dispatch).  Give each XCD a contiguous run of tiles per sweep — whole images
    // — so that the halo rows neighbouring tiles share, and an image's residual / statistics lines, meet in one L2.
    const int G8 = (int)gridDim.x >> 3;
    const int wg = ((gridDim.x & 7) == 0) ? ((int)blockIdx.x & 7) * G8 + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int my_tiles = (ntiles - wg + (int)gridDim.x - 1) / (int)gridDim.x;  // >= 1: grid <= ntiles
    const int S = my_tiles * nchunk;
    const int H = a.H;

    if (wave < 4) {
        // =============================================== consumers ===============================================
        const int col = lane & 15, g = lane >> 4;
        // weights packed for the 16x16x32 B operand: [cout/16][step][tap][lane][8]  (pack_conv_weights_ws_kernel)
        const size_t wstride = (size_t)nchunk * 9 * 512;  // elements per 16-output-channel group
        // Weight fragments come through a buffer resource over this wave's 64 output channels: wave-uniform byte offsets
        // travel in SGPRs (soffset), the lane part is one constant VGPR, so a load costs no vector address arithmetic.
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(reinterpret_cast<const T*>(a.wpack_ws) + (size_t)(wave * 4) * wstride), 0, 0x7fffffff, 0x00020000);
        const int wvoff = lane * 16;
        const int wsb = (int)(wstride * 2);  // bytes between 16-channel groups
        auto load_w = [&](int byte_off) -> Frag8<T> {
            Frag8<T> f;
            f.v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, byte_off, 0));
            return f;
        };

        Frag8<T> bq[WS_RING][4];
#pragma unroll
        for (int j = 0; j < WS_RING; ++j)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bq[j][nt] = load_w(nt * wsb + j * 1024);

        f32x4 acc[8][4];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

        const int lane_off = g * G::PLANE + col * WS_PA;

        ws_barrier();  // step 0 is staged
        int c = 0;
        for (int s = 0; s < S; ++s) {
            const char* abase = abuf0 + (s & 1) * G::ABUF + lane_off;
            const int cn = (c + 1 == nchunk) ? 0 : c + 1;
            const int wcur = c * (9 * 1024);                          // byte offset of this step's taps
            const int wnxt = (s + 1 < S) ? cn * (9 * 1024) : wcur;  // the very last refills re-read this step (never used)

            // half-tap j: tap j >> 1, pixel rows 4 (j & 1) .. +3
            auto read_a = [&](int j, Frag8<T> (&af)[4]) {
                const int tap = j >> 1, half = j & 1;
                const int off = (tap / 3 + half * 4) * G::RS + (tap % 3) * WS_PA;
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = load_frag(reinterpret_cast<const T*>(abase + off + m * G::RS));
            };
            auto mma16x = [&](int j, const Frag8<T> (&af)[4]) {
                const int tap = j >> 1, half = j & 1;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        if (!(ABL & 16)) mma32(acc[half * 4 + m][nt], af[m], bq[tap % WS_RING][nt]);
                if ((ABL & 8) || !half) return;
                const int pn = (tap + WS_RING < 9) ? wcur + (tap + WS_RING) * 1024 : wnxt + (tap + WS_RING - 9) * 1024;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bq[tap % WS_RING][nt] = load_w(pn + nt * wsb);
            };
            Frag8<T> a0[4], a1[4];
            if (!(ABL & 16)) read_a(0, a0);
#pragma unroll
            for (int j = 0; j < WS_HSTEPS && !(ABL & 16); j += 2) {
                read_a(j + 1, a1);
                __builtin_amdgcn_sched_barrier(0);
                mma16x(j, a0);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 2 < WS_HSTEPS) read_a(j + 2, a0);
                __builtin_amdgcn_sched_barrier(0);
                mma16x(j + 1, a1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c + 1 == nchunk && !(ABL & 4)) {
                // tile finished: hand the fp32 accumulators to the producers and start the next tile from zero
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        // this lane's quad: pixel (row mt, column col), channels wave*64 + nt*16 + 4 g .. +3 = 16-byte slot
                        // wave*16 + nt*4 + g of the pixel's 1 KiB row, XOR-ed with col & 7: the 8 lanes of a ds_write_b128 group
                        // (8 pixels, 1 KiB apart) land on 8 different slots; the producers read slot q of pixel p at q ^ (p & 7)
                        const int slot = ((wave * 16 + nt * 4 + g) ^ (col & 7)) << 4;
                        *reinterpret_cast<f32x4*>(dbuf + mt * 16384 + col * 1024 + slot) = acc[mt][nt];
                        acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
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
        hlds[i] = oct * G::PLANE + hy * G::RS + hx * WS_PA;
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
        float2 ab[8];
    };
    // Per-tile addressing of this thread's three halo pixels, refreshed when the staged step enters a new tile: 32-bit byte
    // offsets into either source tensor (a whole 32-channel step lies in one of them, C1 % 32 == 0) on top of a wave-uniform
    // base pointer, so that a step's loads cost no vector address arithmetic.
    unsigned off1[3], off2[3];
    bool valid[3];
    const float2* abn = a.ab;
    auto tile_setup = [&](int t) __attribute__((always_inline)) {
        int n, slot, row0, col0;
        tile_coord(t, n, slot, row0, col0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int y = row0 + hdy[i], x = col0 + hdx[i];
            valid[i] = (y >= 0) && (y < H) && (x >= 0) && (x < G::W) && (i < 2 || third);
            const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), G::W - 1);
            const int sy = (RES == RES_UP) ? (yc >> 1) : yc, sx = (RES == RES_UP) ? (xc >> 1) : xc;
            const unsigned pix = (unsigned)((n * a.Hs + sy) * a.Ws + sx);
            off1[i] = (pix * (unsigned)a.C1 + oct * 8) * 2u;
            off2[i] = (pix * (unsigned)a.C2 + oct * 8) * 2u;
        }
        abn = a.ab + (size_t)n * Cin;
    };
    auto stage_load = [&](Staged& st, int t, int c32) __attribute__((always_inline)) {
        if (c32 == 0) tile_setup(t);
        const bool first = c32 * WS_KC < a.C1;  // wave-uniform
        const char* base = first ? reinterpret_cast<const char*>(src1) + c32 * (WS_KC * 2)
                                 : reinterpret_cast<const char*>(src2) + (c32 * WS_KC - a.C1) * 2;
#pragma unroll
        for (int i = 0; i < 3; ++i) st.raw[i] = load_frag(reinterpret_cast<const T*>(base + (first ? off1[i] : off2[i])));
        if (PRO == PRO_NONE) return;
        const float2* p = abn + c32 * WS_KC + oct * 8;
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
            bf16x8 pk = st.raw[i].v;
            if (PRO != PRO_NONE) {
                float v[8];
                widen8(st.raw[i], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) pk[j] = (__bf16)silu_f<true>(fmaf(v[j], st.ab[j].x, st.ab[j].y));
            }
            // out-of-image halo pixels are zero: select on the four packed dwords, not on the eight floats
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            u32x4 w = __builtin_bit_cast(u32x4, pk);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = valid[i] ? w[j] : 0u;
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
    // pixel p = part*(128/NQ) + j*4 + psub of the tile (row p >> 4, column p & 15): a wave-uniform tile/part base plus this
    // lane's 32-bit element offset plus a compile-time term per j
    constexpr int ROWS_PER_PART = 8 / WS_NQ;
    const unsigned loff = (unsigned)(psub * 256 + co);
    auto tile_base = [&](int t, int part, int& n, int& slot) -> size_t {
        int row0, col0;
        tile_coord(t, n, slot, row0, col0);
        return (((size_t)n * H + row0 + part * ROWS_PER_PART) * G::W + col0) * 256;
    };
    auto retire_prefetch = [&](int t, int part) __attribute__((always_inline)) {
        int n, slot;
        const size_t tb = tile_base(t, part, n, slot);
        radd = bias4;
        if (a.temb) radd += *reinterpret_cast<const f32x4*>(a.temb + (size_t)n * a.temb_stride + co);
        radd *= a.scale;  // (acc + add + resid) * scale evaluated as fma(acc, scale, add * scale) [+ fma(resid, scale, .)]
        if (has_resid) {
            const T* rb = resid + tb;
#pragma unroll
            for (int j = 0; j < WS_QJ; ++j) rr[j] = raw_load4(rb + (loff + (unsigned)(((j >> 2) * G::W + (j & 3) * 4) * 256)));
        }
    };
    auto retire_part = [&](int t, int part) __attribute__((always_inline)) {
        int n, slot;
        const size_t tb = tile_base(t, part, n, slot);
        T* ob = out + tb;
        const char* dsrc = dbuf + (part * (128 / WS_NQ) + psub) * 1024;
        f32x4 dv[WS_QJ];
#pragma unroll
        for (int j = 0; j < WS_QJ; ++j) {
            dv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // pixel p = part * (128 / NQ) + 4 j + psub: its slots are XOR-ed with p & 7 = ((j & 1) << 2) + psub (128 / NQ is a multiple of 8)
            if (!(ABL & 64)) dv[j] = *reinterpret_cast<const f32x4*>(dsrc + j * 4096 + ((q ^ (((j & 1) << 2) + psub)) << 4));
        }
#pragma unroll
        for (int j = 0; j < WS_QJ; ++j) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(dv[j][e], a.scale, radd[e]);
            if (has_resid) {
                const f32x4 rw = widen4(rr[j]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(rw[e], a.scale, v[e]);
            }
            if (!(ABL & 32)) store4(ob + (loff + (unsigned)(((j >> 2) * G::W + (j & 3) * 4) * 256)), v);
            // GroupNorm partial sums from the fp32 values: the bf16 rounding of the stored tensor is zero-mean noise of 2^-9
            // relative size per element, far below the statistics' own resolution over >= 8192 elements per group
            ssum += (v[0] + v[1]) + (v[2] + v[3]);
            ssq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
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
    int t_cur = wg, c = 0;   // step s
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
        const int t_last = wg + (my_tiles - 1) * gstride;
        for (int part = 0; part < WS_NQ; ++part) {
            if (part) retire_prefetch(t_last, part);
            retire_part(t_last, part);
        }
    }
}

// packed[n16][step][tap][lane][j] = W[cout = n16*16 + (lane & 15)][cin = step*32 + 8*(lane >> 4) + j][tap]: the B operand of
// v_mfma_f32_16x16x32_bf16, one coalesced 16-byte load per lane and fragment
__global__ void pack_conv_weights_ws_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int cout, int cin) {
    const size_t total = (size_t)cout * cin * 9;
    const int nstep = cin / WS_KC;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t t = idx;
        const int j = t % 8; t /= 8;
        const int lane = t % 64; t /= 64;
        const int tap = t % 9; t /= 9;
        const int step = t % nstep; t /= nstep;
        const int n16 = (int)t;
        const int co = n16 * 16 + (lane & 15);
        const int ci = step * WS_KC + 8 * (lane >> 4) + j;
        out[idx] = (__bf16)w[((size_t)co * cin + ci) * 9 + tap];
    }
}

int g_ws_cus[16] = {};  // CU count per device

template <int RES, int LOGW, int ABL = 0, int NQ = 4, int PRO = PRO_GN_SILU>
int launch_ws_one(const ConvArgs& a, hipStream_t stream, bool prepare_only) {
    using G = WsGeom<LOGW>;
    auto kern = conv3_ws_kernel<RES, LOGW, ABL, NQ, PRO>;
    const size_t lds = ws_lds_bytes();
    static bool attr_done[16] = {};
    const int dev = fg_device_slot();
    if (dev < 0) return (int)hipErrorInvalidDevice;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done[dev] = true;
    }
    if (!g_ws_cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return (int)hipErrorUnknown;
        g_ws_cus[dev] = n;
    }
    if (prepare_only) return 0;
    const int ntiles = a.B * G::TPI;
    const int grid = ntiles < g_ws_cus[dev] ? ntiles : g_ws_cus[dev];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WS_NTHR), lds, stream, a, ntiles);
    return (int)hipGetLastError();
}

}  // namespace

// bf16 3x3 GroupNorm+SiLU convs with 256 output channels at 32x32 / 16x16 (RES_NONE or RES_UP)
bool conv_ws_supported(int dtype, int ks, int pro, int res, int outmode, const ConvArgs& a) {
    const int cin = a.C1 + a.C2;
    // the staged loads use 32-bit byte offsets into each source tensor
    const size_t src_bytes = (size_t)a.B * a.Hs * a.Ws * (size_t)(a.C1 > a.C2 ? a.C1 : a.C2) * 2;
    if (src_bytes >= (1ull << 32)) return false;
    if (pro == PRO_NONE)  // operand taken as it lies: no resampling, at least 8 steps per tile
        return dtype == 1 && ks == 3 && res == RES_NONE && outmode == OUT_NHWC && (a.W == 32 || a.W == 16) && a.H == a.W &&
               a.Hs == a.H && a.Ws == a.W && a.Cout == 256 && (cin % 64) == 0 && cin / WS_KC >= 8 && (a.C1 % WS_KC) == 0 &&
               a.wpack_ws != nullptr;
    return dtype == 1 && ks == 3 && pro == PRO_GN_SILU && (res == RES_NONE || res == RES_UP) && outmode == OUT_NHWC &&
           (a.W == 32 || a.W == 16) && a.H == a.W && a.Cout == 256 && (cin % 64) == 0 && cin / WS_KC >= 4 &&
           (a.C1 % WS_KC) == 0 && a.ab != nullptr && a.wpack_ws != nullptr;
}

int launch_conv_ws(int res, const ConvArgs& a, hipStream_t stream, bool prepare_only, int pro) {
    if (pro == PRO_NONE) {
        if (res != RES_NONE || (a.C1 + a.C2) / WS_KC < 8) return (int)hipErrorInvalidValue;
        return a.W == 32 ? launch_ws_one<RES_NONE, 5, 0, 4, PRO_NONE>(a, stream, prepare_only)
                         : launch_ws_one<RES_NONE, 4, 0, 4, PRO_NONE>(a, stream, prepare_only);
    }
    if ((a.C1 + a.C2) / WS_KC == 4 || (prepare_only && a.C1 == -128)) {
        // 128 input channels: four steps per tile, retire in two parts (the U-Net's first block, 32x32, no resampling)
        if (res != RES_NONE || a.W != 32) return (int)hipErrorInvalidValue;
        return launch_ws_one<RES_NONE, 5, 0, 2>(a, stream, prepare_only);
    }
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

// whether a 3x3 conv with these dimensions can take the wave-specialised kernel (decides if its weights are also packed for it)
bool conv_ws_shape_ok(int dtype, int cout, int cin, int res) {
    return dtype == 1 && cout == 256 && (res == 32 || res == 16) && (cin % 64) == 0 && cin / WS_KC >= 4;
}

int launch_pack_conv_weights_ws(const float* w_oihw, void* wpack_ws, int cout, int cin, hipStream_t stream) {
    const size_t total = (size_t)cout * cin * 9;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv_weights_ws_kernel, dim3(grid), dim3(256), 0, stream, w_oihw, (__bf16*)wpack_ws, cout, cin);
    return (int)hipGetLastError();
}
