// Wave-specialised, persistent 3x3 convolution for the split-bf16 ("bf16x3", FG_DTYPE_BF16X3) compute mode: fp32 tensors in
// memory, every product a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (common.h).
// Same operation as conv_fused_kernel<bf16x3, 3, ...> (GroupNorm-apply + SiLU -> conv -> bias / temb / residual / scale + the
// GroupNorm partial statistics of the result; reference fastgen/networks/EDM/network.py:93-126, 274-299), for 256 output
// channels at 32x32 / 16x16 with or without the nearest 2x up-sampling folded into the load.
//
// Why not conv_fused_kernel: measured there (profiles/r02_x3a_*), the matrix pipe is 76 % busy but the chip holds only
// 1.72 GHz under the 32x32x16 shape; the 16x16x32 shape sustains a higher clock on real data (profiles/r01_micro_mfma_peak_vs_data.txt),
// and what is left of the idle time is the serial prologue / epilogue of every workgroup.
//
// One 768-thread workgroup per CU, persistent over 8x16-pixel tiles (three waves per SIMD, 168 registers each):
//   waves 0-3  consumer group A: output channels   0..127 (32 per wave), accumulators 8 pixel rows x 2 x (16 ch x 16 px)
//   waves 4-7  consumer group B: output channels 128..255, running TWO pipeline steps behind group A
//   waves 8-11 producers: stage the next step's (8+2) x (16+2) halo of 32 input channels into LDS — GroupNorm affine + SiLU once
//              per element, then the hi / lo split — and retire finished half tiles: fp32 accumulators from the LDS hand-off,
//              bias / temb / residual / scale, store, GroupNorm partial sums.
// Consumers do nothing but ds_read_b128 (activations) -> v_mfma <- weights streamed from L2 through a 3-tap register ring.
// The MFMA runs transposed (A = weights [16 ch x 32 k], B = activations [32 k x 16 px]): a lane then holds 4 CONSECUTIVE channels
// of one pixel, so the hand-off is 16 ds_write_b128 per lane and tile instead of 64 ds_write_b32.
//
// LDS = all 160 KiB: [0, 64 KiB) hand-off of ONE half tile [128 px][128 ch] fp32 (16-byte slots XOR-swizzled by pixel & 7:
// conflict-free for the consumers' stores and the producers' loads), then a ring of FOUR halo buffers of 24 KiB (planes 0-3:
// hi of the four 8-channel groups, planes 4-7: lo; 16-byte pixel pitch inside a plane, planes 3072 B = 0 mod 256 B apart: every
// ds_read_b128 lane group covers one 256-byte LDS line, for every tap shift - the layout of conv_ws.hip).
// The two-step skew between the consumer groups is what lets ONE 64 KiB hand-off serve both: group A dumps at the end of its
// tile's last step s, the producers retire that half during step s+1, group B dumps at the end of s+2, retired during s+3, and
// A's next dump comes at s + (steps per tile >= 4).  Every hand-off and every halo buffer changes hands across the ONE
// s_barrier that ends each step; no flags, no atomics.  Cost of the skew: two idle steps per launch and group.
#include "common.h"
#include "conv.h"

namespace {

constexpr int X3_NTHR = 768;
constexpr int X3_KC = 32;                  // input channels per pipeline step
constexpr int X3_PA = 16;                  // LDS bytes per halo pixel inside one plane
constexpr int X3_PLANE = 3072;             // >= 10 * 18 * 16 = 2880, multiple of 256 B
constexpr int X3_ABUF = 8 * X3_PLANE;      // 24576
constexpr int X3_RING = 4;
constexpr int X3_HOFF = 128 * 512;         // 65536
constexpr int X3_LDS = X3_HOFF + X3_RING * X3_ABUF;  // 163840 = the CU's whole LDS
constexpr int X3_STEPB = 9 * 2048;         // bytes of packed weights per step and 16-channel group: [tap][part][lane][8] bf16

template <int LOGW>
struct X3Geom {
    static constexpr int W = 1 << LOGW;
    static constexpr int TW = 16, TH = 8;
    static constexpr int TCOLS = W / TW;
    static constexpr int TPI = (W / TH) * TCOLS;  // tiles per image
    static constexpr int HW_ = TW + 2, HH_ = TH + 2;
    static constexpr int HALO_PIX = HW_ * HH_;    // 180
    static constexpr int RS = HW_ * X3_PA;        // 288
};

__device__ __forceinline__ void x3_barrier() {
    // this wave's LDS traffic retired, then the workgroup barrier; vmcnt is NOT drained (weight ring, stores stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct X3Frag {
    bf16x8 hi, lo;
};

template <int RES, int LOGW, int PRO>
__global__ __launch_bounds__(X3_NTHR) void conv3_x3ws_kernel(const ConvArgs a, const int ntiles) {
    using G = X3Geom<LOGW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hbuf = smem;
    char* const ring = smem + X3_HOFF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = a.C1 + a.C2;
    const int nchunk = Cin / X3_KC;  // steps per tile (>= 4, checked by the launcher)
    // workgroup b runs on XCD b % 8: contiguous runs of tiles (whole images) per XCD, as in conv_ws.hip
    const int G8 = (int)gridDim.x >> 3;
    const int wg = ((gridDim.x & 7) == 0) ? ((int)blockIdx.x & 7) * G8 + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int gstride = (int)gridDim.x;
    const int my_tiles = (ntiles - wg + gstride - 1) / gstride;  // >= 1: grid <= ntiles
    const int S = my_tiles * nchunk;
    const int NSTEP = S + 3;  // group B finishes at global step S + 1, its last half tile is retired during S + 2
    const int H = a.H;

    if (wave < 8) {
        // ================================================= consumers =================================================
        const int grp = wave >> 2, wl = wave & 3;
        const int col = lane & 15, g4 = lane >> 4;
        // packed weights [cout/16][step][tap][part][lane][8] (pack_conv_weights_x3ws_kernel); this wave: groups 8 grp + 2 wl, +1
        const size_t wstride = (size_t)nchunk * 9 * 1024;  // bf16 elements per 16-channel group
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<__bf16*>(reinterpret_cast<const __bf16*>(a.wpack_ws) + (size_t)(grp * 8 + wl * 2) * wstride), 0, 0x7fffffff,
            0x00020000);
        const int wvoff = lane * 16;
        const int wsb = (int)(wstride * 2);
        auto load_w = [&](int byte_off) -> X3Frag {
            X3Frag f;
            f.hi = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, byte_off, 0));
            f.lo = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, byte_off + 1024, 0));
            return f;
        };
        X3Frag wq[3][2];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wq[j][nt] = load_w(nt * wsb + j * 2048);

        f32x4 acc[8][2];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

        const int lane_off = g4 * X3_PLANE + col * X3_PA;
        // hand-off slot of this lane's channel quad (8 wl + 4 nt + g4 of the half's 32) for pixel (row m, column col)
        const int dl0 = col * 512 + ((8 * wl + ((g4) ^ (col & 7))) << 4);
        const int dl1 = col * 512 + ((8 * wl + ((4 + g4) ^ (col & 7))) << 4);

        x3_barrier();  // step 0 is staged
        int c = 0;
        for (int s = 0; s < NSTEP; ++s) {
            const int ls = s - 2 * grp;  // this group's step
            if (ls >= 0 && ls < S) {
                const char* abase = ring + (ls & 3) * X3_ABUF + lane_off;
                const int cn = (c + 1 == nchunk) ? 0 : c + 1;
                const int wcur = c * X3_STEPB;
                const int wnxt = (ls + 1 < S) ? cn * X3_STEPB : wcur;  // the very last refills re-read this step (never used)
                // quarter-tap q: tap q >> 2, pixel rows 2 (q & 3), +1
                auto read_x = [&](int q, X3Frag (&xf)[2]) {
                    const int tap = q >> 2, m0 = 2 * (q & 3);
                    const int off = (tap / 3 + m0) * G::RS + (tap % 3) * X3_PA;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        xf[m].hi = *reinterpret_cast<const bf16x8*>(abase + off + m * G::RS);
                        xf[m].lo = *reinterpret_cast<const bf16x8*>(abase + off + m * G::RS + 4 * X3_PLANE);
                    }
                };
                auto mma12 = [&](int q, const X3Frag (&xf)[2]) {
                    const int tap = q >> 2, m0 = 2 * (q & 3);
                    const X3Frag(&wf)[2] = wq[tap % 3];
                    // small terms first; every accumulator is touched once per group of four MFMAs
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[m0 + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt].hi, xf[m].lo, acc[m0 + m][nt], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[m0 + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt].lo, xf[m].hi, acc[m0 + m][nt], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[m0 + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt].hi, xf[m].hi, acc[m0 + m][nt], 0, 0, 0);
                    if ((q & 3) != 3) return;
                    // the tap's fragments are dead: refill their ring slot three taps ahead
                    const int pn = (tap + 3 < 9) ? wcur + (tap + 3) * 2048 : wnxt + (tap + 3 - 9) * 2048;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) wq[tap % 3][nt] = load_w(pn + nt * wsb);
                };
                X3Frag x0[2], x1[2];
                read_x(0, x0);
#pragma unroll
                for (int q = 0; q < 36; q += 2) {
                    read_x(q + 1, x1);
                    __builtin_amdgcn_sched_barrier(0);
                    mma12(q, x0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q + 2 < 36) read_x(q + 2, x0);
                    __builtin_amdgcn_sched_barrier(0);
                    mma12(q + 1, x1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (c + 1 == nchunk) {
                    // tile finished: hand this group's half of the fp32 accumulators to the producers, restart from zero
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        *reinterpret_cast<f32x4*>(hbuf + m * 8192 + dl0) = acc[m][0];
                        *reinterpret_cast<f32x4*>(hbuf + m * 8192 + dl1) = acc[m][1];
                        acc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                c = cn;
            }
            x3_barrier();
        }
        return;
    }

    // ================================================= producers =================================================
    const int ptid = tid - 512;
    const int pw = wave - 8;
    const int oct = ptid & 3;  // 8-channel group of the 32-channel step
    const float* src1 = reinterpret_cast<const float*>(a.src1);
    const float* src2 = reinterpret_cast<const float*>(a.src2);
    int hdy[3], hdx[3], hlds[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int hq = min((ptid >> 2) + 64 * i, G::HALO_PIX - 1);
        const int hy = hq / G::HW_, hx = hq - hy * G::HW_;
        hdy[i] = hy - 1;
        hdx[i] = hx - 1;
        hlds[i] = oct * X3_PLANE + hy * G::RS + hx * X3_PA;
    }
    const bool third = (ptid >> 2) + 128 < G::HALO_PIX;  // item 2 exists for 52 of the 64 pixel slots

    auto tile_coord = [&](int t, int& n, int& slot, int& row0, int& col0) {
        n = t / G::TPI;
        slot = t - n * G::TPI;
        row0 = (slot / G::TCOLS) * G::TH;
        col0 = (slot % G::TCOLS) * G::TW;
    };

    struct Staged {
        f32x4 raw[3][2];
        float2 ab[8];
    };
    unsigned off1[3], off2[3];  // byte offsets of this thread's halo pixels into either source tensor (< 4 GiB, checked by the launcher)
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
            off1[i] = (pix * (unsigned)a.C1 + oct * 8) * 4u;
            off2[i] = (pix * (unsigned)a.C2 + oct * 8) * 4u;
        }
        abn = a.ab + (size_t)n * Cin;
    };
    auto stage_load = [&](Staged& st, int t, int c32) __attribute__((always_inline)) {
        if (c32 == 0) tile_setup(t);
        const bool first = c32 * X3_KC < a.C1;  // wave-uniform: a whole step lies in one source (C1 % 32 == 0)
        const char* base = first ? reinterpret_cast<const char*>(src1) + c32 * (X3_KC * 4)
                                 : reinterpret_cast<const char*>(src2) + (c32 * X3_KC - a.C1) * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const char* p = base + (first ? off1[i] : off2[i]);
            st.raw[i][0] = *reinterpret_cast<const f32x4*>(p);
            st.raw[i][1] = *reinterpret_cast<const f32x4*>(p + 16);
        }
        if (PRO == PRO_NONE) return;
        const float2* p = abn + c32 * X3_KC + oct * 8;
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
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = st.raw[i][j >> 2][j & 3];
                if (PRO != PRO_NONE) v[j] = silu_f<true>(fmaf(v[j], st.ab[j].x, st.ab[j].y));
            }
            bf16x8 hi, lo;
            split8(v, hi, lo);
            // out-of-image halo pixels are zero: select on the packed dwords
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            u32x4 wh = __builtin_bit_cast(u32x4, hi), wl_ = __builtin_bit_cast(u32x4, lo);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wh[j] = valid[i] ? wh[j] : 0u;
                wl_[j] = valid[i] ? wl_[j] : 0u;
            }
            if (i < 2 || third) {
                *reinterpret_cast<u32x4*>(abuf + hlds[i]) = wh;
                *reinterpret_cast<u32x4*>(abuf + hlds[i] + 4 * X3_PLANE) = wl_;
            }
        }
    };

    // retire one half (128 channels) of tile t from the hand-off: this wave owns pixels [32 pw, 32 pw + 32), lanes 0-31 / 32-63 the
    // 32 channel quads of two neighbouring pixels per pass, 16 passes in two groups of 8 (loads of a group in flight together)
    const int chq = lane & 31, psub = lane >> 5;
    float* const out = reinterpret_cast<float*>(a.out);
    const float* const resid = reinterpret_cast<const float*>(a.resid);
    const bool has_resid = __builtin_amdgcn_readfirstlane(resid != nullptr);
    auto retire_half = [&](int t, int half) __attribute__((always_inline)) {
        int n, slot, row0, col0;
        tile_coord(t, n, slot, row0, col0);
        const int co = half * 128 + chq * 4;
        f32x4 radd = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) radd = *reinterpret_cast<const f32x4*>(a.bias + co);
        if (a.temb) radd += *reinterpret_cast<const f32x4*>(a.temb + (size_t)n * a.temb_stride + co);
        radd *= a.scale;  // (acc + add + resid) * scale evaluated as fma(acc, scale, add * scale) [+ fma(resid, scale, .)]
        const size_t tb = (((size_t)n * H + row0) * G::W + col0) * 256 + co;
        float ssum = 0.f, ssq = 0.f;
#pragma unroll 1
        for (int grp8 = 0; grp8 < 2; ++grp8) {
            f32x4 rr[8], dv[8];
            unsigned go[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int px = pw * 32 + 2 * (grp8 * 8 + j) + psub;
                go[j] = (unsigned)(((px >> 4) * G::W + (px & 15)) * 256);
                rr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (has_resid) rr[j] = *reinterpret_cast<const f32x4*>(resid + tb + go[j]);
                dv[j] = *reinterpret_cast<const f32x4*>(hbuf + px * 512 + ((chq ^ (px & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(dv[j][e], a.scale, radd[e]);
                if (has_resid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(rr[j][e], a.scale, v[e]);
                }
                *reinterpret_cast<f32x4*>(out + tb + go[j]) = v;
                ssum += (v[0] + v[1]) + (v[2] + v[3]);
                ssq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
        }
        ssum += __shfl_xor(ssum, 32);
        ssq += __shfl_xor(ssq, 32);
        // one statistics slot per (tile, producer wave): [n][TPI * 4][64 quads]
        if (a.stats && psub == 0) a.stats[(((size_t)n * G::TPI + slot) * 4 + pw) * 64 + half * 32 + chq] = make_float2(ssum, ssq);
    };

    // ---- step machine: during global step s the producers (1) issue the loads of step s+1, (2) retire the half tile a consumer
    // group dumped at the end of step s-1, (3) transform + park step s+1 in ring[(s+1) & 3] -----------------------------------
    auto advance = [&](int& t, int& cc) {
        if (++cc == nchunk) {
            cc = 0;
            t += gstride;
        }
    };
    Staged st;
    int t1 = wg, c1 = 0;  // the step being staged
    stage_load(st, t1, c1);
    stage_store(st, ring);
    advance(t1, c1);
    x3_barrier();
    int sa = 0;  // s % nchunk, tracked incrementally
    for (int s = 0; s < NSTEP; ++s) {
        const bool more = s + 1 < S;
        if (more) stage_load(st, t1, c1);
        // group A finished tile k = s / nchunk - 1 at the end of step s - 1 when s % nchunk == 0, nchunk <= s <= S
        if (sa == 0 && s >= nchunk && s <= S) retire_half(wg + (s / nchunk - 1) * gstride, 0);
        // group B runs two steps behind: its half of tile k was dumped at the end of step k * nchunk + nchunk + 1
        if (sa == 2 % nchunk && s - 2 >= nchunk && s - 2 <= S && (s - 2) % nchunk == 0) retire_half(wg + ((s - 2) / nchunk - 1) * gstride, 1);
        if (more) {
            stage_store(st, ring + ((s + 1) & 3) * X3_ABUF);
            advance(t1, c1);
        }
        if (++sa == nchunk) sa = 0;
        x3_barrier();
    }
}

// packed[n16][step][tap][part][lane][j], part 0 = bf16(w), part 1 = bf16(w - hi),
// w = W[cout = n16*16 + (lane & 15)][cin = step*32 + 8*(lane >> 4) + j][tap]: the 16-row x 32-k operand of v_mfma_f32_16x16x32_bf16
__global__ void pack_conv_weights_x3ws_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int cout, int cin) {
    const size_t total = (size_t)cout * cin * 9;
    const int nstep = cin / X3_KC;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t t = idx;
        const int j = t % 8; t /= 8;
        const int lane = t % 64; t /= 64;
        const int tap = t % 9; t /= 9;
        const int step = t % nstep; t /= nstep;
        const int n16 = (int)t;
        const int co = n16 * 16 + (lane & 15);
        const int ci = step * X3_KC + 8 * (lane >> 4) + j;
        const float v = w[((size_t)co * cin + ci) * 9 + tap];
        const __bf16 hi = (__bf16)v;
        const size_t o = (idx / 512) * 1024 + (idx % 512);
        out[o] = hi;
        out[o + 512] = (__bf16)(v - (float)hi);
    }
}

int g_x3_cus[16] = {};

template <int RES, int LOGW, int PRO>
int launch_x3_one(const ConvArgs& a, hipStream_t stream, bool prepare_only) {
    using G = X3Geom<LOGW>;
    auto kern = conv3_x3ws_kernel<RES, LOGW, PRO>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
    static bool attr_done[16] = {};  // per device: the attribute belongs to the device's copy of the code object
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
        if (e != hipSuccess) return (int)e;
        attr_done[dev] = true;
    }
    if (!g_x3_cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return (int)hipErrorUnknown;
        g_x3_cus[dev] = n;
    }
    if (prepare_only) return 0;
    const int ntiles = a.B * G::TPI;
    const int grid = ntiles < g_x3_cus[dev] ? ntiles : g_x3_cus[dev];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(X3_NTHR), X3_LDS, stream, a, ntiles);
    return (int)hipGetLastError();
}

}  // namespace

// statistics slots per image written by this kernel: one per (tile, producer wave)
int conv_x3ws_stat_slots(int W) { return (W == 32 ? X3Geom<5>::TPI : X3Geom<4>::TPI) * 4; }

bool conv_x3ws_shape_ok(int cout, int cin, int res) { return cout == 256 && (res == 32 || res == 16) && (cin % X3_KC) == 0 && cin / X3_KC >= 4; }

bool conv_x3ws_supported(int ks, int pro, int res, int outmode, const ConvArgs& a) {
    const int cin = a.C1 + a.C2;
    const size_t src_bytes = (size_t)a.B * a.Hs * a.Ws * (size_t)(a.C1 > a.C2 ? a.C1 : a.C2) * 4;
    if (src_bytes >= (1ull << 32)) return false;  // the staged loads use 32-bit byte offsets into each source tensor
    if (ks != 3 || outmode != OUT_NHWC || a.H != a.W || !conv_x3ws_shape_ok(a.Cout, cin, a.W) || (a.C1 % X3_KC) || !a.wpack_ws) return false;
    if (pro == PRO_NONE) return res == RES_NONE && a.Hs == a.H && a.Ws == a.W;
    return pro == PRO_GN_SILU && (res == RES_NONE || res == RES_UP) && a.ab != nullptr;
}

int launch_conv_x3ws(int res, const ConvArgs& a, hipStream_t stream, bool prepare_only, int pro) {
    if (pro == PRO_NONE) {
        if (res != RES_NONE) return (int)hipErrorInvalidValue;
        return a.W == 32 ? launch_x3_one<RES_NONE, 5, PRO_NONE>(a, stream, prepare_only) : launch_x3_one<RES_NONE, 4, PRO_NONE>(a, stream, prepare_only);
    }
    if (a.W == 32)
        return res == RES_UP ? launch_x3_one<RES_UP, 5, PRO_GN_SILU>(a, stream, prepare_only) : launch_x3_one<RES_NONE, 5, PRO_GN_SILU>(a, stream, prepare_only);
    return res == RES_UP ? launch_x3_one<RES_UP, 4, PRO_GN_SILU>(a, stream, prepare_only) : launch_x3_one<RES_NONE, 4, PRO_GN_SILU>(a, stream, prepare_only);
}

int launch_pack_conv_weights_x3ws(const float* w_oihw, void* wpack_ws, int cout, int cin, hipStream_t stream) {
    const size_t total = (size_t)cout * cin * 9;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv_weights_x3ws_kernel, dim3(grid), dim3(256), 0, stream, w_oihw, (__bf16*)wpack_ws, cout, cin);
    return (int)hipGetLastError();
}
