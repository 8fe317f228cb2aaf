// Output head of the EDM U-Net on the matrix cores:
//     F = aux_conv(silu(aux_norm(x)))                      reference fastgen/networks/EDM/network.py:553-557
//     D = c_skip * x_t + c_out * F                          precond_output, :798-805
// aux_conv is a 3x3 convolution to img_channels (3) outputs: as a GEMM its N is tiny, so instead of the 256-wide conv
// kernel this one pads N to a single 32-column MFMA tile and splits the 128-pixel tile over the four waves along M.
// The activation halo is normalised + SiLU'd once per 64-channel chunk into LDS (same image layout as conv.hip) and
// read by all nine taps; the kernel is bound by reading x once (1.5x with the halo rows).
#include "common.h"
#include "misc.h"

namespace {

constexpr int APITCH = 144;
constexpr int HWID = 34, HROWS = 6;  // halo of a 4 x 32 pixel tile

// T: compute type of common.h (float / __bf16 / bf16x3); x is in its storage type, the weights in its packed type
template <typename T>
__global__ __launch_bounds__(256) void aux_head_kernel(const typename DT<T>::ST* __restrict__ x, const float2* __restrict__ ab,
                                                       const typename DT<T>::WT* __restrict__ wpack, const float* __restrict__ bias,
                                                       const float* __restrict__ x_t, const float* __restrict__ coef,
                                                       float* __restrict__ out, float* __restrict__ raw, int B, int C, int cout) {
    typedef typename DT<T>::ST ST;
    typedef typename DT<T>::WT WT;
    constexpr int KC = DT<T>::KC, KK = KC / 16, OPP = KC / 8;
    constexpr int FB = DT<T>::FRAG_BYTES, WP = DT<T>::WPARTS;
    constexpr bool FAST = DT<T>::FAST;
    constexpr int LOG_OPP = (OPP == 8) ? 3 : 2;
    constexpr int PSTRIDE = 256 / OPP;
    constexpr int HALO_PIX = HROWS * HWID;
    constexpr int NITEMS = (HALO_PIX * OPP + 255) / 256;
    __shared__ __attribute__((aligned(16))) char smem[HALO_PIX * APITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.x >> 3, row0 = (blockIdx.x & 7) * 4;
    const int oct = tid & (OPP - 1), hq0 = tid >> LOG_OPP;
    const int nchunk = C / KC;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        // ---- stage: silu(a*x+b) of the 6 x 34 halo, this chunk's KC channels --------------------------------
        float2 abr[8];
        {
            const float2* p = ab + (size_t)n * C + chunk * KC + oct * 8;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(p + j);
                abr[j] = make_float2(q[0], q[1]);
                abr[j + 1] = make_float2(q[2], q[3]);
            }
        }
#pragma unroll
        for (int i = 0; i < NITEMS; ++i) {
            const int hq = hq0 + i * PSTRIDE;
            if (hq < HALO_PIX) {
                const int hx = hq % HWID, hy = hq / HWID;
                const int y = row0 + hy - 1, xx = hx - 1;
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = 0.f;
                if (y >= 0 && y < 32 && xx >= 0 && xx < 32) {
                    const ST* p = x + (((size_t)n * 32 + y) * 32 + xx) * C + chunk * KC + oct * 8;
                    const f32x4 lo = load4(p), hi = load4(p + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = silu_f<FAST>(fmaf(lo[j], abr[j].x, abr[j].y));
                        o[j + 4] = silu_f<FAST>(fmaf(hi[j], abr[j + 4].x, abr[j + 4].y));
                    }
                }
                lds_store_a<T>(smem + hq * APITCH + oct * FB, o);
            }
        }
        __syncthreads();
        // ---- multiply: wave w = image row row0 + w of the tile, 9 taps x KK k-steps, N padded to 32 ----------------
        const WT* wp = wpack + (size_t)chunk * 9 * KK * 512 * WP + lane * 8;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const char* abase = smem + ((wave + tap / 3) * HWID + r + tap % 3) * APITCH + h * FB;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const Frag8<T> af = lds_read_a<T>(abase, kk);
                const Frag8<T> bf = load_wfrag<T>(wp + (tap * KK + kk) * 512 * WP);
                mma16(acc, af, bf);
            }
        }
        __syncthreads();
    }
    // ---- epilogue: lanes r < cout hold output channel r for 16 pixels of the row ---------------------------------
    if (r < cout) {
        const float c_skip = coef[2 * B + n], c_out = coef[3 * B + n];
        const float bv = bias[r];
        const int y = row0 + wave;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const size_t o = (((size_t)n * cout + r) * 32 + y) * 32 + acc_row(i, h);
            out[o] = c_skip * x_t[o] + c_out * (acc[i] + bv);
            if (raw) raw[o] = acc[i] + bv;  // the network's own output F, kept for the forward-mode pass
        }
    }
}

// packed[chunk][tap][kk][lane][j] = W[co = lane & 31][ci = chunk*KC + kk*16 + 8*(lane>>5) + j][tap], zero for co >= cout
template <typename T>
__global__ void pack_aux_weights_kernel(const float* __restrict__ w, typename DT<T>::WT* __restrict__ out, int C, int cout) {
    constexpr int KC = DT<T>::KC, KK = KC / 16;
    const int total = C * 9 * 32;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int t = idx;
        const int j = t % 8; t /= 8;
        const int lane = t % 64; t /= 64;
        const int kk = t % KK; t /= KK;
        const int tap = t % 9; t /= 9;
        const int chunk = t;
        const int co = lane & 31;
        const int ci = chunk * KC + kk * 16 + 8 * (lane >> 5) + j;
        const float v = (co < cout) ? w[((size_t)co * C + ci) * 9 + tap] : 0.f;
        if constexpr (DT<T>::WPARTS == 1) {
            out[idx] = (typename DT<T>::WT)v;
        } else {  // bf16x3: the lo plane of a fragment follows its hi plane
            const size_t o = ((size_t)idx / 512) * 1024 + (idx % 512);
            const __bf16 hi = (__bf16)v;
            out[o] = hi;
            out[o + 512] = (__bf16)(v - (float)hi);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stem: out = conv3x3(c_in * x_t) + bias, img_channels (3) -> 128 channels (precond_input :771-773 folded into the
// first encoder conv :426).  K = 27 is padded to two 16-deep MFMA steps; the A fragment (32 pixels x 16 k) is gathered
// directly from the NCHW fp32 input (lanes = consecutive x: coalesced), the four 32-channel B tiles are pre-packed.
// One wave = 32 consecutive pixels x 128 channels.  Bound by writing the NHWC output.
template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ c_in,
                                                   const T* __restrict__ wpack, const float* __restrict__ bias,
                                                   T* __restrict__ out, float2* __restrict__ stats, int B, int res, int cin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int hw = res * res;
    const long long p0 = ((long long)blockIdx.x * 4 + wave) * 32;  // first pixel (flattened over the batch) of this wave
    const int n = (int)(p0 / hw);
    if (n >= B) return;
    const int pix = (int)(p0 % hw) + r;
    const int y = pix / res, xx = pix % res;
    const float ci_scale = c_in[n];
    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = s * 16 + 8 * h + j;  // k = ci*9 + kh*3 + kw (OIHW order)
            const int ci = k / 9, kh = (k % 9) / 3, kw = k % 3;
            const int yy = y + kh - 1, xs = xx + kw - 1;
            v[j] = 0.f;
            if (k < cin * 9 && yy >= 0 && yy < res && xs >= 0 && xs < res)
                v[j] = ci_scale * x[(((size_t)n * cin + ci) * res + yy) * res + xs];
        }
        Frag8<T> af;
        store_frag(reinterpret_cast<T*>(&af), v);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const Frag8<T> bf = load_frag(wpack + ((s * 4 + nt) * 64 + lane) * 8);
            mma16(acc[nt], af, bf);
        }
    }
    T* obase = out + ((size_t)n * hw + (p0 % hw)) * 128;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const float bv = bias[nt * 32 + r];
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = acc[nt][i] + bv;
            obase[(size_t)acc_row(i, h) * 128 + nt * 32 + r] = (T)v;
            sv += v;
            qv = fmaf(v, v, qv);
        }
        // GroupNorm partial statistics of this wave's 32 pixels (slot = wave within the image), per channel quad: the
        // format the conv epilogues write (misc.hip gn_finalize_kernel), so block0's norm0 needs no pass over the tensor
        if (stats) {
            sv += __shfl_xor(sv, 32), qv += __shfl_xor(qv, 32);
            sv += __shfl_xor(sv, 1), qv += __shfl_xor(qv, 1);
            sv += __shfl_xor(sv, 2), qv += __shfl_xor(qv, 2);
            if (h == 0 && (r & 3) == 0)
                stats[((size_t)n * (hw / 32) + (int)(p0 % hw) / 32) * 32 + nt * 8 + (r >> 2)] = make_float2(sv, qv);
        }
    }
}

// packed[s][nt][lane][j] = W[co = nt*32 + (lane&31)][k = s*16 + 8*(lane>>5) + j], zero for k >= cin*9
template <typename T>
__global__ void pack_stem_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int cin) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * 4 * 64 * 8) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, nt = (idx >> 9) & 3, s = idx >> 11;
    const int co = nt * 32 + (lane & 31), k = s * 16 + 8 * (lane >> 5) + j;
    out[idx] = (k < cin * 9) ? (T)w[(size_t)co * cin * 9 + k] : (T)0.f;
}

}  // namespace

size_t aux_pack_elems(int C) { return (size_t)C * 9 * 32; }

int launch_pack_aux_weights(int dtype, const float* w, void* out, int C, int cout, hipStream_t s) {
    const int grid = (int)((aux_pack_elems(C) + 255) / 256);
    if (dtype == 2)
        hipLaunchKernelGGL(pack_aux_weights_kernel<bf16x3>, dim3(grid), dim3(256), 0, s, w, (__bf16*)out, C, cout);
    else if (dtype)
        hipLaunchKernelGGL(pack_aux_weights_kernel<__bf16>, dim3(grid), dim3(256), 0, s, w, (__bf16*)out, C, cout);
    else
        hipLaunchKernelGGL(pack_aux_weights_kernel<float>, dim3(grid), dim3(256), 0, s, w, (float*)out, C, cout);
    return (int)hipGetLastError();
}

// supported: res == 32, C a multiple of the chunk size, cout <= 32
int aux_head_supported(int dtype, int res, int C, int cout) { return res == 32 && C % (dtype == 1 ? 64 : 32) == 0 && cout <= 32; }

int launch_aux_head(int dtype, const void* x, const float2* ab, const void* wpack, const float* bias, const float* x_t,
                    const float* coef, float* out, int B, int C, int cout, hipStream_t s, float* raw) {
    if (dtype == 2)
        hipLaunchKernelGGL(aux_head_kernel<bf16x3>, dim3(B * 8), dim3(256), 0, s, (const float*)x, ab, (const __bf16*)wpack, bias, x_t, coef, out, raw, B, C, cout);
    else if (dtype)
        hipLaunchKernelGGL(aux_head_kernel<__bf16>, dim3(B * 8), dim3(256), 0, s, (const __bf16*)x, ab, (const __bf16*)wpack, bias, x_t, coef, out, raw, B, C, cout);
    else
        hipLaunchKernelGGL(aux_head_kernel<float>, dim3(B * 8), dim3(256), 0, s, (const float*)x, ab, (const float*)wpack, bias, x_t, coef, out, raw, B, C, cout);
    return (int)hipGetLastError();
}

// ---- stem -------------------------------------------------------------------------------------------------------
int stem_supported(int res, int cin, int cout) { return cout == 128 && cin * 9 <= 32 && (res * res) % 32 == 0; }
size_t stem_pack_elems() { return 2 * 4 * 64 * 8; }
int launch_pack_stem_weights(int dtype, const float* w, void* out, int cin, hipStream_t s) {
    if (dtype)
        hipLaunchKernelGGL(pack_stem_weights_kernel<__bf16>, dim3(16), dim3(256), 0, s, w, (__bf16*)out, cin);
    else
        hipLaunchKernelGGL(pack_stem_weights_kernel<float>, dim3(16), dim3(256), 0, s, w, (float*)out, cin);
    return (int)hipGetLastError();
}
int launch_stem(int dtype, const float* x, const float* c_in, const void* wpack, const float* bias, void* out, float2* stats,
                int B, int res, int cin, hipStream_t s) {
    const long long waves = (long long)B * res * res / 32;
    dim3 grid((unsigned)((waves + 3) / 4));
    if (dtype)
        hipLaunchKernelGGL(stem_kernel<__bf16>, grid, dim3(256), 0, s, x, c_in, (const __bf16*)wpack, bias, (__bf16*)out, stats, B, res, cin);
    else
        hipLaunchKernelGGL(stem_kernel<float>, grid, dim3(256), 0, s, x, c_in, (const float*)wpack, bias, (float*)out, stats, B, res, cin);
    return (int)hipGetLastError();
}
