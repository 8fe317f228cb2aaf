// Small kernels around the fused conv: GroupNorm statistics, embedding MLP, stem / output convolutions with the EDM
// preconditioning folded in, the sampler's elementwise steps and the on-device normal generator.
// Reference line numbers are relative to fastgen/networks/EDM/network.py unless another file is named.
#include "common.h"
#include "misc.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// GroupNorm statistics -> per-(image, channel) affine coefficients  y = a*x + b
//   a = rstd * gamma,  b = beta - mean * a      (GroupNorm.forward :141-149; groups = min(32, C/4), biased variance)
// x is the virtual concat [x1 (C1) | x2 (C2)], NHWC.  One workgroup per image; a thread owns one channel quad.
// HBM-bound: reads the tensor once with 16-byte loads.
template <typename AT>
__global__ __launch_bounds__(512) void gn_coeffs_kernel(const AT* __restrict__ x1, int C1, const AT* __restrict__ x2,
                                                        int C2, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, float2* __restrict__ ab,
                                                        int hw, float2* __restrict__ mr) {
    const int C = C1 + C2;
    const int Q = C >> 2;                 // channel quads per pixel
    const int lanes = blockDim.x / Q;     // pixel lanes
    const int groups = min(32, C / 4);
    const int cpg = C / groups;
    const int n = blockIdx.x;
    const int t = threadIdx.x;
    __shared__ double s_sum[512], s_sq[512];
    __shared__ float s_mean[32], s_rstd[32];
    float s = 0.f, ss = 0.f;
    if (t < Q * lanes) {
        const int q = t % Q, pl = t / Q;
        const int c = q * 4;
        const AT* base = (c < C1) ? x1 + (size_t)n * hw * C1 + c : x2 + (size_t)n * hw * C2 + (c - C1);
        const int stride = (c < C1) ? C1 : C2;
        // fp32 partials over <= hw/lanes pixels, four independent chains per thread; combined in fp64 below
        for (int p = pl; p < hw; p += lanes) {
            const f32x4 v = load4(base + (size_t)p * stride);
            s += (v[0] + v[1]) + (v[2] + v[3]);
            ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    }
    s_sum[t] = s;
    s_sq[t] = ss;
    __syncthreads();
    if (t < groups) {
        // quads of group t: [t*cpg/4, (t+1)*cpg/4) x all pixel lanes (cpg is a multiple of 4)
        double a = 0.0, b = 0.0;
        const int q0 = t * cpg / 4, q1 = (t + 1) * cpg / 4;
        for (int pl = 0; pl < lanes; ++pl)
            for (int q = q0; q < q1; ++q) {
                a += s_sum[pl * Q + q];
                b += s_sq[pl * Q + q];
            }
        const double cnt = (double)cpg * hw;
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[t] = (float)mean;
        s_rstd[t] = (float)(1.0 / sqrt(var + (double)eps));
        if (mr) mr[(size_t)n * groups + t] = make_float2(s_mean[t], s_rstd[t]);  // kept for the backward pass
    }
    __syncthreads();
    for (int c = t; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const float a = s_rstd[g] * gamma[c];
        ab[(size_t)n * C + c] = make_float2(a, fmaf(-a, s_mean[g], beta[c]));
    }
}

// GroupNorm coefficients from the partial statistics the conv epilogues leave behind (conv.hip): no pass over the
// tensor.  st[b][slot][quad] = {sum, sum of squares} over a pixel tile of channels 4*quad..4*quad+3; the normalised
// tensor is the virtual concat of up to two producers.  One workgroup per image, one thread per channel; combined in
// fp64 in a fixed order (deterministic).
__global__ __launch_bounds__(512) void gn_finalize_kernel(const float2* __restrict__ st1, int C1, int S1, const float2* __restrict__ st2, int C2,
                                                         int S2, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         float2* __restrict__ ab, int hw, float2* __restrict__ mr) {
    // Two levels, fixed order (deterministic), fp64: (1) thread (slot lane sg, channel quad q) adds the slots sg, sg + SG, ... of
    // its quad - the loads of a quad's up to 32 slots (conv_ws3.hip writes one per tile and producer wave) are spread over SG
    // threads instead of one thread walking them all; (2) one thread per channel adds its group's quads x lanes.
    const int C = C1 + C2, Qt = C >> 2;
    const int groups = min(32, C / 4);
    const int cpg = C / groups;
    const int n = blockIdx.x;
    const int SG = (int)blockDim.x / Qt;  // >= 4 for C <= 512 (checked by the launcher)
    __shared__ double ps[512], pq[512];
    const int q = threadIdx.x % Qt, sg = threadIdx.x / Qt;
    if (sg < SG) {
        const bool first = q * 4 < C1;
        const float2* st = first ? st1 : st2;
        const int S = first ? S1 : S2;
        const int Q = (first ? C1 : C2) >> 2;
        const int ql = first ? q : q - (C1 >> 2);
        double s = 0.0, v2 = 0.0;
        for (int sl = sg; sl < S; sl += SG) {
            const float2 v = st[((size_t)n * S + sl) * Q + ql];
            s += v.x;
            v2 += v.y;
        }
        ps[sg * Qt + q] = s;
        pq[sg * Qt + q] = v2;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        double s = 0.0, v2 = 0.0;
        for (int quad = g * cpg / 4; quad < (g + 1) * cpg / 4; ++quad)
            for (int l = 0; l < SG; ++l) {
                s += ps[l * Qt + quad];
                v2 += pq[l * Qt + quad];
            }
        const double cnt = (double)cpg * hw;
        const double mean = s / cnt;
        double var = v2 / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float m = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float a = rstd * gamma[c];
        ab[(size_t)n * C + c] = make_float2(a, fmaf(-a, m, beta[c]));
        if (mr && c % cpg == 0) mr[(size_t)n * groups + g] = make_float2(m, rstd);  // kept for the backward pass
    }
}

// ---------------------------------------------------------------------------------------------------
// EDM preconditioning coefficients in fp64 (precond_input :755-778, precond_output :781-805), cast to fp32 exactly
// where the reference casts (`.to(x_t.dtype)`).  coef[5][B] = c_in, c_noise, c_skip, c_out, r_noise.
// drop bit 0 (drop_precond 'input'/'both', :929-934): x and the noise labels pass through unchanged (t, r cast to fp32);
// drop bit 1 ('output'/'both', :959-960): the network output is returned as is (c_skip = 0, c_out = 1).
__global__ void precond_coef_kernel(const double* __restrict__ t, int t_stride, const double* __restrict__ r, int r_stride,
                                    double sigma_data, double sigma_shift, double clamp_min, int drop,
                                    float* __restrict__ coef, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double tv = t[(size_t)b * t_stride];
    const double rv = r ? r[(size_t)b * r_stride] : 0.0;
    const double sd2 = sigma_data * sigma_data;
    if (drop & 1) {
        coef[b] = 1.0f;
        coef[B + b] = (float)tv;
        coef[4 * B + b] = (float)rv;
    } else {
        coef[b] = (float)(1.0 / sqrt(sd2 + tv * tv));
        coef[B + b] = (float)(log(fmax(tv, clamp_min)) / 4.0);
        coef[4 * B + b] = (float)(log(fmax(rv, clamp_min)) / 4.0);
    }
    if (drop & 2) {
        coef[2 * B + b] = 0.0f;
        coef[3 * B + b] = 1.0f;
    } else {
        const double ts = tv - sigma_shift;
        coef[2 * B + b] = (float)(sd2 / (ts * ts + sd2));
        coef[3 * B + b] = (float)(ts * sigma_data / sqrt(ts * ts + sd2));
    }
}

// ---------------------------------------------------------------------------------------------------
// Mapping-network input: positional embedding of c_noise (sin|cos after the flip at :503) + map_label(labels*sqrt(L))
// (:306-319, :501-517).  out [B][N].
// With r_timestep (:401-408, 505-509) a second embedding of the r labels is concatenated: N = noise_ch * 2.
__global__ void mapping_in_kernel(const float* __restrict__ c_noise, const float* __restrict__ r_noise,
                                  const float* __restrict__ freqs, const float* __restrict__ labels, int label_dim,
                                  const float* __restrict__ wl, const float* __restrict__ bl, float* __restrict__ out, int B,
                                  int N, int noise_ch, const float* __restrict__ aug, const float* __restrict__ wa, int aug_dim) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N) return;
    const int b = idx / N, j = idx % N;
    const int half = noise_ch / 2;
    const int jj = j % noise_ch;
    const float lab = (j < noise_ch) ? c_noise[b] : r_noise[b];
    const float ang = lab * freqs[jj % half];
    float v = (jj < half) ? sinf(ang) : cosf(ang);
    if (label_dim > 0) {
        float acc = 0.f;
        const float sc = sqrtf((float)label_dim);
        if (labels)
            for (int i = 0; i < label_dim; ++i) acc = fmaf(labels[(size_t)b * label_dim + i] * sc, wl[(size_t)j * label_dim + i], acc);
        v += acc + bl[j];
    }
    if (aug) {  // + map_augment(augment_labels), no bias (EDM/network.py:518-519): training-time augmentation pipeline
        float acc = 0.f;
        for (int i = 0; i < aug_dim; ++i) acc = fmaf(aug[(size_t)b * aug_dim + i], wa[(size_t)j * aug_dim + i], acc);
        v += acc;
    }
    out[idx] = v;
}

// ---------------------------------------------------------------------------------------------------
// y[B,O] = act(x[B,I] @ W[O,I]^T + bias)   (Linear.forward :47-51), fp32 FMA, 64x64 tile, 4x4 per thread.
template <int ACT>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, int B, int I,
                                                     int O) {
    __shared__ float xs[16][65], ws[16][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int b0 = blockIdx.y * 64, o0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < I; k0 += 16) {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int rr = e >> 4, kk = e & 15;
            xs[kk][rr] = (b0 + rr < B && k0 + kk < I) ? x[(size_t)(b0 + rr) * I + k0 + kk] : 0.f;
            ws[kk][rr] = (o0 + rr < O && k0 + kk < I) ? w[(size_t)(o0 + rr) * I + k0 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float xv[4], wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = xs[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = ws[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(xv[i], wv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = b0 + ty * 4 + i, o = o0 + tx * 4 + j;
            if (b < B && o < O) {
                float v = acc[i][j] + (bias ? bias[o] : 0.f);
                if (ACT == 1) v = v / (1.0f + expf(-v));
                y[(size_t)b * O + o] = v;
            }
        }
}

// Same product on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 multiply-add, 256 FLOP/clk/CU).  One wave per
// 32x32 output tile, operands straight from global memory (x is <= 1 MB and W is re-read from L2 by the B/32 row tiles).
// Lane (m = lane & 31, h = lane >> 5) fetches the 8 consecutive k = k0 + 8h .. 8h+7 of its row; MFMA i of the 16-wide k
// block then multiplies k = k0 + i (lanes 0-31) and k0 + 8 + i (lanes 32-63) - a permutation of the k order that is the same
// for both operands.  Needs I % KB == 0 (KB = 64 or 16) and O % 32 == 0; ragged B is clamped on load and masked on store.
// A wave covers NB row tiles (32 NB batch rows) with one weight fragment stream.
template <int ACT, int KB, int NB>
__global__ __launch_bounds__(64) void linear_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int B, int I,
                                                          int O) {
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    const int o0 = blockIdx.x * 32, b0 = blockIdx.y * (32 * NB);
    const float4* xp[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) xp[t] = reinterpret_cast<const float4*>(x + (size_t)min(b0 + 32 * t + m, B - 1) * I + 8 * h);
    const float4* wp = reinterpret_cast<const float4*>(w + (size_t)(o0 + m) * I + 8 * h);
    f32x16 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // KB k-values per trip: all of a trip's loads are issued before its MFMAs, so a wave waits for memory I / KB times
    // (the other waves of the SIMD run their MFMAs meanwhile)
    for (int k0 = 0; k0 < I; k0 += KB) {
        float4 xv[NB][KB / 8], wv[KB / 8];
#pragma unroll
        for (int u = 0; u < KB / 16; ++u) {
            wv[2 * u] = wp[k0 / 4 + 4 * u], wv[2 * u + 1] = wp[k0 / 4 + 4 * u + 1];
#pragma unroll
            for (int t = 0; t < NB; ++t) xv[t][2 * u] = xp[t][k0 / 4 + 4 * u], xv[t][2 * u + 1] = xp[t][k0 / 4 + 4 * u + 1];
        }
#pragma unroll
        for (int u = 0; u < KB / 8; ++u)
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[t][u].x, wv[u].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[t][u].y, wv[u].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[t][u].z, wv[u].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[t][u].w, wv[u].w, acc[t], 0, 0, 0);
            }
    }
    // C layout: column (output feature) = lane & 31, rows (batch) 8 * (i >> 2) + 4 * h + (i & 3)
    const int o = o0 + m;
    const float bo = bias ? bias[o] : 0.f;
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int b = b0 + 32 * t + 8 * (i >> 2) + 4 * h + (i & 3);
            if (b < B) {
                float v = acc[t][i] + bo;
                if (ACT == 1) v = v / (1.0f + expf(-v));
                y[(size_t)b * O + o] = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// Stem: out[n,y,x,co] = conv3x3(c_in[n] * x_t)[co] + bias   (precond_input :771-773 folded into enc '{res}x{res}_conv',
// :426).  x_t is NCHW fp32, out NHWC fp32.  One workgroup per image row; memory-bound on the output write.
template <typename AT>
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ c_in,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      AT* __restrict__ out, int res, int cin, int cout) {
    extern __shared__ float sw[];  // [cin*9][cout]
    const int n = blockIdx.x / res, y = blockIdx.x % res;
    const int K = cin * 9;
    for (int e = threadIdx.x; e < K * cout; e += blockDim.x) {
        const int co = e % cout, k = e / cout;
        sw[e] = w[(size_t)co * K + k];  // OIHW: k = ci*9 + kh*3 + kw
    }
    __syncthreads();
    const float ci_scale = c_in[n];
    const int per_px = cout / 16;  // threads per pixel, 16 output channels each
    for (int item = threadIdx.x; item < res * per_px; item += blockDim.x) {
        const int px = item / per_px, cg = item % per_px;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int ci = 0; ci < cin; ++ci)
            for (int kh = 0; kh < 3; ++kh) {
                const int yy = y + kh - 1;
                if (yy < 0 || yy >= res) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    const int xx = px + kw - 1;
                    if (xx < 0 || xx >= res) continue;
                    const float v = ci_scale * x[(((size_t)n * cin + ci) * res + yy) * res + xx];
                    const float* wr = sw + (size_t)(ci * 9 + kh * 3 + kw) * cout + cg * 16;
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
                }
            }
        AT* o = out + (((size_t)n * res + y) * res + px) * cout + cg * 16;
#pragma unroll
        for (int j = 0; j < 16; j += 4)
            store4(o + j, f32x4{acc[j] + bias[cg * 16 + j], acc[j + 1] + bias[cg * 16 + j + 1],
                                acc[j + 2] + bias[cg * 16 + j + 2], acc[j + 3] + bias[cg * 16 + j + 3]});
    }
}

// ---------------------------------------------------------------------------------------------------
// Output head: F = aux_conv(silu(aux_norm(x)))  (:553-557) and D = c_skip * x_t + c_out * F  (precond_output :798-805)
// x NHWC [B,res,res,C] fp32, ab = aux_norm coefficients, w OIHW [cout(<=4)][C][3][3]; x_t / out NCHW.
// One workgroup per image row: thread = (pixel, 1/8 channel slice); the slices meet in a wave shuffle.
template <bool FAST, typename AT>
__global__ __launch_bounds__(256) void aux_out_kernel(const AT* __restrict__ x, const float2* __restrict__ ab,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ x_t, const float* __restrict__ coef,
                                                      float* __restrict__ out, float* __restrict__ raw, int B, int res, int C, int cout) {
    extern __shared__ float smem_f[];
    float* sw = smem_f;                                               // [9][cout][C]
    float2* sab = reinterpret_cast<float2*>(smem_f + 9 * cout * C);   // [C]
    const int n = blockIdx.x / res, y = blockIdx.x % res;
    for (int e = threadIdx.x; e < 9 * cout * C; e += blockDim.x) {
        const int c = e % C, co = (e / C) % cout, tap = e / (C * cout);
        sw[e] = w[((size_t)co * C + c) * 9 + tap];
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) sab[c] = ab[(size_t)n * C + c];
    __syncthreads();
    const int slice = C / 8;
    for (int px0 = 0; px0 < res; px0 += 32) {
        const int px = px0 + (threadIdx.x >> 3), cs = threadIdx.x & 7;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (px < res) {
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = px + tap % 3 - 1;
                if (yy < 0 || yy >= res || xx < 0 || xx >= res) continue;
                const AT* xp = x + (((size_t)n * res + yy) * res + xx) * C + cs * slice;
                for (int c = 0; c < slice; c += 4) {
                    const f32x4 v = load4(xp + c);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const float2 k = sab[cs * slice + c + d];
                        const float s = silu_f<FAST>(fmaf(v[d], k.x, k.y));
                        for (int co = 0; co < cout; ++co)
                            acc[co] = fmaf(s, sw[(tap * cout + co) * C + cs * slice + c + d], acc[co]);
                    }
                }
            }
        }
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            acc[co] += __shfl_xor(acc[co], 1);
            acc[co] += __shfl_xor(acc[co], 2);
            acc[co] += __shfl_xor(acc[co], 4);
        }
        if (cs == 0 && px < res) {
            const float c_skip = coef[2 * B + n], c_out = coef[3 * B + n];
            for (int co = 0; co < cout; ++co) {
                const size_t o = (((size_t)n * cout + co) * res + y) * res + px;
                out[o] = c_skip * x_t[o] + c_out * (acc[co] + bias[co]);
                if (raw) raw[o] = acc[co] + bias[co];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Sampler elementwise steps, fp64 arithmetic then one rounding to fp32 (noise_schedule.py:72-88, 425-449, 544-574).
// t comes from device memory (tp[ti]) when tp != nullptr so a captured graph can be replayed with new timesteps.
__device__ __forceinline__ double pick_t(double tv, const double* tp, int ti) { return tp ? tp[ti] : tv; }

__global__ void latents_kernel(const float* __restrict__ noise, double tv, const double* tp, int ti,
                               float* __restrict__ out, int64_t total) {
    const double t = pick_t(tv, tp, ti);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)((double)noise[i] * t);
}
// sched 0: EDM (alpha = 1, sigma = t; noise_schedule.py:773-777); sched 1: rectified flow (alpha = 1 - t, sigma = t; :1337-1341)
__device__ __forceinline__ double alpha_of(double t, int sched) { return sched ? 1.0 - t : 1.0; }

__global__ void forward_process_kernel(const float* x0, const float* __restrict__ eps, double tv, const double* tp,
                                       int ti, int sched, float* out, int64_t total) {  // out may alias x0
#pragma clang fp contract(off)  // torch evaluates mul, mul, add with separate roundings: no fused multiply-add
    const double t = pick_t(tv, tp, ti);
    const double al = alpha_of(t, sched);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)((double)x0[i] * al + (double)eps[i] * t);
}
__global__ void x0_to_eps_kernel(const float* __restrict__ xt, const float* __restrict__ x0, double tv, const double* tp,
                                 int ti, int sched, double clamp_min, float* __restrict__ out, int64_t total) {
#pragma clang fp contract(off)
    const double t0 = pick_t(tv, tp, ti);
    const double al = alpha_of(t0, sched);
    const double t = (t0 >= 0.0) ? fmax(t0, clamp_min) : fmin(t0, -clamp_min);  // non_zero_clamp, noise_schedule.py:123-129
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)(((double)xt[i] - (double)x0[i] * al) / t);
}
// MeanFlow update x <- x - delta_t * u with delta_t = fp32(t_a - t_b) (mean_flow.py:366-376; t_b = nullptr index < 0: 0).
// Two fp32 roundings (product, then difference) as torch evaluates it: no fused multiply-add.
__global__ void meanflow_update_kernel(const float* x, const float* __restrict__ u, const double* tp, int ia, int ib,
                                       float* out, int64_t total) {  // out may alias x
#pragma clang fp contract(off)
    const float dt = (float)(ib >= 0 ? tp[ia] - tp[ib] : tp[ia]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = x[i] - dt * u[i];
}

// ---------------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: out[4c..4c+3] from counter (c, offset) and key = seed.  The seed/offset pair is read
// from device memory when sp != nullptr (graph replay with a fresh seed).
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}
__global__ void randn_kernel(float* __restrict__ out, int64_t total, uint64_t seed_v, uint64_t offset_v,
                             const uint64_t* sp) {
    const uint64_t seed = sp ? sp[0] : seed_v;
    const uint64_t offset = sp ? sp[1] + offset_v : offset_v;
    const int64_t nquad = (total + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * blockDim.x) {
        uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int rnd = 0; rnd < 10; ++rnd) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float z[4];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float u1 = ((float)c[2 * p] + 0.5f) * 2.3283064365386963e-10f;      // (0,1]
            const float u2 = ((float)c[2 * p + 1] + 0.5f) * 2.3283064365386963e-10f;
            const float rad = sqrtf(-2.0f * logf(fmaxf(u1, 1e-37f)));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            z[2 * p] = rad * cs;
            z[2 * p + 1] = rad * sn;
        }
        for (int d = 0; d < 4; ++d)
            if (4 * q + d < total) out[4 * q + d] = z[d];
    }
}

// copy + layout helpers
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW) {
    const int64_t total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        const int64_t t = i / C;
        const int p = t % HW;
        const int n = t / HW;
        out[i] = in[((int64_t)n * C + c) * HW + p];
    }
}

// Down-sampling blocks: conv0 consumes mean_2x2(silu(gn(x))) (Conv2d.forward :118-123 applied to silu(norm0(x)), :276).
// Doing that inside the conv's operand staging costs 4x the transform work per staged pixel; this pass writes the pooled
// tensor once (HBM-bound: reads x, writes a quarter of it) and the conv then runs without prologue.
// x [B, 2H, 2W, C] -> out [B, H, W, C]; a thread owns 8 channels of one output pixel.
template <typename AT, bool FAST>
__global__ void gn_silu_pool_kernel(const AT* __restrict__ x, const float2* __restrict__ ab, AT* __restrict__ out, int B,
                                    int H, int W, int C) {
    const int oc = C >> 3;
    const int64_t total = (int64_t)B * H * W * oc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % oc);
        int64_t t = i / oc;
        const int xx = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        float2 k[8];
        const float2* kp = ab + (size_t)n * C + o * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) k[j] = kp[j];
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const AT* p = x + ((((size_t)n * 2 * H + 2 * y + (d >> 1)) * 2 * W) + 2 * xx + (d & 1)) * C + o * 8;
            const f32x4 lo = load4(p), hi = load4(p + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] += silu_f<FAST>(fmaf(lo[j], k[j].x, k[j].y));
                acc[j + 4] += silu_f<FAST>(fmaf(hi[j], k[j + 4].x, k[j + 4].y));
            }
        }
        AT* q = out + (size_t)i * 8;
        store4(q, f32x4{0.25f * acc[0], 0.25f * acc[1], 0.25f * acc[2], 0.25f * acc[3]});
        store4(q + 4, f32x4{0.25f * acc[4], 0.25f * acc[5], 0.25f * acc[6], 0.25f * acc[7]});
    }
}

template <typename A, typename B>
__global__ void convert_kernel(const A* __restrict__ in, B* __restrict__ out, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (B)(float)in[i];
}

// ---- sampler loops of the transformer networks (fg_dit_sampler_run / fg_wan_sampler_run) --------------------------------------
// The timestep as the embedder consumes it, written per sample / per frame from the device-resident t_list (graph replay with new
// timesteps): t_e = fp32(scale * t) with the product in fp64 (`noise_scheduler.rescale_t` on the sampler's float64 timesteps, then
// `.to(float32)`: DiT.prepare_t, DiT/network.py:457-462; CausalWan._compute_timestep_inputs, Wan/network_causal.py:1063-1075), the SiT
// flip t_e = 1 - t_e (:503-504) and, for the second embedding, r_e likewise with the 'diff' form r_e = t_e - r_e (:520-521) in fp32.
// ti / ri: index into tp, or -1 = the host value tv / rv (f32in: that value is an fp32 tensor in the reference - the cache-fill call's
// `context_noise` - so the product is an fp32 one), ri == -2: no r.
__global__ void embed_times_kernel(const double* __restrict__ tp, int ti, double tv, int ri, double rv, double scale, int sit, int diff,
                                   int f32in, float* __restrict__ te, float* __restrict__ re, int n) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double t = ti >= 0 ? tp[ti] : tv;
    float t_e = f32in ? (float)scale * (float)t : (float)(scale * t);
    if (sit) t_e = 1.0f - t_e;
    te[i] = t_e;
    if (ri != -2 && re) {
        const double r = ri >= 0 ? tp[ri] : rv;
        float r_e = (float)(scale * r);
        if (diff) r_e = t_e - r_e;
        re[i] = r_e;
    }
}
// x0 = x_t - t * flow in fp64, one rounding to fp32 (`flow_to_x0`, noise_schedule.py:975-1004, 1426-1455: the EDM and RF forms
// coincide); sign = -1 folds the SiT convention's `model_output = -model_output` (DiT/network.py:555-558) in (a negation is exact).
__global__ void flow_to_x0_kernel(const float* __restrict__ xt, const float* __restrict__ v, const double* __restrict__ tp, int ti, float sign,
                                  float* __restrict__ out, int64_t total) {
#pragma clang fp contract(off)
    const double t = tp[ti];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)((double)xt[i] - (double)(sign * v[i]) * t);
}
// One Euler step of the flow ODE with optional classifier-free guidance (`DiT._sample_flow`, DiT/network.py:605-651), every operation an
// fp32 one with its own rounding, as torch evaluates `v = v_uncond + g * (v_cond - v_uncond); x = x + dt * v`, dt = fp32(t_next - t):
// v holds [v_uncond | v_cond] (two batches of `total` elements) when cfg, else the one velocity; out2 (nullable) receives a second
// copy of the new x (the doubled batch of the guided call).
__global__ void euler_step_kernel(const float* x, const float* __restrict__ v, const double* __restrict__ tp, int ti, float g, int cfg, float sign,
                                  float* out, float* out2, int64_t total) {  // out may alias x
#pragma clang fp contract(off)
    const float dt = (float)(tp[ti + 1] - tp[ti]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float vv;
        if (cfg) {
            const float vu = sign * v[i], vc = sign * v[total + i];
            const float d = vc - vu;
            const float gd = g * d;
            vv = vu + gd;
        } else {
            vv = sign * v[i];
        }
        const float dx = dt * vv;
        const float nx = x[i] + dx;
        out[i] = nx;
        if (out2) out2[i] = nx;
    }
}
// rows of `run` contiguous elements between a tensor with row pitch src_pitch and one with dst_pitch (elements): the frame slice
// x[:, :, f0:f1] of a [B, C, F, H, W] video <-> a contiguous chunk [B, C, f1 - f0, H, W] (rows = B * C, run = (f1 - f0) * H * W)
__global__ void copy_rows_kernel(const float* __restrict__ src, int64_t src_pitch, float* __restrict__ dst, int64_t dst_pitch, int64_t run,
                                 int64_t rows) {
    const int64_t total = rows * run;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / run, c = i - r * run;
        dst[r * dst_pitch + c] = src[r * src_pitch + c];
    }
}

inline int ew_grid(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

#define RET_LAST() return (int)hipGetLastError()

int launch_gn_coeffs(int dtype, const void* x1, int c1, const void* x2, int c2, const float* gamma, const float* beta,
                     float eps, float2* ab, int batch, int hw, hipStream_t s, float2* mr) {
    const int C = c1 + c2;
    if (C % 4 || (c1 % 4) || C < 16 || C > 2048) return (int)hipErrorInvalidValue;
    const int groups = C / 4 < 32 ? C / 4 : 32;
    if (C % groups || (C / groups) % 4) return (int)hipErrorInvalidValue;
    const int Q = C / 4;
    if (Q > 512) return (int)hipErrorInvalidValue;
    const int threads = (512 / Q) * Q;
    if (dtype)
        hipLaunchKernelGGL(gn_coeffs_kernel<__bf16>, dim3(batch), dim3(threads), 0, s, (const __bf16*)x1, c1, (const __bf16*)x2, c2, gamma, beta, eps, ab, hw, mr);
    else
        hipLaunchKernelGGL(gn_coeffs_kernel<float>, dim3(batch), dim3(threads), 0, s, (const float*)x1, c1, (const float*)x2, c2, gamma, beta, eps, ab, hw, mr);
    RET_LAST();
}

int launch_gn_finalize(const float2* st1, int c1, int s1, const float2* st2, int c2, int s2, const float* gamma,
                       const float* beta, float eps, float2* ab, int batch, int hw, hipStream_t s, float2* mr) {
    const int C = c1 + c2;
    if (C % 4 || (c1 % 4) || C < 16) return (int)hipErrorInvalidValue;
    const int groups = C / 4 < 32 ? C / 4 : 32;
    if (C % groups || (C / groups) % 4) return (int)hipErrorInvalidValue;
    if (C > 512 || (C % 4) || (c1 % 4)) return (int)hipErrorInvalidValue;  // the kernel's LDS partials: 512 / (C / 4) >= 4 slot lanes
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(batch), dim3(512), 0, s, st1, c1, s1, st2, c2, s2, gamma, beta, eps, ab, hw, mr);
    RET_LAST();
}

int launch_precond_coef(const double* t, int t_stride, const double* r, int r_stride, double sigma_data,
                        double sigma_shift, double clamp_min, int drop, float* coef, int B, hipStream_t s) {
    hipLaunchKernelGGL(precond_coef_kernel, dim3((B + 127) / 128), dim3(128), 0, s, t, t_stride, r, r_stride, sigma_data,
                       sigma_shift, clamp_min, drop, coef, B);
    RET_LAST();
}

int launch_mapping_in(const float* c_noise, const float* r_noise, const float* freqs, const float* labels, int label_dim,
                      const float* wl, const float* bl, float* out, int B, int N, int noise_ch, hipStream_t s, const float* aug, const float* wa, int aug_dim) {
    hipLaunchKernelGGL(mapping_in_kernel, dim3((B * N + 255) / 256), dim3(256), 0, s, c_noise, r_noise, freqs, labels,
                       label_dim, wl, bl, out, B, N, noise_ch, aug, wa, aug_dim);
    RET_LAST();
}

int launch_linear(const float* x, const float* w, const float* bias, float* y, int B, int I, int O, int act_silu,
                  hipStream_t s) {
    if (I % 16 == 0 && O % 32 == 0) {
        // wide outputs (the stacked affines): two row tiles per wave halve the weight traffic; otherwise keep the wave count up
        const bool pair = I % 64 == 0 && B > 32 && O >= 2048;
        dim3 g(O / 32, pair ? (B + 63) / 64 : (B + 31) / 32);
        if (pair) {
            if (act_silu)
                hipLaunchKernelGGL((linear_mfma_kernel<1, 64, 2>), g, dim3(64), 0, s, x, w, bias, y, B, I, O);
            else
                hipLaunchKernelGGL((linear_mfma_kernel<0, 64, 2>), g, dim3(64), 0, s, x, w, bias, y, B, I, O);
        } else if (I % 64 == 0) {
            if (act_silu)
                hipLaunchKernelGGL((linear_mfma_kernel<1, 64, 1>), g, dim3(64), 0, s, x, w, bias, y, B, I, O);
            else
                hipLaunchKernelGGL((linear_mfma_kernel<0, 64, 1>), g, dim3(64), 0, s, x, w, bias, y, B, I, O);
        } else if (act_silu) {
            hipLaunchKernelGGL((linear_mfma_kernel<1, 16, 1>), g, dim3(64), 0, s, x, w, bias, y, B, I, O);
        } else {
            hipLaunchKernelGGL((linear_mfma_kernel<0, 16, 1>), g, dim3(64), 0, s, x, w, bias, y, B, I, O);
        }
        RET_LAST();
    }
    dim3 grid((O + 63) / 64, (B + 63) / 64);
    if (act_silu)
        hipLaunchKernelGGL(linear_kernel<1>, grid, dim3(256), 0, s, x, w, bias, y, B, I, O);
    else
        hipLaunchKernelGGL(linear_kernel<0>, grid, dim3(256), 0, s, x, w, bias, y, B, I, O);
    RET_LAST();
}

int launch_conv_in(int dtype, const float* x, const float* c_in, const float* w, const float* bias, void* out, int B,
                   int res, int cin, int cout, hipStream_t s) {
    if (cout % 16) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)cin * 9 * cout * sizeof(float);
    if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
    if (dtype)
        hipLaunchKernelGGL(conv_in_kernel<__bf16>, dim3(B * res), dim3(256), lds, s, x, c_in, w, bias, (__bf16*)out, res, cin, cout);
    else
        hipLaunchKernelGGL(conv_in_kernel<float>, dim3(B * res), dim3(256), lds, s, x, c_in, w, bias, (float*)out, res, cin, cout);
    RET_LAST();
}

int launch_aux_out(int dtype, const void* x, const float2* ab, const float* w, const float* bias, const float* x_t,
                   const float* coef, float* out, int B, int res, int C, int cout, hipStream_t s, float* raw) {
    if (cout > 4 || C % 32) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)9 * cout * C * sizeof(float) + (size_t)C * sizeof(float2);
    if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
    if (dtype)
        hipLaunchKernelGGL((aux_out_kernel<true, __bf16>), dim3(B * res), dim3(256), lds, s, (const __bf16*)x, ab, w, bias, x_t, coef, out, raw, B, res, C, cout);
    else
        hipLaunchKernelGGL((aux_out_kernel<false, float>), dim3(B * res), dim3(256), lds, s, (const float*)x, ab, w, bias, x_t, coef, out, raw, B, res, C, cout);
    RET_LAST();
}

int launch_latents(const float* noise, double tv, const double* tp, int ti, float* out, int64_t total, hipStream_t s) {
    hipLaunchKernelGGL(latents_kernel, dim3(ew_grid(total)), dim3(256), 0, s, noise, tv, tp, ti, out, total);
    RET_LAST();
}
int launch_forward_process(const float* x0, const float* eps, double tv, const double* tp, int ti, int sched, float* out,
                           int64_t total, hipStream_t s) {
    hipLaunchKernelGGL(forward_process_kernel, dim3(ew_grid(total)), dim3(256), 0, s, x0, eps, tv, tp, ti, sched, out, total);
    RET_LAST();
}
int launch_x0_to_eps(const float* xt, const float* x0, double tv, const double* tp, int ti, int sched, double clamp_min,
                     float* out, int64_t total, hipStream_t s) {
    hipLaunchKernelGGL(x0_to_eps_kernel, dim3(ew_grid(total)), dim3(256), 0, s, xt, x0, tv, tp, ti, sched, clamp_min, out, total);
    RET_LAST();
}
int launch_meanflow_update(const float* x, const float* u, const double* tp, int ia, int ib, float* out, int64_t total,
                           hipStream_t s) {
    hipLaunchKernelGGL(meanflow_update_kernel, dim3(ew_grid(total)), dim3(256), 0, s, x, u, tp, ia, ib, out, total);
    RET_LAST();
}
int launch_embed_times(const double* tp, int ti, double tv, int ri, double rv, double scale, int sit, int diff, int f32in, float* te, float* re,
                       int n, hipStream_t s) {
    hipLaunchKernelGGL(embed_times_kernel, dim3((n + 255) / 256), dim3(256), 0, s, tp, ti, tv, ri, rv, scale, sit, diff, f32in, te, re, n);
    RET_LAST();
}
int launch_flow_to_x0(const float* xt, const float* v, const double* tp, int ti, float sign, float* out, int64_t total, hipStream_t s) {
    hipLaunchKernelGGL(flow_to_x0_kernel, dim3(ew_grid(total)), dim3(256), 0, s, xt, v, tp, ti, sign, out, total);
    RET_LAST();
}
int launch_euler_step(const float* x, const float* v, const double* tp, int ti, float g, int cfg, float sign, float* out, float* out2,
                      int64_t total, hipStream_t s) {
    hipLaunchKernelGGL(euler_step_kernel, dim3(ew_grid(total)), dim3(256), 0, s, x, v, tp, ti, g, cfg, sign, out, out2, total);
    RET_LAST();
}
int launch_copy_rows(const float* src, int64_t src_pitch, float* dst, int64_t dst_pitch, int64_t run, int64_t rows, hipStream_t s) {
    hipLaunchKernelGGL(copy_rows_kernel, dim3(ew_grid(rows * run)), dim3(256), 0, s, src, src_pitch, dst, dst_pitch, run, rows);
    RET_LAST();
}
int launch_randn(float* out, int64_t total, uint64_t seed, uint64_t offset, const uint64_t* seed_dev, hipStream_t s) {
    hipLaunchKernelGGL(randn_kernel, dim3(ew_grid((total + 3) / 4)), dim3(256), 0, s, out, total, seed, offset, seed_dev);
    RET_LAST();
}
int launch_gn_silu_pool(int dtype, const void* x, const float2* ab, void* out, int B, int H, int W, int C, hipStream_t s) {
    if (C % 8) return (int)hipErrorInvalidValue;
    const int64_t total = (int64_t)B * H * W * (C / 8);
    if (dtype)
        hipLaunchKernelGGL((gn_silu_pool_kernel<__bf16, true>), dim3(ew_grid(total)), dim3(256), 0, s, (const __bf16*)x, ab, (__bf16*)out, B, H, W, C);
    else
        hipLaunchKernelGGL((gn_silu_pool_kernel<float, false>), dim3(ew_grid(total)), dim3(256), 0, s, (const float*)x, ab, (float*)out, B, H, W, C);
    RET_LAST();
}
int launch_to_act(int dtype, const float* in, void* out, int64_t total, hipStream_t s) {
    if (dtype)
        hipLaunchKernelGGL((convert_kernel<float, __bf16>), dim3(ew_grid(total)), dim3(256), 0, s, in, (__bf16*)out, total);
    else
        hipLaunchKernelGGL((convert_kernel<float, float>), dim3(ew_grid(total)), dim3(256), 0, s, in, (float*)out, total);
    RET_LAST();
}
int launch_from_act(int dtype, const void* in, float* out, int64_t total, hipStream_t s) {
    if (dtype)
        hipLaunchKernelGGL((convert_kernel<__bf16, float>), dim3(ew_grid(total)), dim3(256), 0, s, (const __bf16*)in, out, total);
    else
        hipLaunchKernelGGL((convert_kernel<float, float>), dim3(ew_grid(total)), dim3(256), 0, s, (const float*)in, out, total);
    RET_LAST();
}
// feature tap: NHWC activation (compute dtype) -> NCHW fp32, 32 x 32 tiles through LDS (both sides coalesced)
template <typename AT>
__global__ __launch_bounds__(256) void nhwc_act_to_nchw_kernel(const AT* __restrict__ in, float* __restrict__ out, int C, int HW) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) tile[i][tx] = (float)in[((size_t)b * HW + p0 + i) * C + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) out[((size_t)b * C + c0 + i) * HW + p0 + tx] = tile[tx][i];
}
int launch_act_to_nchw(int dtype, const void* in, float* out, int B, int C, int HW, hipStream_t s) {
    dim3 grid(HW / 32, C / 32, B);
    if (dtype)
        hipLaunchKernelGGL(nhwc_act_to_nchw_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)in, out, C, HW);
    else
        hipLaunchKernelGGL(nhwc_act_to_nchw_kernel<float>, grid, dim3(256), 0, s, (const float*)in, out, C, HW);
    RET_LAST();
}
// Samples -> image bytes, the step that follows generator_fn in the reference's sample writer
// (scripts/fid/compute_fid_from_ckpts.py:199): (x * 127.5 + 128).clip(0, 255).to(uint8).permute(0, 2, 3, 1).
// fp32 multiply then add (two roundings, as torch evaluates it - no FMA contraction), clamp, truncate; NaN maps to 0.
// One thread per pixel: C coalesced plane reads, C consecutive bytes out.
__global__ __launch_bounds__(256) void images_to_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int64_t npix,
                                                           int C, int HW) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int64_t n = i / HW;
    const int p = (int)(i - n * HW);
    const float* src = x + n * C * HW + p;
    uint8_t* dst = out + i * C;
    for (int c = 0; c < C; ++c) {
        float v = src[(int64_t)c * HW] * 127.5f;
        v = v + 128.0f;
        v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);  // false for NaN on both sides ...
        dst[c] = v == v ? (uint8_t)(int)v : (uint8_t)0;  // ... which is written as 0
    }
}
int launch_images_to_u8(const float* x, uint8_t* out, int64_t B, int C, int HW, hipStream_t s) {
    const int64_t npix = B * HW;
    if (npix <= 0) return 0;
    hipLaunchKernelGGL(images_to_u8_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, x, out, npix, C, HW);
    RET_LAST();
}
int launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int HW, hipStream_t s) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid((int64_t)B * C * HW)), dim3(256), 0, s, in, out, B, C, HW);
    RET_LAST();
}
