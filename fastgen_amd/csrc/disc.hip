// Discriminator_EDM heads (reference fastgen/networks/discriminators.py:62-137) - the last piece of the DMD2 training step
// (SURVEY §8(f)1).  One head per tapped resolution R in {32, 16, 8} on a [B, 256, R, R] feature map of the teacher's encoder:
//
//   while R > 8:  Conv2d(4x4, stride 2, pad 1) -> GroupNorm(32) -> SiLU        (R -> R/2)
//   Conv2d(4x4, stride 2, pad 1) -> GroupNorm -> SiLU                          (8 -> 4)
//   Conv2d(4x4, stride 4, pad 0) -> GroupNorm -> SiLU                          (4 -> 1)
//   Conv2d(1x1, 256 -> 1)                                                      -> one logit per image
//
// 0.1 % of the step's FLOPs, so the layers are composed from pieces that exist: a strided conv is im2col (bf16, K ordered
// (ky, kx, ci) so that a row is written in 16-byte pieces) + the small NT matrix product of attn_bwd.hip (weights permuted to the
// same K order), GroupNorm + SiLU and their backward are the kernels of the U-Net (misc.hip / bwd.hip), the data gradient is
// the product with the transposed weights followed by a gathering col2im, the weight gradient is dY^T cols.  fp32 parameters and
// parameter gradients (reference layouts), bf16 activations, GroupNorm eps = 1e-5 (torch's default, as in the reference).
// disc_run() does forward (-> logits) and, when dlogits is given, the backward (d/dfeat and accumulated parameter gradients),
// recomputing nothing: everything lives in the caller's workspace for the duration of the call.
#include <algorithm>

#include "common.h"
#include "misc.h"

namespace {

constexpr int DC = 256;          // channels (in_channels of the reference's default)
constexpr int DK = DC * 16;      // K of a 4x4 conv
constexpr float DEPS = 1e-5f;

// NCHW fp32 -> NHWC bf16
__global__ void nchw_to_nhwc_bf16_kernel(const float* __restrict__ in, __bf16* __restrict__ out, int C, int HW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int64_t n = pix / HW, p = pix - n * HW;
        out[i] = (__bf16)in[(n * C + c) * HW + p];
    }
}
// cols[(n, oy, ox)][(ky, kx, ci)] = x[n, oy*st - pad + ky, ox*st - pad + kx, ci]; rows >= M (padding of M to 32) are zero
__global__ void im2col_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ cols, int R, int Ro, int st, int pad, int M,
                              int64_t total_oct) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_oct; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % (DC / 8));
        const int tap = (int)((i / (DC / 8)) % 16);
        const int64_t m = i / (DC / 8 * 16);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
        if (m < M) {
            const int ox = (int)(m % Ro), oy = (int)((m / Ro) % Ro);
            const int64_t n = m / ((int64_t)Ro * Ro);
            const int y = oy * st - pad + (tap >> 2), xx = ox * st - pad + (tap & 3);
            if (y >= 0 && y < R && xx >= 0 && xx < R) v = *reinterpret_cast<const bf16x8*>(x + ((n * R + y) * R + xx) * DC + c8 * 8);
        }
        *reinterpret_cast<bf16x8*>(cols + m * DK + tap * DC + c8 * 8) = v;
    }
}
// dx[n, y, x, c] = sum over the windows (oy, ox, ky, kx) that read this pixel of dcols[(n, oy, ox)][(ky, kx, c)]
__global__ void col2im_kernel(const __bf16* __restrict__ dcols, __bf16* __restrict__ dx, int R, int Ro, int st, int pad,
                              int64_t total_oct) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_oct; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % (DC / 8));
        const int64_t pix = i / (DC / 8);
        const int xx = (int)(pix % R), y = (int)((pix / R) % R);
        const int64_t n = pix / ((int64_t)R * R);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int ky = 0; ky < 4; ++ky) {
            const int ty = y + pad - ky;
            if (ty < 0 || ty % st) continue;
            const int oy = ty / st;
            if (oy >= Ro) continue;
            for (int kx = 0; kx < 4; ++kx) {
                const int tx = xx + pad - kx;
                if (tx < 0 || tx % st) continue;
                const int ox = tx / st;
                if (ox >= Ro) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(dcols + ((n * Ro + oy) * Ro + ox) * DK + (ky * 4 + kx) * DC + c8 * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)acc[j];
        *reinterpret_cast<bf16x8*>(dx + pix * DC + c8 * 8) = o;
    }
}
// wb[n][(ky,kx,ci)] = bf16(W[n][ci][ky][kx]);  wbt[(ky,kx,ci)][n] = the same, transposed (for the data gradient)
__global__ void prep_w_kernel(const float* __restrict__ w, __bf16* __restrict__ wb, __bf16* __restrict__ wbt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= DC * DK) return;
    const int k = i % DK, n = i / DK;
    const int tap = k / DC, ci = k % DC;
    const __bf16 v = (__bf16)w[((size_t)n * DC + ci) * 16 + tap];
    wb[i] = v;
    wbt[(size_t)k * DC + n] = v;
}
// dW[n][ci][ky][kx] += dwp[n][(ky,kx,ci)]
__global__ void unperm_add_kernel(const float* __restrict__ dwp, float* __restrict__ dw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= DC * DK) return;
    const int tap = i % 16, ci = (i / 16) % DC, n = i / (16 * DC);
    dw[i] += dwp[(size_t)n * DK + tap * DC + ci];
}
// y[m][n] = bf16(acc[m][n] + bias[n])
__global__ void bias_bf16_kernel(const float* __restrict__ acc, const float* __restrict__ bias, __bf16* __restrict__ y, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        y[i] = (__bf16)(acc[i] + bias[i % DC]);
}
// logits[b] = sum_c a[b][c] w[c] + b0;  one wave per image
__global__ __launch_bounds__(64) void logit_kernel(const __bf16* __restrict__ a, const float* __restrict__ w, const float* __restrict__ b0,
                                                   float* __restrict__ logits) {
    const int b = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < DC; c += 64) s = fmaf((float)a[(size_t)b * DC + c], w[c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) logits[b] = s + b0[0];
}
// backward of the logit layer: da[b][c] = dl[b] w[c];  dw[c] += sum_b dl[b] a[b][c];  db += sum_b dl[b]   (one workgroup)
__global__ __launch_bounds__(256) void logit_bwd_kernel(const __bf16* __restrict__ a, const float* __restrict__ w,
                                                        const float* __restrict__ dl, __bf16* __restrict__ da, float* __restrict__ dw,
                                                        float* __restrict__ db, int B) {
    const int c = threadIdx.x;
    float g = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = dl[b];
        da[(size_t)b * DC + c] = (__bf16)(d * w[c]);
        g = fmaf(d, (float)a[(size_t)b * DC + c], g);
        sb += d;
    }
    if (dw) dw[c] += g;
    if (db && c == 0) db[0] += sb;
}

inline unsigned eb(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}
inline int stages_of(int res) { return res == 32 ? 4 : (res == 16 ? 3 : (res == 8 ? 2 : 0)); }  // strided convs incl. the 4 -> 1 one
inline int rup32(int64_t m) { return (int)((m + 31) / 32 * 32); }

struct Stage {
    int R, Ro, st, pad, M, Mp;
    __bf16 *x, *cols, *y, *act, *wb, *wbt;
    float* accf;
    float2 *ab, *mr;
};

}  // namespace

// parameters per head, in the reference's module order: per strided conv {weight, bias, gn.weight, gn.bias}, then the 1x1 conv
// {weight, bias}
int disc_num_params(int res) { return stages_of(res) ? 4 * stages_of(res) + 2 : 0; }

static size_t disc_plan(int res, int B, char* base, Stage* st, __bf16** xin, __bf16** colsT, __bf16** dyT, __bf16** dcols,
                        __bf16** da, __bf16** dy, float** dwp, float2** P, float2** S, float** vec) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        char* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    };
    const int ns = stages_of(res);
    int R = res;
    size_t max_mk = 0, max_x = (size_t)B * res * res * DC;
    *xin = (__bf16*)take(max_x * 2);
    for (int i = 0; i < ns; ++i) {
        Stage& g = st[i];
        g.R = R;
        g.st = (i == ns - 1) ? 4 : 2;
        g.pad = (i == ns - 1) ? 0 : 1;
        g.Ro = (i == ns - 1) ? 1 : R / 2;
        g.M = B * g.Ro * g.Ro;
        g.Mp = rup32(g.M);
        g.cols = (__bf16*)take((size_t)g.Mp * DK * 2);
        g.accf = (float*)take((size_t)g.Mp * DC * 4);
        g.y = (__bf16*)take((size_t)g.Mp * DC * 2);
        g.act = (__bf16*)take((size_t)g.Mp * DC * 2);
        g.wb = (__bf16*)take((size_t)DC * DK * 2);
        g.wbt = (__bf16*)take((size_t)DC * DK * 2);
        g.ab = (float2*)take((size_t)B * DC * sizeof(float2));
        g.mr = (float2*)take((size_t)B * 32 * sizeof(float2));
        max_mk = std::max(max_mk, (size_t)g.Mp * DK);
        R = g.Ro;
    }
    *colsT = (__bf16*)take(max_mk * 2);
    *dcols = (__bf16*)take(max_mk * 2);
    *dyT = (__bf16*)take((size_t)rup32((int64_t)B * (res / 2) * (res / 2)) * DC * 2);
    *da = (__bf16*)take(max_x * 2);
    *dy = (__bf16*)take(max_x * 2);
    *dwp = (float*)take((size_t)DC * DK * 4);
    *P = (float2*)take((size_t)B * DC * sizeof(float2));
    *S = (float2*)take((size_t)B * 32 * sizeof(float2));
    *vec = (float*)take((size_t)B * DC * 4);
    return off;
}

size_t disc_workspace_bytes(int res, int B) {
    if (!stages_of(res) || B <= 0) return 0;
    Stage st[4];
    __bf16 *a, *b, *c, *d, *e, *f;
    float *g, *v;
    float2 *P, *S;
    return disc_plan(res, B, nullptr, st, &a, &b, &c, &d, &e, &f, &g, &P, &S, &v);
}

// feat [B,256,res,res] NCHW fp32 -> logits [B].  dlogits != nullptr: also the backward; dfeat (nullable) [B,256,res,res] NCHW fp32
// is overwritten, grads[i] (nullable entries, same order and shapes as params) are accumulated.
int disc_run(const float* feat, int res, const float* const* params, float* logits, const float* dlogits, float* dfeat,
             float* const* grads, int B, void* ws, hipStream_t s) {
    const int ns = stages_of(res);
    if (!ns || B <= 0) return (int)hipErrorInvalidValue;
    Stage st[4];
    __bf16 *xin, *colsT, *dyT, *dcols, *da, *dy;
    float *dwp, *vec;
    float2 *P, *S;
    disc_plan(res, B, (char*)ws, st, &xin, &colsT, &dyT, &dcols, &da, &dy, &dwp, &P, &S, &vec);
    hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3(eb((int64_t)B * res * res * DC)), dim3(256), 0, s, feat, xin, DC, res * res,
                       (int64_t)B * res * res * DC);
    int rc;
    const __bf16* x = xin;
    for (int i = 0; i < ns; ++i) {
        Stage& g = st[i];
        const float *W = params[4 * i], *bias = params[4 * i + 1], *gam = params[4 * i + 2], *bet = params[4 * i + 3];
        g.x = const_cast<__bf16*>(x);
        hipLaunchKernelGGL(prep_w_kernel, dim3((DC * DK + 255) / 256), dim3(256), 0, s, W, g.wb, g.wbt);
        hipLaunchKernelGGL(im2col_kernel, dim3(eb((int64_t)g.Mp * 16 * (DC / 8))), dim3(256), 0, s, x, g.cols, g.R, g.Ro, g.st, g.pad, g.M,
                           (int64_t)g.Mp * 16 * (DC / 8));
        if ((rc = launch_nt_gemm(g.cols, g.wb, g.accf, g.Mp, DC, DK, 1.0f, 0, s))) return rc;
        hipLaunchKernelGGL(bias_bf16_kernel, dim3(eb((int64_t)g.M * DC)), dim3(256), 0, s, g.accf, bias, g.y, (int64_t)g.M * DC);
        if ((rc = launch_gn_coeffs(1, g.y, DC, nullptr, 0, gam, bet, DEPS, g.ab, B, g.Ro * g.Ro, s, g.mr))) return rc;
        if ((rc = launch_gn_act(1, 0, g.y, DC, nullptr, 0, g.ab, g.act, B, g.Ro, 0, s))) return rc;
        x = g.act;
    }
    const float *w1 = params[4 * ns], *b1 = params[4 * ns + 1];
    hipLaunchKernelGGL(logit_kernel, dim3(B), dim3(64), 0, s, x, w1, b1, logits);
    if (!dlogits) return (int)hipGetLastError();
    // ---- backward ----------------------------------------------------------------------------------------------------------
    hipLaunchKernelGGL(logit_bwd_kernel, dim3(1), dim3(256), 0, s, x, w1, dlogits, da, grads ? grads[4 * ns] : nullptr,
                       grads ? grads[4 * ns + 1] : nullptr, B);
    for (int i = ns - 1; i >= 0; --i) {
        Stage& g = st[i];
        float *gW = grads ? grads[4 * i] : nullptr, *gb = grads ? grads[4 * i + 1] : nullptr;
        float *gg = grads ? grads[4 * i + 2] : nullptr, *gbt = grads ? grads[4 * i + 3] : nullptr;
        // GroupNorm + SiLU: da (gradient of the activation) -> dy (gradient of the conv output); rows beyond M stay zero
        if (g.Mp != g.M) (void)hipMemsetAsync(dy, 0, (size_t)g.Mp * DC * 2, s);
        if ((rc = launch_gn_bwd(1, 0, g.y, DC, nullptr, 0, da, DC, g.ab, g.mr, params[4 * i + 2], P, S, gg, gbt, nullptr, 0, 0.f, dy, B, g.Ro, 0, s)))
            return rc;
        if (gb) {
            if ((rc = launch_colsum(1, dy, DC, DC, vec, B, g.Ro * g.Ro, 1.0f, s))) return rc;
            if ((rc = launch_batchsum_add(vec, gb, B, DC, s))) return rc;
        }
        if (gW) {  // dWp[n][k] = sum_m dy[m][n] cols[m][k]
            if ((rc = launch_transpose_bf16(dy, dyT, g.Mp, DC, s))) return rc;
            if ((rc = launch_transpose_bf16(g.cols, colsT, g.Mp, DK, s))) return rc;
            if ((rc = launch_nt_gemm(dyT, colsT, dwp, DC, DK, g.Mp, 1.0f, 0, s))) return rc;
            hipLaunchKernelGGL(unperm_add_kernel, dim3((DC * DK + 255) / 256), dim3(256), 0, s, dwp, gW);
        }
        if (i > 0 || dfeat) {  // dcols[m][k] = sum_n dy[m][n] W[n][k];  dx = col2im(dcols)
            if ((rc = launch_nt_gemm(dy, g.wbt, dcols, g.Mp, DK, DC, 1.0f, 1, s))) return rc;
            hipLaunchKernelGGL(col2im_kernel, dim3(eb((int64_t)B * g.R * g.R * (DC / 8))), dim3(256), 0, s, dcols, da, g.R, g.Ro, g.st, g.pad,
                               (int64_t)B * g.R * g.R * (DC / 8));
        }
    }
    if (dfeat) {
        if ((rc = launch_act_to_nchw(1, da, dfeat, B, DC, res * res, s))) return rc;
    }
    return (int)hipGetLastError();
}
