// Fused GroupNorm-apply + SiLU + Conv2d (3x3 / 1x1) implicit-GEMM kernel: host-visible argument block.
#pragma once
#include <hip/hip_runtime.h>

struct ConvArgs {
    // A operand: NHWC fp32 activations; the input channels are the virtual concat [src1 (C1) | src2 (C2)]
    // (torch.cat at EDM/network.py:560 never materialises).
    const float* src1;
    const float* src2;
    int C1, C2;
    int Hs, Ws;  // source spatial size (before the 2x resample folded into the load)
    int H, W;    // output spatial size (H == W, W in {8,16,32})
    int B;
    const float2* ab;   // [B][C1+C2] GroupNorm coefficients y = a*x+b, or nullptr
    const void* wpack;  // packed weights (compute dtype), see pack_conv_weights_kernel
    const float* bias;  // [Cout] or nullptr
    const float* temb;  // [B][temb_stride] per-image per-channel additive term (affine(emb)), or nullptr
    int temb_stride;
    const float* resid;  // [B,H,W,Cout] residual added before `scale`, or nullptr
    float scale;
    float* out;  // [B,H,W,Cout] fp32 (OUT_NHWC)
    int Cout;
    // OUT_QKV: compute-dtype planes q,k: [B][HW][256]; vt: [B][256][HW]
    void* q_out;
    void* k_out;
    void* vt_out;
};

enum { PRO_NONE = 0, PRO_GN = 1, PRO_GN_SILU = 2 };
enum { RES_NONE = 0, RES_DOWN = 1, RES_UP = 2 };
enum { OUT_NHWC = 0, OUT_QKV = 1 };

// dtype: 0 fp32, 1 bf16.  Returns hipError_t as int.
int launch_conv_fused(int dtype, int ks, int pro, int res, int outmode, const ConvArgs& a, hipStream_t stream);
// set the dynamic-LDS attribute of every instantiation of this dtype (call once, outside stream capture)
int conv_prepare_all(int dtype);
// elements of packed weight storage for a conv with these dims
size_t conv_pack_elems(int cout, int cin, int ks);
// pack OIHW fp32 -> fragment order.  qkv_perm != 0 applies the q|k|v de-interleave of EDM/network.py:290-294.
int launch_pack_conv_weights(int dtype, const float* w_oihw, void* wpack, int cout, int cin, int ks, int qkv_perm,
                             hipStream_t stream);
