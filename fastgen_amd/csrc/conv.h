// Fused GroupNorm-apply + SiLU + Conv2d (3x3 / 1x1) implicit-GEMM kernel: host-visible argument block.
#pragma once
#include <hip/hip_runtime.h>

struct ConvArgs {
    // A operand: NHWC activations; the input channels are the virtual concat [src1 (C1) | src2 (C2)]
    // (torch.cat at EDM/network.py:560 never materialises).  Stored in the compute dtype.
    const void* src1;
    const void* src2;
    int C1, C2;
    int Hs, Ws;  // source spatial size (before the 2x resample folded into the load)
    int H, W;    // output spatial size (H == W, W in {8,16,32})
    int B;
    const float2* ab;   // [B][C1+C2] GroupNorm coefficients y = a*x+b, or nullptr
    const void* wpack;  // packed weights (compute dtype), see pack_conv_weights_kernel
    const void* wpack_ws;  // the same weights in the layout of conv_ws.hip (bf16 3x3 convs it supports), or nullptr
    const float* bias;  // [Cout] or nullptr
    const float* temb;  // [B][temb_stride] per-image per-channel additive term (affine(emb)), or nullptr
    int temb_stride;
    const void* resid;   // [B,H,W,Cout] residual (compute dtype) added before `scale`, or nullptr
    float scale;
    void* out;  // [B,H,W,Cout] in the compute dtype (OUT_NHWC)
    int Cout;
    // GroupNorm partial statistics of the OUTPUT tensor, written by the epilogue (or nullptr):
    // stats[b][slot][c/4] = {sum, sum of squares} over the tile's pixels of channels 4q..4q+3; slots per image =
    // conv_stat_slots(W).  gn_finalize turns them into the next norm's coefficients without re-reading the tensor.
    float2* stats;
    int dbg;  // ablation flags for scripts/conv_ablate.py (0 in production): 1 no staging, 2 no weight refill, 4 no epilogue, 8 no MFMA
    // OUT_QKV: compute-dtype planes q,k: [B][HW][256]; vt: [B][256][HW]
    void* q_out;
    void* k_out;
    void* vt_out;
    // Transformer GEMMs (OUT_TOK / OUT_HEADS: the DiT blocks, dit.hip): the same kernel with 128-column output tiles.
    // epilogue: v = acc + bias; act 1: v = gelu_tanh(v); gate: v *= gate[n][co]; + resid; * scale
    const float* gate = nullptr;  // [B][gate_stride] per-image, per-channel multiplier (the adaLN gates), or nullptr
    int gate_stride = 0;
    int act = 0;                  // 0 none, 1 GELU(tanh)
    int heads = 0, head_dim = 0;  // OUT_HEADS: Cout = 3 * heads * head_dim; q_out, k_out [B][heads][T][head_dim], vt_out [B][heads][head_dim][T]
    size_t heads_lo_off = 0;      // ... in the bf16x3 mode each of the three is TWO bf16 planes: hi at the pointer, lo this many elements behind
};

enum { PRO_NONE = 0, PRO_GN = 1, PRO_GN_SILU = 2 };
enum { RES_NONE = 0, RES_DOWN = 1, RES_UP = 2 };
enum { OUT_NHWC = 0, OUT_QKV = 1, OUT_TOK = 2, OUT_HEADS = 3 };

// dtype: 0 fp32, 1 bf16 — also the storage type of every activation tensor (src1/src2/resid/out); 2 = split-bf16 arithmetic
// (common.h bf16x3) on fp32 tensors.
// Returns hipError_t as int.
int launch_conv_fused(int dtype, int ks, int pro, int res, int outmode, const ConvArgs& a, hipStream_t stream);
// partial-statistics slots per image for an output of width W (= pixel tiles per image, 1 when a tile spans images)
int conv_stat_slots(int W);
// wave-specialised persistent kernel for the dominant bf16 3x3 shapes (conv_ws.hip); launch_conv_fused dispatches to it
bool conv_ws_supported(int dtype, int ks, int pro, int res, int outmode, const ConvArgs& a);
bool conv_ws_enabled();
int launch_conv_ws(int res, const ConvArgs& a, hipStream_t stream, bool prepare_only, int pro = PRO_GN_SILU);
bool conv_ws_shape_ok(int dtype, int cout, int cin, int res);
int launch_pack_conv_weights_ws(const float* w_oihw, void* wpack_ws, int cout, int cin, hipStream_t stream);
// conv_ws3.hip: the wave-specialised kernel of the split-bf16 mode (dtype 2), same dispatch rule
bool conv_x3ws_supported(int ks, int pro, int res, int outmode, const ConvArgs& a);
bool conv_x3ws_shape_ok(int cout, int cin, int res);
int conv_x3ws_stat_slots(int W);
int launch_conv_x3ws(int res, const ConvArgs& a, hipStream_t stream, bool prepare_only, int pro = PRO_GN_SILU);
int launch_pack_conv_weights_x3ws(const float* w_oihw, void* wpack_ws, int cout, int cin, hipStream_t stream);
// statistics slots per image that launch_conv_fused will write for this call (depends on the kernel the dispatch picks)
int conv_launch_stat_slots(int dtype, int ks, int pro, int res, int outmode, const ConvArgs& a);
int launch_conv_ws_debug(const ConvArgs& a, int abl, hipStream_t stream);  // ablation builds of the 32x32 shape
int launch_conv_debug(int dtype, const ConvArgs& a, hipStream_t stream);  // ablation build, honours a.dbg
// set the dynamic-LDS attribute of every instantiation of this dtype (call once, outside stream capture)
int conv_prepare_all(int dtype);
// elements of packed weight storage for a conv with these dims
size_t conv_pack_elems(int cout, int cin, int ks);
// pack OIHW fp32 -> fragment order.  qkv_perm != 0 applies the q|k|v de-interleave of EDM/network.py:290-294.
int launch_pack_conv_weights(int dtype, const float* w_oihw, void* wpack, int cout, int cin, int ks, int qkv_perm,
                             hipStream_t stream);
