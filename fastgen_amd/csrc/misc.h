// Launchers of the small kernels (misc.hip) and attention (attn.hip).  All return hipError_t as int.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// dtype (0 fp32 / 1 bf16) is the storage type of activation tensors passed as void*
int launch_gn_coeffs(int dtype, const void* x1, int c1, const void* x2, int c2, const float* gamma, const float* beta,
                     float eps, float2* ab, int batch, int hw, hipStream_t s, float2* mr = nullptr);  // mr[b][group] = {mean, rstd}
int launch_gn_finalize(const float2* st1, int c1, int s1, const float2* st2, int c2, int s2, const float* gamma,
                       const float* beta, float eps, float2* ab, int batch, int hw, hipStream_t s, float2* mr = nullptr);
int launch_precond_coef(const double* t, int t_stride, const double* r, int r_stride, double sigma_data,
                        double sigma_shift, double clamp_min, int drop, float* coef, int B, hipStream_t s);
int launch_mapping_in(const float* c_noise, const float* r_noise, const float* freqs, const float* labels, int label_dim,
                      const float* wl, const float* bl, float* out, int B, int N, int noise_ch, hipStream_t s, const float* aug = nullptr, const float* wa = nullptr, int aug_dim = 0);
int launch_linear(const float* x, const float* w, const float* bias, float* y, int B, int I, int O, int act_silu,
                  hipStream_t s);
int launch_conv_in(int dtype, const float* x, const float* c_in, const float* w, const float* bias, void* out, int B,
                   int res, int cin, int cout, hipStream_t s);
int launch_aux_out(int dtype, const void* x, const float2* ab, const float* w, const float* bias, const float* x_t,
                   const float* coef, float* out, int B, int res, int C, int cout, hipStream_t s, float* raw = nullptr);
int launch_latents(const float* noise, double tv, const double* tp, int ti, float* out, int64_t total, hipStream_t s);
int launch_forward_process(const float* x0, const float* eps, double tv, const double* tp, int ti, int sched, float* out,
                           int64_t total, hipStream_t s);
int launch_x0_to_eps(const float* xt, const float* x0, double tv, const double* tp, int ti, int sched, double clamp_min,
                     float* out, int64_t total, hipStream_t s);
int launch_meanflow_update(const float* x, const float* u, const double* tp, int ia, int ib, float* out, int64_t total,
                           hipStream_t s);
int launch_randn(float* out, int64_t total, uint64_t seed, uint64_t offset, const uint64_t* seed_dev, hipStream_t s);
// sampler loops of the transformer networks (misc.hip; engine_sampler.inc)
int launch_embed_times(const double* tp, int ti, double tv, int ri, double rv, double scale, int sit, int diff, int f32in, float* te, float* re,
                       int n, hipStream_t s);
int launch_flow_to_x0(const float* xt, const float* v, const double* tp, int ti, float sign, float* out, int64_t total, hipStream_t s);
int launch_euler_step(const float* x, const float* v, const double* tp, int ti, float g, int cfg, float sign, float* out, float* out2,
                      int64_t total, hipStream_t s);
int launch_copy_rows(const float* src, int64_t src_pitch, float* dst, int64_t dst_pitch, int64_t run, int64_t rows, hipStream_t s);
// out[B,H,W,C] = mean over 2x2 of silu(a*x+b), x [B,2H,2W,C]
int launch_gn_silu_pool(int dtype, const void* x, const float2* ab, void* out, int B, int H, int W, int C, hipStream_t s);
int launch_to_act(int dtype, const float* in, void* out, int64_t total, hipStream_t s);
int launch_from_act(int dtype, const void* in, float* out, int64_t total, hipStream_t s);
// dropout of conv1's operand in training mode: probability, block index (its own Philox stream), seed of the call; p == 0: off
struct DropArgs {
    float p = 0.f;
    uint32_t block = 0;
    uint64_t seed = 0;
};
int launch_dropout_mask(float* out, int64_t total, DropArgs drop, hipStream_t s);
// bwd.hip: elementwise / reduction pieces of the block backward pass; dtype = storage type of the activation / gradient tensors
// (1 bf16, 0 fp32), as everywhere in this header
int launch_gn_act(int dtype, int mode, const void* x1, int c1, const void* x2, int c2, const float2* ab, void* out, int B, int res, int rm,
                  hipStream_t s, DropArgs drop = DropArgs{});  // res = output resolution; rm 0 none, 1 down (avg 2x2), 2 up (nearest)
int launch_gn_bwd(int dtype, int mode, const void* x1, int c1, const void* x2, int c2, const void* dact, int cd, const float2* ab,
                  const float2* mr, const float* gamma, float2* P, float2* S, float* dgamma, float* dbeta, const void* add, int ca,
                  float add_scale, void* dx, int B, int res, int rm, hipStream_t s, void* dx2 = nullptr, int accumulate = 0,
                  DropArgs drop = DropArgs{});
                  // res = the norm's (input) resolution; dx2: separate dense tensor for the second concat source; accumulate: +=
int launch_colsum(int dtype, const void* t, int ct, int C, float* out, int B, int hw, float scale, hipStream_t s, int out_stride = 0);
int launch_transpose_f32(const float* in, float* out, int R, int Cc, hipStream_t s);
int launch_batchsum_add(const float* in, float* out, int B, int C, hipStream_t s, float* out2 = nullptr, int in_stride = 0);  // in row stride (0 = C)
int launch_scale_to_act(int dtype, const float* in, void* out, float scale, int64_t total, hipStream_t s);
int launch_slice_to_f32(int dtype, const void* in, int cs, int c_off, float* out, int C, int64_t npix, hipStream_t s);
int launch_affine_bwd(const float* dtemb, const float* emb, const float* w, float* dw, float* demb, int B, int C, int K, hipStream_t s);
int launch_dgrad_weights(const float* w, float* wt, int cout, int cin, int cin_pad, int taps, hipStream_t s);
int launch_head_grad(int dtype, const float* dout, const float* c_out, void* out, int B, int C, int Cp, int hw, hipStream_t s);
int launch_stem_operand(int dtype, const float* x, const float* c_in, void* out, int B, int C, int Cp, int hw, hipStream_t s);
int launch_add_sub_tensor(const float* src, int Is, float* dst, int O, int I, int T, hipStream_t s);
int launch_pad_rows(const float* src, float* dst, int O, int Op, int IT, hipStream_t s);
int launch_add_act(int dtype, void* dst, const void* src, int64_t total, hipStream_t s);
int launch_silu_bwd(const float* dy, const float* pre, float* dpre, int total, hipStream_t s);
int launch_linear_bwd(const float* dy, const float* x, const float* w, float* dw, float* db, float* dx, int B, int C, int K,
                      float scale, hipStream_t s, int dy_stride = 0);
int launch_add_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int hw, hipStream_t s);
int launch_input_grad(int dtype, const void* da, int cd, const float* c_in, const float* c_skip, const float* dout, float* dx, int B, int C, int hw,
                      hipStream_t s);
int launch_gn_jvp(int dtype, int mode, const void* x1, int c1, const void* x2, int c2, const void* xd, const float2* ab, const float2* mr,
                  float2* P, float2* S, void* out, int B, int res, hipStream_t s, DropArgs drop = DropArgs{});
int launch_jvp_coef(const double* t, const double* r, const float* vt, const float* vr, double sigma_data, double sigma_shift, int drop,
                    float* ct, int B, hipStream_t s);
int launch_jvp_embed(const float* c_noise, const float* r_noise, const float* dc, const float* dr, const float* freqs, float* out, int B,
                     int N, int noise_ch, hipStream_t s);
int launch_jvp_input(const float* vx, const float* x, const float* c_in, const float* dc_in, float* out, int B, int chw, hipStream_t s);
int launch_jvp_output(int dtype, const void* fd, int cf, const float* out, const float* x, const float* vx, const float* ct, float* jvp, int B, int C,
                      int hw, hipStream_t s);
int launch_fill_f32(float* p, float v, int n, hipStream_t s);
int launch_fill_f2(float2* p, float a, float b, int n, hipStream_t s);
int launch_attention_jvp(int dtype, const void* q, const void* k, const void* vt, const void* qd, const void* kd, const void* vtd, void* od,
                         void* scratch, int B, int T, int C, hipStream_t s);
// attn_bwd.hip
size_t attention_backward_scratch_bytes(int dtype, int B, int T, int C);
int launch_attention_backward(int dtype, const void* q, const void* k, const void* vt, const void* dO, void* dq, void* dk, void* dvt,
                              void* scratch, int B, int T, int C, hipStream_t s);
int launch_qkv_interleave(int dtype, const void* dq, const void* dk, const void* dvt, void* out, int B, int T, int C, hipStream_t s);
int launch_nt_gemm(const void* A, const void* Bm, void* C, int M, int N, int K, float scale, int out_bf16, hipStream_t s);
int launch_transpose_bf16(const void* in, void* out, int R, int Cc, hipStream_t s);
// disc.hip: Discriminator_EDM heads (networks/discriminators.py:62-137)
int disc_num_params(int res);
size_t disc_workspace_bytes(int res, int B);
int disc_run(const float* feat, int res, const float* const* params, float* logits, const float* dlogits, float* dfeat,
             float* const* grads, int B, void* ws, hipStream_t s);
// wgrad.hip: weight gradient of a 3x3 / 1x1 convolution (training step)
int conv_wgrad_supported(int res, int cin, int cout, int ks);
size_t conv_wgrad_workspace_bytes(int B, int res, int cin, int cout, int ks);
int launch_conv_wgrad(int mode, const void* act, const void* dy, float* dw, int B, int res, int cin, int cout, int ks, int accumulate,
                      void* workspace, hipStream_t s, float scale = 1.0f);  // dw (+)= scale * sum
int launch_images_to_u8(const float* x, uint8_t* out, int64_t B, int C, int HW, hipStream_t s);
int launch_act_to_nchw(int dtype, const void* in, float* out, int B, int C, int HW, hipStream_t s);  // HW, C multiples of 32
int launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int HW, hipStream_t s);

// Single-head self-attention over T tokens, head dim 256 (AttentionOp + value product, EDM/network.py:160-168, 295-296).
// q,k: [B][T][256], vt: [B][256][T], out [B][T][256]; dtype 0: fp32 tensors, exact fp32 products / 1: bf16 / 2: fp32 tensors,
// split-bf16 products (FG_DTYPE_BF16X3).
int launch_attention(int dtype, const void* q, const void* k, const void* vt, void* out, int B, int T, hipStream_t s);

// MFMA output head (aux.hip): aux_conv(silu(aux_norm(x))) + EDM output preconditioning, res == 32.
size_t aux_pack_elems(int C);
int launch_pack_aux_weights(int dtype, const float* w, void* out, int C, int cout, hipStream_t s);
int aux_head_supported(int dtype, int res, int C, int cout);
int launch_aux_head(int dtype, const void* x, const float2* ab, const void* wpack, const float* bias, const float* x_t,
                    const float* coef, float* out, int B, int C, int cout, hipStream_t s, float* raw = nullptr);

// MFMA stem conv (aux.hip): conv3x3(c_in * x_t) + bias, img_channels -> 128, NCHW fp32 in, NHWC activations out.
int stem_supported(int res, int cin, int cout);
size_t stem_pack_elems();
int launch_pack_stem_weights(int dtype, const float* w, void* out, int cin, hipStream_t s);
int launch_stem(int dtype, const float* x, const float* c_in, const void* wpack, const float* bias, void* out, float2* stats,
                int B, int res, int cin, hipStream_t s);

// dit.hip: the non-GEMM pieces of a DiT forward (reference fastgen/networks/DiT/network.py); dtype = token-tensor storage (1 bf16, 0 fp32)
int launch_dit_ln_modulate(int dtype, int D, const void* x, const float* mod, int mod_stride, int shift_off, int scale_off, void* y,
                           int ntok, int tokens_per_image, hipStream_t s);
// fp32 tokens in, y as [hi | lo] bf16 planes [ntok][2 D] (the split-bf16 token GEMM's A operand)
int launch_dit_ln_modulate_split(int D, const float* x, const float* mod, int mod_stride, int shift_off, int scale_off, void* y, int ntok,
                                 int tokens_per_image, hipStream_t s);
int launch_dit_final(int dtype, int D, const void* x, const float* mod, const float* w, const float* bias, float* out, int ntok, int grid,
                     int p, int C, hipStream_t s);
int launch_dit_patch_embed(int dtype, const float* x, const float* w, const float* bias, const float* pos, void* out, int B, int C, int grid,
                           int p, int D, hipStream_t s);
int launch_dit_fourier(const float* t, float* f, int B, int dim, hipStream_t s);
int launch_dit_cond(const float* t_emb, const float* r_emb, const float* table, const int64_t* cls, float* c, float* sc, int B, int D,
                    int rows, int* err, hipStream_t s);
// mode 2 (bf16x3): q, k, vt are hi / lo bf16 planes, lo_off elements apart (conv.hip OUT_HEADS writes them so); else lo_off unused
int launch_dit_attention(int mode, const void* q, const void* k, const void* vt, void* out, int B, int heads, int head_dim, size_t lo_off,
                         hipStream_t s);

// gemm.hip: token GEMM of the transformer blocks in the bf16 compute mode.  out[tok][n] = epilogue(sum_k A[tok][k] W[n][k] + bias[n]),
// A [M][K] bf16, W [N][K] bf16 (launch_cvt_bf16 of the fp32 parameter), K % 64 == 0, N % 16 == 0.
// Token epilogue (heads == 0): v = acc + bias; act == 1: tanh-GELU; gate: v *= gate[((row0 + row) / gate_rows) * gate_stride + n];
// resid: v += resid[row][n]; out [M][N] bf16.  Head-split epilogue (heads > 0, N = 3 heads head_dim): q | k [B][heads][T][head_dim],
// v^T [B][heads][head_dim][T] with B index (row0 + row) / T.
struct GemmArgs {
    const void* A = nullptr;
    const void* W = nullptr;
    const float* bias = nullptr;
    void* out = nullptr;
    int M = 0, N = 0, K = 0;
    int act = 0;
    const float* gate = nullptr;
    int gate_stride = 0, gate_rows = 1;
    const void* resid = nullptr;
    void *q = nullptr, *k = nullptr, *vt = nullptr;
    int heads = 0, head_dim = 0, T = 0;
    int row0 = 0;  // global index of row 0 (set by the launcher when it cuts M into pieces below the 2 GiB buffer-offset range)
    int lda = 0, ldw = 0;         // row pitches in elements when they differ from K (the split-bf16 flavour; set by launch_gemm_x3)
    int a_wrap = 0;               // ... its A' = [hi | hi | lo] read out of [hi | lo]: K-steps >= a_wrap read a_wrap steps back
    size_t lo_off = 0;            // ... head-split epilogue: elements from a hi plane to its lo plane
    float* scratch = nullptr;     // optional fp32 scratch for split-K partial sums (short grids; see gemm.hip gm_pick_ksplit)
    size_t scratch_bytes = 0;
    int ksplit = 1;               // set by the launcher
    int out_f32 = 0;              // 1: out (and resid) fp32 [M][N] instead of bf16 (ping-pong kernel only: M, N >= 256)
    int variant = -1;  // -1: default (env FASTGEN_AMD_GEMM_PP, 1 unless set to 0); 0 register-staged kernel; 1 LDS-DMA ping-pong kernel; 2 its narrow-tile form
    int xn = 1;    // 0: linear tile order; 1: XCD-aware order, split chosen by the launcher; 2 / 4 / 8: that many XCD columns over N
};
bool gemm_bf16_supported(const GemmArgs& a);
int launch_gemm_bf16(const GemmArgs& a, hipStream_t s, bool prepare_only = false);
int launch_cvt_bf16(const float* in, void* out, size_t n, hipStream_t s);
// the same kernel in the split-bf16 (fp32-grade) mode: see gemm.hip.  A = [hi | lo] planes [M][2 K] bf16, W = launch_split3_weights
// of the fp32 weight ([N][3 K]), fp32 epilogue; out_mode 0: fp32 [M][N] (+ fp32 resid), 1: [hi | lo] planes [M][2 N]; heads > 0: head-split
// hi / lo planes (q, k, vt + lo_off).  M >= 256, N >= 256, K % 64 == 0.
int launch_gemm_x3(const GemmArgs& a, int out_mode, hipStream_t s);
int launch_split3_weights(const float* w, void* out, int N, int K, hipStream_t s);
int launch_split_planes(const float* x, void* out, int64_t M, int K, hipStream_t s);

// wan.hip: the non-GEMM pieces of the causal video DiT (reference fastgen/networks/Wan/network_causal.py); bf16 token tensors
int launch_wan_patch_embed(const float* x, const float* w, const float* bias, void* out, int B, int C, int Fr, int H, int W, int D, hipStream_t s);
int launch_wan_mod(const float* table, const float* tproj, float* mod, int rows, int J, int D, hipStream_t s);
int launch_wan_outmod(const float* table, const float* temb, float* mod, int rows, int D, hipStream_t s);
int launch_silu(const float* x, float* y, int64_t n, hipStream_t s);
int launch_cvt_rows_bf16(const float* x, void* y, int64_t n, hipStream_t s);
int launch_wan_rope_table(const float2* tab_t, const float2* tab_h, const float2* tab_w, float2* cs, int L, int fs, int gw, int start, int S,
                          int nt, int nh, int nw, hipStream_t s);
// RMSNorm over D (w nullable: plain copy) + optional interleaved-pair RoPE (cs [L][64] {cos, sin}) of rows [rows][ld_src], row
// (b = r / L, l = r % L) written to dst + b * dst_bs + (dst_row0 + l) * ld_dst
int launch_rms_rope(int D, const void* src, int ld_src, const float* w, float eps, const float2* cs, void* dst, int64_t dst_bs, int dst_row0,
                    int ld_dst, int rows, int L, hipStream_t s);
// softmax(q k^T / sqrt(128)) v per (batch, head), head dim 128: q [B][Lq] rows of ldq elements (head h at column 128 h), k / v
// [B][Lkv] rows of ldk, out [B][Lq] rows of ldo; *_bs = elements between batches
// scratch: fa128_scratch_bytes(B, heads, Lq) bytes for key-split partials (nullptr: never split)
size_t fa128_scratch_bytes(int B, int heads, int Lq);
// the same kernel for head dim hd = 128 | 72 (DiT-XL/2's 16 x 72)
int launch_fa(int hd, const void* q, int ldq, int64_t q_bs, const void* k, const void* v, int ldk, int64_t kv_bs, void* out, int ldo, int64_t o_bs,
              int B, int heads, int Lq, int Lkv, hipStream_t s, void* scratch = nullptr, int force_split = 0);
int launch_fa128(const void* q, int ldq, int64_t q_bs, const void* k, const void* v, int ldk, int64_t kv_bs, void* out, int ldo, int64_t o_bs,
                 int B, int heads, int Lq, int Lkv, hipStream_t s, void* scratch = nullptr);
int launch_wan_final(int D, const void* x, const float* mod, const float* w, const float* bias, float* out, int ntok, int Fr, int gh, int gw, int C,
                     float eps, hipStream_t s);
