/*
 * fastgen_amd — C ABI of the MI355X-native few-step diffusion sampling path (libfastgen_amd.so).
 *
 * Drop-in boundary for ONE hot path of the reference (paths relative to the reference repo root):
 *     FastGenModel.generator_fn / _student_sample_loop      fastgen/methods/model.py:315-420
 *       -> EDMPrecond.forward -> SongUNet.forward             fastgen/networks/EDM/network.py:881-974, 489-574
 *       -> EDMNoiseSchedule.{latents,forward_process,x0_to_eps}  fastgen/networks/noise_schedule.py:72-88,425-449,544-574
 *
 * Everything here is plain C: opaque handle, raw device pointers, sizes, an explicit HIP stream
 * (passed as void* = hipStream_t).  No torch types.  All functions return 0 on success and a
 * non-zero FG_E* code on failure; fg_last_error() returns a thread-local message.  A handle is
 * re-entrant per stream only in the sense that calls on ONE handle must be serialised by the
 * caller (one Python thread per process in the reference).
 *
 * Tensors at the boundary keep the reference's layout and dtype: images are fp32 NCHW [B,C,H,W],
 * timesteps are fp64 [B] (noise_schedule.py:41,50), class labels fp32 [B,label_dim], parameters
 * are fp32 in the reference's state-dict shapes (OIHW conv weights).  Internally activations are
 * NHWC and weights are re-packed into MFMA fragment order; that never leaks through this ABI.
 */
#ifndef FASTGEN_AMD_H
#define FASTGEN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Every entry point below is exported with default visibility; nothing else leaves the library (it is built with
 * -fvisibility=hidden). */
#define FG_API __attribute__((visibility("default")))

#define FG_OK 0
#define FG_EINVAL 1      /* bad argument / unsupported configuration */
#define FG_ENOTREADY 2   /* a parameter is unbound or weights were not packed */
#define FG_EHIP 3        /* a HIP runtime call failed (message holds hipGetErrorString) */
#define FG_ENOMEM 4      /* workspace too small */

#define FG_DTYPE_F32 0   /* exact fp32: v_mfma_f32_32x32x2_f32, fp32 activations */
#define FG_DTYPE_BF16 1  /* bf16 MFMA operands, fp32 accumulate, bf16 activation storage, fp32 norm statistics / softmax */
/* Split-bf16 convolutions on fp32 tensors: every conv operand x is split into hi = bf16(x), lo = bf16(x - hi) and a product is
 * a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on the bf16 matrix pipe with fp32 accumulate (about 2^-17 relative per product, against
 * 2^-11 for the TF32 arithmetic the reference enables on NVIDIA GPUs, fastgen/utils/scripts.py:43-45, which gfx950 does not have);
 * activations, statistics, attention, embedding MLP and preconditioning exactly as FG_DTYPE_F32.  Three bf16 MFMAs per product:
 * roofline 2.5 PFLOP/s / 3, against 157 TFLOP/s for FG_DTYPE_F32. */
#define FG_DTYPE_BF16X3 2

#define FG_SAMPLE_SDE 0  /* student_sample_type='sde'  (methods/model.py:358-359) */
#define FG_SAMPLE_ODE 1  /* student_sample_type='ode'  (methods/model.py:360-361) */

#define FG_LOOP_X0 0        /* FastGenModel._student_sample_loop: x0 prediction + re-noise (methods/model.py:315-372) */
#define FG_LOOP_MEANFLOW 1  /* MeanFlowModel._student_sample_loop: x -= dt * u(x,t,r) (consistency_model/mean_flow.py:336-381) */
#define FG_LOOP_EULER 2     /* DiT._sample_flow: Euler steps of the flow ODE, optional classifier-free guidance (DiT/network.py:605-651);
                               fg_dit_sampler_run only */

#define FG_SCHEDULE_EDM 0   /* EDMNoiseSchedule: alpha = 1, sigma = t, t in [0.002, 80] (noise_schedule.py:729-777) */
#define FG_SCHEDULE_RF 1    /* RFNoiseSchedule:  alpha = 1 - t, sigma = t, t in [0, 0.999] (noise_schedule.py:1306-1341) */

#define FG_DROP_PRECOND_INPUT 1   /* drop_precond 'input'  (EDM/network.py:929-934); 'both' = INPUT | OUTPUT */
#define FG_DROP_PRECOND_OUTPUT 2  /* drop_precond 'output' (EDM/network.py:959-960) */

#define FG_MAX_LEVELS 8

/* kwargs of EDMPrecond(model_type="SongUNet", embedding_type="positional", encoder_type=decoder_type="standard",
 * resample_filter=[1,1], dropout=0) — fastgen/configs/net.py:29-48 and EDM/network.py:347-367, 809-821. */
typedef struct fg_edm_config {
    int img_resolution;                  /* 32 */
    int img_channels;                    /* 3 */
    int label_dim;                       /* 10 (0 = unconditional) */
    int augment_dim;                     /* 9: map_augment exists in the state dict but is skipped when sampling */
    int model_channels;                  /* 128 */
    int num_levels;                      /* len(channel_mult) */
    int channel_mult[FG_MAX_LEVELS];     /* {2,2,2} */
    int channel_mult_emb;                /* 4 */
    int num_blocks;                      /* 4 */
    int num_attn_resolutions;
    int attn_resolutions[FG_MAX_LEVELS]; /* {16} */
    int channel_mult_noise;              /* 1 */
    double sigma_data;                   /* 0.5 */
    double sigma_shift;                  /* 0.0 (applied in eval mode only, EDM/network.py:956: see fg_edm_set_training) */
    int compute_dtype;                   /* FG_DTYPE_* */
    int r_timestep;                      /* 1: second (target-time) embedding, cond_channels doubled (EDM/network.py:376,401-408) */
    int drop_precond;                    /* bit mask of FG_DROP_PRECOND_* (0 = full EDM preconditioning) */
    int schedule;                        /* FG_SCHEDULE_*: the noise schedule the sampler loop re-noises with */
} fg_edm_config;

typedef struct fg_edm fg_edm; /* opaque */

/* Thread-local message of the last failing call on this thread ("" if none). */
FG_API const char* fg_last_error(void);
/* "fastgen_amd <version> gfx950" */
FG_API const char* fg_version(void);

/* ---- network object: replaces EDMPrecond.__init__ / state-dict ownership (EDM/network.py:808-839) ---- */
FG_API int fg_edm_create(const fg_edm_config* cfg, fg_edm** out);
FG_API void fg_edm_destroy(fg_edm* h);

/* State-dict view: the entries of the reference's EDMPrecond.state_dict() that carry weights (parameters
 * only; the constant resample_filter buffers are folded into the kernels).  Same names, same shapes. */
FG_API int fg_edm_num_params(const fg_edm* h);
FG_API int fg_edm_param_info(const fg_edm* h, int index, const char** name, int* ndim, int64_t shape[4]);

/* Bind a parameter to caller-owned DEVICE memory (fp32, reference layout).  The pointer is borrowed: it is
 * read again by every fg_edm_pack_weights() and must stay valid until then.  Used instead of a one-shot
 * load so FSDP2 / optimizer updates can re-bind and re-pack. */
FG_API int fg_edm_bind_param(fg_edm* h, const char* name, const float* device_ptr, int64_t numel);
/* Build the kernel-layout copies (MFMA fragment order, compute dtype) of all bound parameters. */
FG_API int fg_edm_pack_weights(fg_edm* h, void* stream);

/* Bytes of caller-provided scratch needed for a batch (activations, skip stack, norm statistics ...). */
FG_API size_t fg_edm_workspace_bytes(const fg_edm* h, int batch);

/* EDMPrecond.forward(x_t, t, condition=class_labels, r=r, fwd_pred_type=net_pred_type) in eval mode
 * (EDM/network.py:881-974).  x_t,out: [B,C,H,W] fp32; t: [B] fp64; r: [B] fp64, required iff cfg.r_timestep (else
 * NULL, :505-510); class_labels: [B,label_dim] fp32 or NULL (NULL = the reference's zeros([1,label_dim]) broadcast,
 * :919-925).  All device pointers.  x_t is not modified; out may not alias x_t.  emb_out (nullable):
 * [B, model_channels*channel_mult_emb] mapping-network output. */
FG_API int fg_edm_forward(fg_edm* h, const float* x_t, const double* t, const double* r, const float* class_labels, float* out,
                   float* emb_out, int batch, void* workspace, size_t workspace_bytes, void* stream);

/* The same forward with the encoder feature taps of SongUNet.forward (EDM/network.py:525-544, 562-567): tap i is the i-th
 * encoder block whose name contains "block3" (the last block of resolution level i for num_blocks = 4), the tensors the
 * DMD2 discriminator reads (methods/distribution_matching/dmd2.py:142).  features: HOST array of
 * fg_edm_num_feature_taps() device pointers, each NULL (not requested: the reference's feature_indices set) or an
 * [B, channels, res, res] fp32 NCHW buffer.  out == NULL is return_features_early: the decoder is not run. */
FG_API int fg_edm_num_feature_taps(const fg_edm* h);
FG_API int fg_edm_feature_info(const fg_edm* h, int index, const char** key, int* channels, int* resolution);
FG_API int fg_edm_forward_features(fg_edm* h, const float* x_t, const double* t, const double* r, const float* class_labels,
                            float* out, float* const* features, int batch, void* workspace, size_t workspace_bytes,
                            void* stream);

/* FastGenModel.generator_fn (methods/model.py:374-420) around one of the student sampling loops:
 *   x = noise * sigma(t_list[0]);
 *   FG_LOOP_X0 (model.py:315-372, x0-predicting network without r_timestep):
 *     for i: x0 = forward(x, t_i); if t_{i+1} > 0: x = alpha(t_{i+1}) x0 + sigma(t_{i+1}) eps_i; return x0
 *     ('ode': eps_i = x0_to_eps(x, x0, t_i)).
 *   FG_LOOP_MEANFLOW (mean_flow.py:336-381, flow-predicting r_timestep network):
 *     'sde': x -= t_i * forward(x, t_i, r=0); if t_{i+1} > 0: x = forward_process(x, eps_i, t_{i+1});
 *     'ode': x -= (t_i - t_{i+1}) * forward(x, t_i, r=t_{i+1});   return x.
 * t_list: HOST array of steps+1 doubles, t_list[steps] must be 0 (model.py:410).  sample_type FG_SAMPLE_*.
 * eps: device [steps-1][B,C,H,W] noise to inject in 'sde' mode, or NULL to draw it on device from
 * (seed, step) with Philox4x32-10.  use_graph != 0 replays a cached hipGraph of the whole loop. */
FG_API int fg_sampler_run(fg_edm* h, const float* noise, const float* class_labels, const double* t_list, int steps,
                   int sample_type, int loop_kind, const float* eps, uint64_t seed, float* out, int batch,
                   void* workspace, size_t workspace_bytes, int use_graph, void* stream);

/* EDMNoiseSchedule.get_t_list(sample_steps) (noise_schedule.py:940-973) -> steps+1 doubles on the host. */
FG_API int fg_edm_t_list(int sample_steps, double* out_host);
/* RFNoiseSchedule.get_t_list(sample_steps) = BaseNoiseSchedule.get_t_list (noise_schedule.py:259-272). */
FG_API int fg_rf_t_list(int sample_steps, double* out_host);

/* ---- measurement hook (bench.py): time every launch of the dominant kernel — the fused 3x3 conv at 32x32 output
 * without resampling — with a hipEvent pair on the stream it is launched on.  While active the sampler runs eagerly
 * (no graph replay).  profile_end() synchronises the events and returns launches, summed milliseconds and summed
 * algorithmic FLOPs (2 * B*H*W * Cout * 9*Cin per launch). */
FG_API int fg_edm_profile_begin(fg_edm* h);
FG_API int fg_edm_profile_end(fg_edm* h, int64_t* launches, double* total_ms, double* total_flops);

/* ---- single-op entry points (used by the parity tests; same kernels the network path launches) ---- */

/* One UNetBlock (EDM/network.py:274-299) by index into the encoder+decoder block list.  Inputs NHWC fp32:
 * x1 [B,Hin,Win,c1] (+ optional x2 [B,Hin,Win,c2] = the skip tensor of the decoder's channel concat),
 * emb [B,emb_channels]; out [B,H,W,cout] NHWC. */
FG_API int fg_edm_num_blocks(const fg_edm* h);
FG_API int fg_edm_block_info(const fg_edm* h, int index, const char** key, int* cin, int* cout, int* res_in, int* res_out,
                      int* has_attention);
FG_API int fg_edm_run_block(fg_edm* h, int index, const float* x1, int c1, const float* x2, int c2, const float* emb,
                     float* out, int batch, void* workspace, size_t workspace_bytes, void* stream);

/* GroupNorm statistics folded with the affine parameters: ab[b][c] = {a, b} with y = a*x + b
 * (GroupNorm.forward, EDM/network.py:141-149; groups = min(32, C/4)).  x NHWC fp32 [B,HW,C]. */
FG_API int fg_op_gn_coeffs(const float* x1, int c1, const float* x2, int c2, const float* gamma, const float* beta,
                    float eps, float* ab_out, int batch, int hw, void* stream);

/* Elementwise sampler steps in fp64 (noise_schedule.py:72-88, 425-449, 544-574); n = elements per sample. */
FG_API int fg_op_latents(const float* noise, double t_init, float* out, int64_t total, void* stream);
FG_API int fg_op_forward_process(const float* x0, const float* eps, double t, int schedule, float* out, int64_t total,
                          void* stream);
FG_API int fg_op_x0_to_eps(const float* xt, const float* x0, double t, int schedule, float* out, int64_t total, void* stream);
/* ---- training step, first kernel (forward-only sampling does not use it) --------------------------------
 * Weight gradient of Conv2d.forward (EDM/network.py:93-126) as autograd computes it in the DMD2 student / fake-score
 * updates: dw[co][ci][ky][kx] (+)= sum_{n,y,x} dy[n,y,x,co] * act[n,y+ky-ks/2,x+kx-ks/2,ci], zero padding.
 * act [B,res,res,cin] and dy [B,res,res,cout] are NHWC bf16 (the conv's input operand and its output gradient),
 * dw is fp32 OIHW [cout,cin,ks,ks]; accumulate != 0 adds to dw.  Deterministic (fixed-order split-K reduction through
 * the workspace).  res in {8,16,32}, cin % 32 == 0 (ks = 3) / cin % 128 == 0 (ks = 1), cout % 128 == 0, ks in {1,3}. */
FG_API size_t fg_op_conv_wgrad_workspace_bytes(int batch, int res, int cin, int cout, int ks);
FG_API int fg_op_conv_wgrad(const void* act, const void* dy, float* dw, int batch, int res, int cin, int cout, int ks,
                     int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* The same weight gradient in the split-bf16 compute mode (FG_DTYPE_BF16X3): act and dy are NHWC fp32, each split into hi / lo
 * bf16 planes on the way into LDS, dy_lo*a_hi + dy_hi*a_lo + dy_hi*a_hi with fp32 accumulation.  Same shapes, workspace size
 * (fg_op_conv_wgrad_workspace_bytes) and determinism. */
FG_API int fg_op_conv_wgrad_f32(const float* act, const float* dy, float* dw, int batch, int res, int cin, int cout, int ks,
                                int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of one UNetBlock (EDM/network.py:274-299) as autograd computes it (bf16 or bf16x3 compute mode; every block variant: attention,
 * 2x down / up resampling, channel-concat input).  Same tensor conventions as fg_edm_run_block (NHWC fp32 x1 / x2 / dout / dx1 / dx2,
 * emb and demb [B, emb_channels]).  The block's forward is recomputed first.  Parameter gradients are ACCUMULATED into the fp32
 * buffers bound with fg_edm_bind_grad (same names and shapes as the parameters; unbound = not computed); demb is accumulated as
 * well (zero it first); dx1 / dx2 are overwritten (either may be NULL). */
FG_API int fg_edm_bind_grad(fg_edm* h, const char* name, float* grad, int64_t numel);
FG_API size_t fg_edm_block_backward_workspace_bytes(const fg_edm* h, int index, int batch);
FG_API int fg_edm_run_block_backward(fg_edm* h, int index, const float* x1, int c1, const float* x2, int c2, const float* emb,
                              const float* dout, float* dx1, float* dx2, float* demb, int batch, void* workspace,
                              size_t workspace_bytes, void* stream);

/* Whole-network backward (bf16 or bf16x3 compute mode; the exact-fp32 mode has none) of EDMPrecond.forward, what autograd computes for the student / fake-score network in
 * fastgen/methods/distribution_matching/dmd2.py.  have_forward == 0: runs the kept forward itself first (-> out, as
 * fg_edm_forward_train); have_forward != 0: a fg_edm_forward_train of the same batch / workspace / inputs precedes this call and its
 * per-block stash is differentiated as it stands - nothing but the cheap conv operands silu(norm(x)) is recomputed.  dout [B,C,H,W]
 * fp32 is dL/d(out).  Gradients of all bound parameters (fg_edm_bind_grad) are ACCUMULATED.  The gradient of x_t and gradients
 * arriving at the feature taps are served by fg_edm_backward_ex below (this entry point = that one with dfeatures = dx_t = NULL). */
FG_API size_t fg_edm_backward_workspace_bytes(const fg_edm* h, int batch);
FG_API int fg_edm_backward(fg_edm* h, const float* x_t, const double* t, const double* r, const float* labels, const float* dout,
                    float* out, int have_forward, int batch, void* workspace, size_t workspace_bytes, void* stream);
/* The forward half on its own (-> out), leaving in `workspace` (sized by fg_edm_backward_workspace_bytes) what the backward
 * reads.  A following fg_edm_backward(..., have_forward = 1, same batch, same workspace, same x_t / labels) skips its own
 * forward (out may then be NULL); nothing else may write to that workspace in between. */
FG_API int fg_edm_forward_train(fg_edm* h, const float* x_t, const double* t, const double* r, const float* labels, float* out,
                         float* const* features, int batch, void* workspace, size_t workspace_bytes, void* stream);
/* fg_edm_backward with the gradient paths DMD2's GAN branch uses (dmd2.py:137-146: the frozen teacher's feature taps feed the
 * discriminator, whose loss is differentiated back to the teacher's INPUT): dfeatures[tap] (array as in fg_edm_forward_features,
 * entries nullable, NCHW fp32) joins the gradient of that encoder output; dout == NULL means the forward returned the taps early
 * and only the encoder is differentiated; dx_t (nullable, [B,C,H,W] fp32) receives dL/dx_t.  features / out of
 * fg_edm_forward_train follow fg_edm_forward_features (out == NULL: early return). */
FG_API int fg_edm_backward_ex(fg_edm* h, const float* x_t, const double* t, const double* r, const float* labels, const float* dout,
                       const float* const* dfeatures, float* out, float* dx_t, int have_forward, int batch, void* workspace,
                       size_t workspace_bytes, void* stream);

/* fg_edm_backward_ex in up to three calls, in this order, on the same arguments and workspace (the state in between lives in the
 * workspace; nothing else may write to it): FG_BWD_DECODER = [kept forward if have_forward == 0] + output head + decoder blocks,
 * FG_BWD_ENCODER = feature-tap gradients + encoder blocks + stem (+ dx_t), FG_BWD_EMBED = embedding MLP.  When a call returns, every
 * parameter gradient of its part is enqueued complete - a data-parallel wrapper can start reducing the decoder's gradients while
 * the encoder is still being differentiated (the module's autograd nodes are split this way, fastgen/utils/distributed/ddp.py:44-72).
 * parts is a mask: FG_BWD_DECODER | FG_BWD_ENCODER | FG_BWD_EMBED in one call is fg_edm_backward_ex. */
#define FG_BWD_DECODER 1
#define FG_BWD_ENCODER 2
#define FG_BWD_EMBED 4
FG_API int fg_edm_backward_part(fg_edm* h, const float* x_t, const double* t, const double* r, const float* labels, const float* dout,
                                const float* const* dfeatures, float* out, float* dx_t, int have_forward, int parts, int batch,
                                void* workspace, size_t workspace_bytes, void* stream);

/* One head of Discriminator_EDM (networks/discriminators.py:62-137) on a [B,256,res,res] NCHW fp32 feature tap, res in {8,16,32}:
 * strided 4x4 convs + GroupNorm + SiLU down to 1x1, then a 1x1 conv to one logit per image.  params: fg_disc_edm_num_params(res)
 * device pointers in the reference's module order ({conv.weight, conv.bias, gn.weight, gn.bias} per strided conv, then the 1x1
 * conv's weight and bias), fp32, reference shapes.  dlogits == NULL: forward only.  Otherwise also the backward of the same call:
 * dfeat (nullable, overwritten) and grads (nullable array / entries; same order and shapes as params; accumulated). */
FG_API int fg_disc_edm_num_params(int res);
FG_API size_t fg_disc_edm_workspace_bytes(int res, int batch);
FG_API int fg_disc_edm_run(const float* feat, int res, const float* const* params, float* logits, const float* dlogits, float* dfeat,
                    float* const* grads, int batch, void* workspace, size_t workspace_bytes, void* stream);

/* Forward-mode derivative (`torch.func.jvp(net, (x_t, t, r), tangents)` in MeanFlowModel._jvp / sCM,
 * consistency_model/mean_flow.py:240-250, sCM.py:179): out = EDMPrecond.forward(x_t, t, r), jvp = its directional derivative along
 * (vx [B,C,H,W], vt [B], vr [B]) (vt / vr nullable = 0; fp32).  bf16 or bf16x3 compute mode; workspace sized by
 * fg_edm_backward_workspace_bytes (the pass runs next to the kept forward and reuses its per-block stash). */
FG_API int fg_edm_jvp(fg_edm* h, const float* x_t, const double* t, const double* r, const float* labels, const float* vx, const float* vt,
               const float* vr, float* out, float* jvp, int batch, void* workspace, size_t workspace_bytes, void* stream);

/* The module's train()/eval() state as far as the arithmetic depends on it: `sigma_shift = None if self.training else
 * self.sigma_shift` (EDM/network.py:956).  training != 0: every following forward / backward / jvp / sampler call of this handle
 * evaluates precond_output without the shift; 0 (the state of a new handle): with cfg.sigma_shift.  Dropout is separate
 * (fg_edm_set_dropout): the reference's F.dropout also follows self.training, the module mirrors that when it sets both. */
FG_API int fg_edm_set_training(fg_edm* h, int training);

/* Augmentation labels of the training-time augmentation pipeline (EDM/network.py:495, 518-519, 903-915: condition =
 * {"aug_condition", "orig_condition"}): [B, augment_dim] fp32, added to the mapping network's input through map_augment by every
 * following forward / backward / jvp call of this handle until reset with NULL (borrowed pointer; must cover the call's batch). */
FG_API int fg_edm_set_augment(fg_edm* h, const float* augment_labels);

/* Training-mode dropout of conv1's operand in every UNetBlock (EDM/network.py:283-284; the SFT config trains with p = 0.13):
 * honoured by fg_edm_forward_train / fg_edm_backward(_ex) / fg_edm_jvp (never by the inference entry points), p = 0 turns it off.
 * The mask is Philox4x32-10(seed; element, block) - the backward regenerates it - so a forward and its backward must see the same
 * (p, seed); set p before sizing the workspace (one more tensor per block).  fg_op_dropout_mask writes the keep factors
 * (0 or 1/(1-p)) of one block's operand, flattened [B, H*W, C] (parity tests). */
FG_API int fg_edm_set_dropout(fg_edm* h, float p, uint64_t seed);
FG_API int fg_op_dropout_mask(float* out, int64_t total, float p, uint32_t block_index, uint64_t seed, void* stream);

/* ---- DiT: the class-conditional diffusion transformer of fastgen/networks/DiT/network.py -------------------------
 * kwargs of DiT(input_size, patch_size, in_channels, hidden_size, depth, num_heads, mlp_ratio, num_classes, class_dropout_prob,
 * r_timestep, ...) (:233-253; configs/net.py:98-127).  Supported: 256 tokens (input_size / patch_size == 16), hidden_size in
 * {384, 768, 1024, 1152} with head dim 64 or 72 (DiT-S/2, B/2, L/2, XL/2), mlp_hidden % 128 == 0. */
typedef struct fg_dit_config {
    int input_size;      /* 32 (latent resolution) */
    int patch_size;      /* 2 */
    int in_channels;     /* 4 */
    int hidden_size;     /* 1152 */
    int depth;           /* 28 */
    int num_heads;       /* 16 */
    int mlp_hidden;      /* int(hidden_size * mlp_ratio) = 4608 */
    int embedding_rows;  /* num_classes + (class_dropout_prob > 0): rows of y_embedder.class_embeddings (:116-118) */
    int r_timestep;      /* 1: second time embedding r_embedder (:276-279) */
    int compute_dtype;   /* FG_DTYPE_* */
} fg_dit_config;
typedef struct fg_dit fg_dit; /* opaque */

FG_API int fg_dit_create(const fg_dit_config* cfg, fg_dit** out);
FG_API void fg_dit_destroy(fg_dit* h);
/* State-dict view: the reference's names and shapes (restated timm attribute names for the patch embedding / attention / MLP
 * layers), `pos_embed` (a persistent buffer there) included.  bind + pack: unlike the fg_edm handle (it borrows its parameters' memory until
 * the next pack) this engine copies at pack time - see fg_dit_pack_group. */
FG_API int fg_dit_num_params(const fg_dit* h);
FG_API int fg_dit_param_info(const fg_dit* h, int index, const char** name, int* ndim, int64_t shape[4]);
FG_API int fg_dit_bind_param(fg_dit* h, const char* name, const float* device_ptr, int64_t numel);
FG_API int fg_dit_pack_weights(fg_dit* h, void* stream);
/* fg_dit_pack_weights for the parameters whose names start with `prefix` and not with `exclude` (nullable) - "blocks.7.", "x_embedder.",
 * ... : the unit a sharded-data-parallel wrapper gathers at a time (DiT.fully_shard, DiT/network.py:402-420).  Packing COPIES: the block
 * linears into their GEMM layouts, every other parameter into engine-owned fp32 storage, so a bound pointer is read during
 * fg_dit_pack_group / fg_dit_pack_weights only and may be freed once that call's work on `stream` has been ordered before the free.
 * fg_dit_forward runs once every parameter has been packed since its last bind. */
FG_API int fg_dit_pack_group(fg_dit* h, const char* prefix, const char* exclude, void* stream);
FG_API size_t fg_dit_workspace_bytes(const fg_dit* h, int batch);
/* The network part of DiT.forward (:511-547): patch embedding + positional table, conditioning vector, the transformer blocks,
 * output projection, unpatchify.  x_t, out: [B,C,H,W] fp32; t, r: [B] fp32 AS THE EMBEDDERS SEE THEM (after prepare_t's rescaling,
 * :457-462, the SiT flip :503-504 and, for time_cond_type 'diff', the t - r difference :520-521: scalar host-side plumbing);
 * r NULL = no r embedding; class_ids: [B] int64 row of the class table (num_classes = the unconditional row, :493-498); an id outside
 * [0, embedding_rows) makes that sample's output NaN and the NEXT call on the handle return FG_EINVAL (the reference's nn.Embedding
 * device-asserts; the check cannot surface in the asynchronous call itself);
 * cond_out (nullable): [B, hidden_size] conditioning vector c.  Prediction-type conversion and the SiT sign stay with the caller. */
FG_API int fg_dit_forward(fg_dit* h, const float* x_t, const float* t, const float* r, const int64_t* class_ids, float* out,
                          float* cond_out, int batch, void* workspace, size_t workspace_bytes, void* stream);

/* fg_dit_forward with the block-output taps of DiT.forward (`feature_indices` / `return_features_early`, :483-484, 536-543, 563-566):
 * feature_blocks: HOST array of num_features ascending block indices; features: HOST array of as many device buffers, each
 * [B, tokens, hidden_size] fp32 = the token tensor behind that block.  out == NULL: return after the last requested block. */
FG_API int fg_dit_forward_features(fg_dit* h, const float* x_t, const float* t, const float* r, const int64_t* class_ids, float* out,
                                   float* cond_out, const int* feature_blocks, float* const* features, int num_features, int batch,
                                   void* workspace, size_t workspace_bytes, void* stream);

/* The sampling loops around fg_dit_forward as ONE call (replayed as one hipGraph per (batch, steps, pointers) when use_graph != 0):
 *   FG_LOOP_X0        FastGenModel.generator_fn + _student_sample_loop (methods/model.py:315-420): x = noise * sigma(t_0); per step
 *                     x0 = convert(forward(x, t_i)) [flow -> x0: x - t v, fp64]; if t_{i+1} > 0: x = alpha(t_{i+1}) x0 + sigma(t_{i+1}) eps_i
 *                     ('ode': eps_i = x0_to_eps(x, x0, t_i)); returns the last x0.
 *   FG_LOOP_MEANFLOW  MeanFlowModel._student_sample_loop (consistency_model/mean_flow.py:336-381), as fg_sampler_run's.
 *   FG_LOOP_EULER     DiT._sample_flow (DiT/network.py:605-651): x += fp32(t_{i+1} - t_i) * v, with neg_class_ids != NULL
 *                     v = v_uncond + guidance_scale * (v_cond - v_uncond) from ONE forward of the doubled batch [x | x], [neg | cond].
 * The time conditioning that DiT.forward derives on the host (prepare_t's rescaling :457-462, the SiT flip :503-504 and sign :555-558,
 * the 'diff' form of r :520-521) is described by fg_dit_sampler_config and evaluated on the device from the resident t_list.
 * noise, out: [B,C,H,W] fp32; class_ids / neg_class_ids: [B] int64; t_list: HOST array of steps+1 doubles ending in 0; eps as for
 * fg_sampler_run.  Bit-identical to the same loop run step by step through fg_dit_forward and the fg_op_* elementwise kernels. */
typedef struct fg_dit_sampler_config {
    double t_scale;          /* the embedder sees fp32(t_scale * t): noise_scheduler.num_steps (1000) with scale_t on the RF schedule, else 1 */
    double guidance_scale;   /* FG_LOOP_EULER with neg_class_ids */
    int use_sit_convention;  /* t_e = 1 - t_e, flow output negated */
    int time_cond_diff;      /* time_cond_type == "diff": r_e = t_e - r_e */
    int net_pred_flow;       /* net_pred_type == "flow" (1) or "x0" (0) */
    int schedule;            /* FG_SCHEDULE_* of net.noise_scheduler */
} fg_dit_sampler_config;
FG_API size_t fg_dit_sampler_workspace_bytes(const fg_dit* h, int batch, int guided);
FG_API int fg_dit_sampler_run(fg_dit* h, const fg_dit_sampler_config* cfg, const float* noise, const int64_t* class_ids,
                              const int64_t* neg_class_ids, const double* t_list, int steps, int sample_type, int loop_kind, const float* eps,
                              uint64_t seed, float* out, int batch, void* workspace, size_t workspace_bytes, int use_graph, void* stream);

/* ---- Causal video DiT (SURVEY 8(f)3 and the CausVid row of 8(a); reference fastgen/networks/Wan/network_causal.py `CausalWan`, driven
 * chunk by chunk by `CausVidModel._student_sample_loop`, fastgen/methods/distribution_matching/causvid.py:87-185) --------------
 * The arithmetic of this network is diffusers' WanTransformer3DModel (un-vendored, diffusers==0.35.1): the HIP path follows the
 * restatement in oracle/wan_ref.py - PARITY UNPINNED.  bf16 token tensors and GEMM operands with fp32 accumulation,
 * fp32 norms / softmax / modulation: the reference runs this network in bf16 (configs/experiments/WanT2V/config_sf.py:19).
 * fg_wan_config = the fields of the transformer's config the path reads.  Supported: head_dim 128, inner dim (heads x 128) in
 * {256, 384, 1536 (1.3B), 2048, 5120 (14B)}, patch (1,2,2), ffn_dim / text_dim % 64 == 0, freq_dim 256, cross_attn_norm. */
typedef struct fg_wan_config {
    int num_heads;         /* 12 */
    int head_dim;          /* 128 */
    int in_channels;       /* 16 */
    int out_channels;      /* 16 */
    int text_dim;          /* 4096 */
    int freq_dim;          /* 256 */
    int ffn_dim;           /* 8960 */
    int num_layers;        /* 30 */
    int rope_max_seq_len;  /* 1024 */
    int chunk_size;        /* CausalFastGenNetwork.chunk_size, 3 (frames per autoregressive chunk; the sampler's business) */
    int total_num_frames;  /* 21: capacity of the self-attention KV caches, in frames */
    float eps;             /* 1e-6 */
} fg_wan_config;
typedef struct fg_wan fg_wan; /* opaque */

FG_API int fg_wan_create(const fg_wan_config* cfg, fg_wan** out);
FG_API void fg_wan_destroy(fg_wan* h);
/* State-dict view: `transformer.` + diffusers' parameter names (the reference's CausalWan.state_dict()); shapes up to 5-d (the
 * Conv3d patch embedding); bind / pack as for fg_edm. */
FG_API int fg_wan_num_params(const fg_wan* h);
FG_API int fg_wan_param_info(const fg_wan* h, int index, const char** name, int* ndim, int64_t shape[5]);
FG_API int fg_wan_bind_param(fg_wan* h, const char* name, const float* device_ptr, int64_t numel);
FG_API int fg_wan_pack_weights(fg_wan* h, void* stream);
/* As fg_dit_pack_group: "transformer.blocks.7." packs one block, prefix "transformer." with exclude "transformer.blocks." the rest
 * (the grouping of Wan.fully_shard, Wan/network.py:761-782).  Bound pointers are read at pack time only. */
FG_API int fg_wan_pack_group(fg_wan* h, const char* prefix, const char* exclude, void* stream);
/* Workspace for one forward over a chunk of `frames` latent frames of height x width (also enough for fg_wan_set_text). */
FG_API size_t fg_wan_workspace_bytes(const fg_wan* h, int batch, int frames, int height, int width);
/* `CausalWan.clear_caches` (:1030-1054): zero the self-attention caches, forget the text (cross-attention) caches. */
FG_API int fg_wan_clear_caches(fg_wan* h, void* stream);
/* The text condition: text [B, text_len, text_dim] fp32 (the text encoder's output).  Runs condition_embedder.text_embedder and
 * every block's attn2 k / v projections (+ norm_k) once: the reference's static cross-attention cache (:331-360). */
FG_API int fg_wan_set_text(fg_wan* h, const float* text, int batch, int text_len, void* workspace, size_t workspace_bytes, void* stream);
/* One network call of the autoregressive sampler: `CausalWan.forward(x_t, t, cache_tag="pos", cur_start_frame, store_kv, is_ar=True)`
 * up to the raw network output (:1077-1160; the flow -> x0 conversion stays with the caller).  x_t, out: [B, C, frames, H, W] fp32;
 * t_frames: [B * frames] fp32 AS THE EMBEDDER SEES THEM (rescale_t: 1000 t, `_compute_timestep_inputs` :1063-1075).  The chunk's
 * keys / values are written to cache rows [cur_start_frame * frame_seqlen, +frames * frame_seqlen) in every call and attention runs
 * over rows [0, that end): with store_kv = 0 the rows are rewritten by the chunk's later store_kv = 1 call, as in the reference's
 * loop (causvid.py:150-172), where only that call moves the cache length.  Restriction: a store_kv = 0 call whose rows were filled by
 * an earlier store_kv = 1 call (re-evaluating a stored chunk; the reference leaves its cache untouched there) returns FG_EINVAL. */
FG_API int fg_wan_forward(fg_wan* h, const float* x_t, const float* t_frames, float* out, int batch, int frames, int height, int width,
                          int cur_start_frame, int store_kv, void* workspace, size_t workspace_bytes, void* stream);
/* The teacher- / diffusion-forcing call over ALL frames: `CausalWan.forward(x_t, t, is_ar=False)` with frames == total_num_frames, where the
 * reference attends through the block-wise causal mask of `_prepare_blockwise_causal_attn_mask` (network_causal.py:131-196, 673-680):
 * a query sees the keys up to the end of its own chunk of chunk_size frames (the first chunk also holds frames % chunk_size).
 * t_frames [B * frames] may differ per frame (diffusion forcing).  RoPE starts at frame 0; the self-attention caches are neither
 * read nor written.  fg_wan_workspace_bytes(batch, total_num_frames, ...) covers this call's extra K / V buffers. */
FG_API int fg_wan_forward_block_causal(fg_wan* h, const float* x_t, const float* t_frames, float* out, int batch, int frames, int height,
                                       int width, void* workspace, size_t workspace_bytes, void* stream);

/* The chunk-by-chunk student loop of the causal video DiT as ONE call: `CausVidModel._student_sample_loop` (causvid.py:87-185), the
 * segment loop of `generator_fn_extrapolation` (:283-345, prefill_frames > 0) and the no-grad form of
 * `SelfForcingModel.rollout_with_gradient` (self_forcing.py:92-241, exit_steps != NULL).  x: [B, C, frames, H, W] fp32 latents (already
 * scaled by sigma(t_0) where the caller's generator_fn does that), overwritten in place with the generated frames.  Chunks: chunk_size
 * frames each, the remainder of frames joins the FIRST chunk.  Per chunk: for i in 0 .. last (last = steps - 1, or exit_steps[chunk]):
 * x0 = convert(fg_wan_forward(x_chunk, t_i, cur_start_frame, store_kv = 0)); unless i == last or t_{i+1} == 0 re-noise to t_{i+1} ('sde':
 * fresh noise, 'ode': the noise implied by (x_chunk, x0)); then the cache-fill call fg_wan_forward(x0 [re-noised to context_noise],
 * t = 0 [context_noise], store_kv = 1).  Chunks inside [0, prefill_frames) only run the cache-fill call at t = 0 on x as given.
 * The self-attention caches are emptied first and cleared (zeroed, text forgotten) at the end, as the reference's loop does: call
 * fg_wan_set_text before every run.  eps (nullable): steps - 1 (+ 1 if context_noise > 0) noise videos [B, C, frames, H, W] to inject,
 * indexed by re-noising step (last: the cache call's); NULL: Philox4x32-10 from (seed; chunk, step).  use_graph != 0: one hipGraph per
 * chunk (start frame, frame count and key length are baked in), replayed while shapes and pointers stay the same.  Bit-identical to
 * the same sequence of fg_wan_forward + fg_op_* calls. */
typedef struct fg_wan_sampler_config {
    double t_scale;        /* the embedder sees fp32(t_scale * t): noise_scheduler.num_steps (1000) on the RF schedule */
    double context_noise;  /* 0: the cache-fill call sees the clean chunk at t = 0 */
    int net_pred_flow;     /* net_pred_type == "flow" (1) or "x0" (0) */
    int schedule;          /* FG_SCHEDULE_* */
    int prefill_frames;    /* frames at the head of x that only fill the caches (multiple of chunk_size; frames then too) */
} fg_wan_sampler_config;
FG_API size_t fg_wan_sampler_workspace_bytes(const fg_wan* h, int batch, int frames, int height, int width);
FG_API int fg_wan_sampler_run(fg_wan* h, const fg_wan_sampler_config* cfg, float* x, const double* t_list, int steps, int sample_type,
                              const int* exit_steps, const float* eps, uint64_t seed, int batch, int frames, int height, int width,
                              void* workspace, size_t workspace_bytes, int use_graph, void* stream);

/* Samples to image bytes, the step after generator_fn in the reference's sample writer
 * (scripts/fid/compute_fid_from_ckpts.py:199): out[n,y,x,c] = uint8(clip(images[n,c,y,x] * 127.5 + 128, 0, 255)),
 * fp32 multiply then add, truncation; NaN -> 0.  images NCHW fp32, out NHWC bytes. */
FG_API int fg_op_images_to_u8(const float* images, uint8_t* out, int64_t batch, int channels, int height, int width,
                       void* stream);
/* Standard normal draws: Philox4x32-10(key = seed, counter = (offset, index/4)) + Box-Muller. */
FG_API int fg_op_randn(float* out, int64_t total, uint64_t seed, uint64_t offset, void* stream);
/* softmax(q k^T / sqrt(head_dim)) v per (batch, head), head_dim 128 or 72, bf16 tensors q / out [batch, lq, heads * head_dim], k / v
 * [batch, lkv, heads * head_dim], fp32 online softmax (wan.hip fa_kernel: the causal video DiT's KV-cache and text attention,
 * fastgen/networks/Wan/network_causal.py:331-412, and DiT-XL/2's 16 x 72 attention, fastgen/networks/DiT/network.py:168,191).
 * Exposed for the parity tests. */
FG_API int fg_op_attention(const void* q, const void* k, const void* v, void* out, int batch, int heads, int head_dim, int lq, int lkv,
                           void* stream);
/* The same with the number of key splits fixed: nsplit 0 = the launcher's cost model (what the networks run), 1 = one workgroup walks
 * all keys of its query tile, 2 .. 8 = that many workgroups share them and a merge pass combines their partial outputs by
 * log-sum-exp.  For the parity tests: the split paths against each other at the video DiT's full sequence lengths. */
FG_API int fg_op_attention_split(const void* q, const void* k, const void* v, void* out, int batch, int heads, int head_dim, int lq, int lkv,
                                 int nsplit, void* stream);
/* The transformer blocks' token GEMM in the bf16 compute mode (gemm.hip; reference: the nn.Linear calls of DiTBlock,
 * fastgen/networks/DiT/network.py:168-198, under bf16 autocast): out[m][n] = resid[m][n] + gate[(m / gate_rows) * gate_stride + n] *
 * act(sum_k a[m][k] w[n][k] + bias[n]) with a [m][k], w [n][k], resid / out [m][n] in bf16, fp32 accumulation, bias / gate fp32;
 * act 0 none, 1 GELU(tanh) (anything else: FG_EINVAL); bias, gate, resid nullable.  k % 64 == 0, n % 16 == 0.  tile_order: 0 linear, 1 XCD-aware (launcher's
 * choice), 2 / 4 / 8 XCD columns over n; + 16 forces the register-staged kernel, + 32 the LDS-DMA ping-pong kernel (default: the
 * latter where m, n >= 256), + 256 its narrow-tile form (256 x 128: what short token counts - one sample of the video DiT - take), + 512 the experimental
 * one-wave-per-SIMD kernel (token epilogue, m, n >= 256; no engine uses it), + 64 lets grids of fewer tiles than
 * half the CUs split K (fp32 partial sums + a finishing pass).  Exposed for the parity tests and scripts/gemm_bench.py. */
FG_API int fg_op_gemm_bf16(const void* a, const void* w, const float* bias, void* out, int m, int n, int k, int act, const float* gate,
                           int gate_stride, int gate_rows, const void* resid, int tile_order, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FASTGEN_AMD_H */
