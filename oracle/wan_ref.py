"""CPU restatement (torch, fp32) of the reference's causal video DiT path - TEST INFRASTRUCTURE ONLY (imported by tests/, never by
the product path).  PARITY UNPINNED: the arithmetic of this network lives in the un-vendored `diffusers==0.35.1`
(`WanTransformer3DModel`, fastgen requirements.txt:6; imported at fastgen/networks/Wan/network_causal.py:31-38), which is absent from
/root/reference and from this image, and the reference's own tests hold shapes only for it.  What is restated:

  * from the reference's own files (cited per function): the causal overrides of fastgen/networks/Wan/network_causal.py - RoPE with
    a frame offset (:79-128), the KV-cache append attention of `CausalWanAttnProcessor` (:199-464, autoregressive branch :377-412),
    the per-frame adaLN block (`_wan_block_forward_inline_cache`, :467-550), the per-frame output modulation of `classify_forward`
    (fastgen/networks/Wan/network.py:226-247), `CausalWan.forward` (:1077-1190 of network_causal.py: per-frame timesteps, rescale_t,
    flow -> x0 conversion), and `CausVidModel._student_sample_loop` (fastgen/methods/distribution_matching/causvid.py:87-185);
  * from diffusers 0.35.1's published algorithm (module names = its state-dict keys): `WanTransformer3DModel` - Conv3d patch
    embedding (1,2,2); `WanTimeTextImageEmbedding` (`Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0)` -> Linear - SiLU -
    Linear; `time_proj(silu(temb))` -> 6 D; `PixArtAlphaTextProjection` Linear - GELU(tanh) - Linear); `WanRotaryPosEmbed` (head dim
    split t/h/w = hd - 2 (hd // 3) | hd // 3 | hd // 3, `get_1d_rotary_pos_embed(theta=10000, repeat_interleave_real=True,
    freqs_dtype=float64)`); `WanTransformerBlock` (FP32LayerNorm eps 1e-6 without affine for norm1 / norm3, with affine for norm2
    (cross_attn_norm), `WanAttention` with bias-carrying q / k / v / out linears and RMSNorm(eps 1e-6, affine) over the FULL inner
    dimension on q and k ("rms_norm_across_heads"), `FeedForward(gelu-approximate)`, `scale_shift_table` [1, 6, D]); output
    `scale_shift_table` [1, 2, D], FP32LayerNorm, `proj_out`, un-patchify.

All tensors fp32 here (the reference runs the network in bf16: WanT2V/config_sf.py:19); the GPU tests hold the HIP path (bf16
operands, fp32 accumulation / statistics) to a bf16 tolerance against this."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


@dataclass
class WanConfig:
    num_heads: int = 12
    head_dim: int = 128
    in_channels: int = 16
    out_channels: int = 16
    text_dim: int = 4096
    freq_dim: int = 256
    ffn_dim: int = 8960
    num_layers: int = 30
    eps: float = 1e-6
    rope_max_seq_len: int = 1024
    chunk_size: int = 3
    total_num_frames: int = 21

    @property
    def dim(self) -> int:
        return self.num_heads * self.head_dim


WAN_1_3B = WanConfig()
TINY = WanConfig(num_heads=2, head_dim=128, text_dim=128, ffn_dim=512, num_layers=2, chunk_size=2, total_num_frames=6)


def state_dict_shapes(cfg: WanConfig) -> Dict[str, tuple]:
    """Keys / shapes of `CausalWan.state_dict()` (= `transformer.` + diffusers' WanTransformer3DModel keys, module order of its
    constructor; `logvar_linear` is the reference's addition, Wan/network.py:607-609)."""
    D, Fd = cfg.dim, cfg.ffn_dim
    sd = {"scale_shift_table": (1, 2, D), "patch_embedding.weight": (D, cfg.in_channels, 1, 2, 2), "patch_embedding.bias": (D,)}
    ce = "condition_embedder."
    sd.update({ce + "time_embedder.linear_1.weight": (D, cfg.freq_dim), ce + "time_embedder.linear_1.bias": (D,),
               ce + "time_embedder.linear_2.weight": (D, D), ce + "time_embedder.linear_2.bias": (D,),
               ce + "time_proj.weight": (6 * D, D), ce + "time_proj.bias": (6 * D,),
               ce + "text_embedder.linear_1.weight": (D, cfg.text_dim), ce + "text_embedder.linear_1.bias": (D,),
               ce + "text_embedder.linear_2.weight": (D, D), ce + "text_embedder.linear_2.bias": (D,)})
    for i in range(cfg.num_layers):
        b = f"blocks.{i}."
        sd[b + "scale_shift_table"] = (1, 6, D)
        for att in ("attn1", "attn2"):
            for lin in ("to_q", "to_k", "to_v", "to_out.0"):
                sd[b + f"{att}.{lin}.weight"] = (D, D)
                sd[b + f"{att}.{lin}.bias"] = (D,)
            sd[b + f"{att}.norm_q.weight"] = (D,)
            sd[b + f"{att}.norm_k.weight"] = (D,)
        sd[b + "norm2.weight"] = (D,)
        sd[b + "norm2.bias"] = (D,)
        sd[b + "ffn.net.0.proj.weight"] = (Fd, D)
        sd[b + "ffn.net.0.proj.bias"] = (Fd,)
        sd[b + "ffn.net.2.weight"] = (D, Fd)
        sd[b + "ffn.net.2.bias"] = (D,)
    sd["proj_out.weight"] = (cfg.out_channels * 4, D)
    sd["proj_out.bias"] = (cfg.out_channels * 4,)
    sd["logvar_linear.weight"] = (1, D)
    sd["logvar_linear.bias"] = (1,)
    return {"transformer." + k: v for k, v in sd.items()}


def random_state_dict(cfg: WanConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded O(1)-signal weights (fan-in scaled matrices, small biases, norm weights around 1) - every branch contributes."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in state_dict_shapes(cfg).items():
        if k.endswith("norm_q.weight") or k.endswith("norm_k.weight") or k.endswith("norm2.weight"):
            v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("scale_shift_table"):
            v = torch.randn(shp, generator=g) / shp[-1] ** 0.5
        elif k.endswith(".bias"):
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = int(torch.tensor(shp[1:]).prod())
            v = torch.randn(shp, generator=g) * fan_in ** -0.5
        sd[k] = v
    return sd


def timesteps_proj(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`: [cos | sin] of t * 10000^(-j / half)."""
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t.float()[:, None] * freq[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def rope_tables(cfg: WanConfig):
    """`WanRotaryPosEmbed.__init__`: per-axis cos / sin [max_seq_len, axis_dim] with every frequency repeated twice, float64 angles."""
    hd = cfg.head_dim
    h_dim = w_dim = 2 * (hd // 6)
    t_dim = hd - h_dim - w_dim
    cos, sin = [], []
    for d in (t_dim, h_dim, w_dim):
        freqs = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float64)[: d // 2] / d))
        ang = torch.outer(torch.arange(cfg.rope_max_seq_len, dtype=torch.float64), freqs)
        cos.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin.append(ang.sin().repeat_interleave(2, dim=1).float())
    return cos, sin, (t_dim, h_dim, w_dim)


def rope_for_chunk(cfg: WanConfig, frames: int, gh: int, gw: int, start_frame: int):
    """`_rope_forward_with_time_offset` (network_causal.py:79-128): [frames * gh * gw, head_dim] cos and sin with the temporal rows taken
    from `start_frame` on (clamped to the table's last row)."""
    cos, sin, _ = rope_tables(cfg)
    total = cos[0].shape[0]
    idx = torch.clamp(torch.arange(start_frame, start_frame + frames), max=total - 1)
    cf, sf = cos[0][idx], sin[0][idx]
    c = torch.cat([cf[:, None, None, :].expand(frames, gh, gw, -1), cos[1][:gh][None, :, None, :].expand(frames, gh, gw, -1),
                   cos[2][:gw][None, None, :, :].expand(frames, gh, gw, -1)], dim=-1)
    s = torch.cat([sf[:, None, None, :].expand(frames, gh, gw, -1), sin[1][:gh][None, :, None, :].expand(frames, gh, gw, -1),
                   sin[2][:gw][None, None, :, :].expand(frames, gh, gw, -1)], dim=-1)
    return c.reshape(frames * gh * gw, -1), s.reshape(frames * gh * gw, -1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """`apply_rotary_emb` of the processor (network_causal.py:274-289): x [B, L, H, hd], interleaved pairs."""
    x1, x2 = x[..., 0::2], x[..., 1::2]
    c, s = cos[None, :, None, 0::2], sin[None, :, None, 1::2]
    out = torch.empty_like(x)
    out[..., 0::2] = x1 * c - x2 * s
    out[..., 1::2] = x1 * s + x2 * c
    return out


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def layer_norm(x: torch.Tensor, eps: float, w=None, b=None) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`dispatch_attention_fn(q, k, v)` on [B, L, H, hd] tensors: softmax(q k^T / sqrt(hd)) v, flattened over heads.  mask [Lq, Lk]
    bool (True = attend): the flex_attention call under a block mask (network_causal.py:437)."""
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=mask)
    return o.transpose(1, 2).flatten(2, 3)


def blockwise_causal_mask(num_frames: int, frame_seqlen: int, chunk_size: int) -> torch.Tensor:
    """`_prepare_blockwise_causal_attn_mask` (network_causal.py:131-196) as a dense [L, L] bool matrix: (kv < end of q's chunk) | (kv == q);
    chunks of chunk_size frames, the first one front-loaded with num_frames % chunk_size.  (The reference's zero padding to a multiple
    of 128 tokens adds rows / columns no real query sees and is dropped from its output, :418-443.)"""
    nch, rem = num_frames // chunk_size, num_frames % chunk_size
    counts = [rem] if nch == 0 else [chunk_size + rem] + [chunk_size] * (nch - 1)
    ends = torch.zeros(num_frames * frame_seqlen, dtype=torch.long)
    start = 0
    for nf in counts:
        n = nf * frame_seqlen
        ends[start:start + n] = start + n
        start += n
    idx = torch.arange(num_frames * frame_seqlen)
    return (idx[None, :] < ends[:, None]) | (idx[None, :] == idx[:, None])


class CausalWanRef:
    """The network with its external KV caches (network_causal.py:708-812): per block a self-attention cache of
    `total_num_frames * frame_seqlen` tokens and a static cross-attention cache."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg: WanConfig):
        self.cfg = cfg
        self.p = {k[len("transformer."):]: v.float() for k, v in sd.items()}
        self.clear_caches()

    def clear_caches(self):
        self.self_kv: List[Optional[dict]] = [None] * self.cfg.num_layers
        self.cross_kv: List[Optional[dict]] = [None] * self.cfg.num_layers

    def _lin(self, x, name):
        return F.linear(x, self.p[name + ".weight"], self.p[name + ".bias"])

    def forward(self, x_t: torch.Tensor, t: torch.Tensor, text: torch.Tensor, cur_start_frame: int = 0, store_kv: bool = False,
                trace: Optional[dict] = None, block_causal: bool = False) -> torch.Tensor:
        """The raw network output (net_pred_type: flow) of `CausalWan.forward(..., is_ar=True)`.  x_t [B, C, F, H, W]; t [B] in the
        schedule's units (rescaled by 1000 here, `_compute_timestep_inputs` :1063-1075) - every frame of the chunk gets t - or [B, F]
        (one per frame: diffusion forcing).
        block_causal: the `is_ar=False` call over all total_num_frames frames - self-attention under the block-wise causal mask, no
        self-attention cache read or written (the reference allocates none for a full-length call, :681-690)."""
        cfg, p = self.cfg, self.p
        B, C, Fr, H, W = x_t.shape
        gh, gw, D = H // 2, W // 2, cfg.dim
        fs = gh * gw  # frame_seqlen
        L = Fr * fs
        ts = (1000.0 * t.float()).view(B, -1).expand(B, Fr).reshape(-1)  # [B * F]
        if block_causal:
            assert Fr == cfg.total_num_frames and cur_start_frame == 0 and not store_kv
            mask = blockwise_causal_mask(Fr, fs, cfg.chunk_size)
        cos, sin = rope_for_chunk(cfg, Fr, gh, gw, cur_start_frame)
        # patch embedding: Conv3d kernel = stride = (1, 2, 2); tokens ordered (f, h, w)
        hs = F.conv3d(x_t.float(), p["patch_embedding.weight"], p["patch_embedding.bias"], stride=(1, 2, 2)).flatten(2).transpose(1, 2)
        # condition embedder (per frame)
        temb = self._lin(F.silu(self._lin(timesteps_proj(ts, cfg.freq_dim), "condition_embedder.time_embedder.linear_1")),
                         "condition_embedder.time_embedder.linear_2")  # [B * F, D]
        tproj = self._lin(F.silu(temb), "condition_embedder.time_proj").view(B, Fr, 6, D)
        ctx = self._lin(F.gelu(self._lin(text.float(), "condition_embedder.text_embedder.linear_1"), approximate="tanh"),
                        "condition_embedder.text_embedder.linear_2")  # [B, Lt, D]
        if trace is not None:
            trace["tokens"], trace["temb"], trace["ctx"] = hs.clone(), temb.clone(), ctx.clone()
        cache_start = cur_start_frame * fs
        cap = cfg.total_num_frames * fs
        for i in range(cfg.num_layers):
            b = f"blocks.{i}."
            mod = p[b + "scale_shift_table"].view(1, 1, 6, D) + tproj  # [B, F, 6, D]
            shift, scale, gate, c_shift, c_scale, c_gate = [mod[:, :, j] for j in range(6)]

            def per_frame(x, sc, sh):
                return (x.view(B, Fr, fs, D) * (1 + sc[:, :, None]) + sh[:, :, None]).reshape(B, L, D)

            # 1. self-attention over the cached frames and this chunk (:377-412)
            y = per_frame(layer_norm(hs, cfg.eps), scale, shift)
            q = rms_norm(self._lin(y, b + "attn1.to_q"), p[b + "attn1.norm_q.weight"], cfg.eps).view(B, L, cfg.num_heads, -1)
            k = rms_norm(self._lin(y, b + "attn1.to_k"), p[b + "attn1.norm_k.weight"], cfg.eps).view(B, L, cfg.num_heads, -1)
            v = self._lin(y, b + "attn1.to_v").view(B, L, cfg.num_heads, -1)
            q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
            if block_causal:
                att = self._lin(sdpa(q, k, v, mask), b + "attn1.to_out.0")
            else:
                if self.self_kv[i] is None:
                    self.self_kv[i] = {"k": torch.zeros(B, cap, cfg.num_heads, cfg.head_dim), "v": torch.zeros(B, cap, cfg.num_heads, cfg.head_dim)}
                kv = self.self_kv[i]
                if store_kv:
                    kv["k"][:, cache_start:cache_start + L] = k
                    kv["v"][:, cache_start:cache_start + L] = v
                k_full = torch.cat([kv["k"][:, :cache_start], k], dim=1)
                v_full = torch.cat([kv["v"][:, :cache_start], v], dim=1)
                att = self._lin(sdpa(q, k_full, v_full), b + "attn1.to_out.0")
            hs = hs + (att.view(B, Fr, fs, D) * gate[:, :, None]).reshape(B, L, D)
            # 2. cross-attention to the text (static cache, :331-360)
            y = layer_norm(hs, cfg.eps, p[b + "norm2.weight"], p[b + "norm2.bias"])
            q2 = rms_norm(self._lin(y, b + "attn2.to_q"), p[b + "attn2.norm_q.weight"], cfg.eps).view(B, L, cfg.num_heads, -1)
            if self.cross_kv[i] is None or store_kv:
                k2 = rms_norm(self._lin(ctx, b + "attn2.to_k"), p[b + "attn2.norm_k.weight"], cfg.eps).view(B, -1, cfg.num_heads, cfg.head_dim)
                v2 = self._lin(ctx, b + "attn2.to_v").view(B, -1, cfg.num_heads, cfg.head_dim)
                if store_kv:
                    self.cross_kv[i] = {"k": k2, "v": v2}
            else:
                k2, v2 = self.cross_kv[i]["k"], self.cross_kv[i]["v"]
            hs = hs + self._lin(sdpa(q2, k2, v2), b + "attn2.to_out.0")
            # 3. feed-forward
            y = per_frame(layer_norm(hs, cfg.eps), c_scale, c_shift)
            ff = self._lin(F.gelu(self._lin(y, b + "ffn.net.0.proj"), approximate="tanh"), b + "ffn.net.2")
            hs = hs + (ff.view(B, Fr, fs, D) * c_gate[:, :, None]).reshape(B, L, D)
            if trace is not None:
                trace[f"block{i}"] = hs.clone()
        # output: per-frame modulation (Wan/network.py:226-247), projection, un-patchify
        so = p["scale_shift_table"].view(1, 1, 2, D) + temb.view(B, Fr, 1, D)
        y = (layer_norm(hs, cfg.eps).view(B, Fr, fs, D) * (1 + so[:, :, 1, None]) + so[:, :, 0, None]).reshape(B, L, D)
        o = self._lin(y, "proj_out").view(B, Fr, gh, gw, 1, 2, 2, -1)
        return o.permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(B, -1, Fr, H, W)


def student_sample_loop(net: CausalWanRef, x: torch.Tensor, t_list: torch.Tensor, text: torch.Tensor, eps_list: Optional[list] = None,
                        sample_type: str = "sde") -> torch.Tensor:
    """`CausVidModel._student_sample_loop` (causvid.py:87-185) on the RF schedule (alpha = 1 - t, sigma = t), context_noise = 0:
    per chunk N x {x0 = x_t - t * flow(x_t, t); re-noise to t_next}, then one forward at t_list[-1] that fills the KV cache.
    eps_list: the injected noise draws of the 'sde' branch, in call order."""
    net.clear_caches()
    x = x.clone()
    B, Fr = x.shape[0], x.shape[2]
    cs = net.cfg.chunk_size
    nchunks, rem = Fr // cs, Fr % cs
    draws = iter(eps_list or [])
    for i in range(max(1, nchunks)):
        if nchunks == 0:
            start, end = 0, rem
        else:
            start = 0 if i == 0 else cs * i + rem
            end = cs * (i + 1) + rem
        x_next = x[:, :, start:end]
        for step in range(len(t_list) - 1):
            t_cur = t_list[step].expand(B)
            x_cur = x_next
            flow = net.forward(x_cur, t_cur, text, cur_start_frame=start, store_kv=False)
            x_next = x_cur - float(t_list[step]) * flow  # convert_model_output flow -> x0 (noise_schedule.py:666-726, RF)
            t_next = float(t_list[step + 1])
            if t_next > 0:
                if sample_type == "sde":
                    eps = next(draws)
                else:
                    eps = (x_cur - (1 - float(t_list[step])) * x_next) / float(t_list[step])  # x0_to_eps
                x_next = (1 - t_next) * x_next + t_next * eps
        x[:, :, start:end] = x_next
        net.forward(x_next, t_list[-1].expand(B), text, cur_start_frame=start, store_kv=True)
    net.clear_caches()
    return x


def self_forcing_rollout(net: CausalWanRef, noise: torch.Tensor, t_list: torch.Tensor, text: torch.Tensor, end_steps: list,
                         same_step_across_blocks: bool = True, eps_list: Optional[list] = None, sample_type: str = "sde") -> torch.Tensor:
    """`SelfForcingModel.rollout_with_gradient` (self_forcing.py:92-241) without autograd, context_noise = 0, RF schedule: per chunk,
    denoise down to the chunk's exit step (end_steps[0] for all chunks when same_step_across_blocks), keep that step's x0, then one
    forward on it at t = 0 that fills the KV cache.  A frame remainder is not a chunk of its own: it rides with the first chunk.
    eps_list: the injected noise draws of the 'sde' branch, in call order."""
    net.clear_caches()
    B, Fr = noise.shape[0], noise.shape[2]
    cs = net.cfg.chunk_size
    nblocks, rem = Fr // cs, Fr % cs
    draws = iter(eps_list or [])
    outs = []
    for b in range(nblocks):
        start = 0 if b == 0 else cs * b + rem
        end = cs * (b + 1) + rem
        x = noise[:, :, start:end]
        exit_step = end_steps[0] if same_step_across_blocks else end_steps[b]
        for step in range(len(t_list)):
            tc = float(t_list[step])
            x0 = x - tc * net.forward(x, t_list[step].expand(B), text, cur_start_frame=start, store_kv=False)
            if step == exit_step:
                break
            tn = float(t_list[step + 1])
            eps = next(draws) if sample_type == "sde" else (x - (1 - tc) * x0) / tc
            x = (1 - tn) * x0 + tn * eps
        outs.append(x0)
        net.forward(x0, torch.zeros(B, dtype=torch.float64), text, cur_start_frame=start, store_kv=True)
    net.clear_caches()
    return torch.cat(outs, dim=2) if outs else torch.empty_like(noise)


def extrapolate(net: CausalWanRef, noise: torch.Tensor, t_list: torch.Tensor, text: torch.Tensor, num_segments: int, overlap_frames: int,
                vae_roundtrip, fresh_noise: list) -> torch.Tensor:
    """`CausVidModel.generator_fn_extrapolation` (causvid.py:188-397), 'ode' re-noising, context_noise = 0, RF schedule.
    vae_roundtrip(latents) -> latents: encode(decode(.)) of the caller's VAE on the overlapped tail; fresh_noise: the tensors the
    reference draws with randn_like for the segments after the first, in order."""
    B, _, T = noise.shape[:3]
    cs = net.cfg.chunk_size
    draws = iter(fresh_noise)
    t0 = float(t_list[0])

    def run_segment(lat, prefill):
        x = lat.clone()
        net.clear_caches()
        for start in range(0, prefill, cs):
            net.forward(x[:, :, start:min(start + cs, prefill)], torch.zeros(B, dtype=torch.float64), text, cur_start_frame=start, store_kv=True)
        x[:, :, prefill:] = x[:, :, prefill:] * t0  # latents = noise * sigma(t_list[0])
        for start in range(prefill, T, cs):
            end = min(start + cs, T)
            x_next = x[:, :, start:end]
            for step in range(len(t_list) - 1):
                tc = float(t_list[step])
                x_cur = x_next
                x_next = x_cur - tc * net.forward(x_cur, t_list[step].expand(B), text, cur_start_frame=start, store_kv=False)
                tn = float(t_list[step + 1])
                if tn > 0:
                    eps = (x_cur - (1 - tc) * x_next) / tc
                    x_next = (1 - tn) * x_next + tn * eps
            x[:, :, start:end] = x_next
            net.forward(x_next, torch.zeros(B, dtype=torch.float64), text, cur_start_frame=start, store_kv=True)
        net.clear_caches()
        return x

    segs, cur, prefill = [], noise, 0
    for i in range(num_segments):
        seg = run_segment(cur, prefill)
        segs.append(seg if i == 0 or overlap_frames == 0 else seg[:, :, overlap_frames:])
        if i == num_segments - 1:
            break
        if overlap_frames == 0:
            cur, prefill = next(draws), 0
            continue
        tail = vae_roundtrip(seg)[:, :, -overlap_frames:]
        if overlap_frames > 1:
            tail = torch.cat([tail[:, :, :1], seg[:, :, -(overlap_frames - 1):]], dim=2)
        cur = next(draws).clone()
        cur[:, :, :overlap_frames] = tail
        prefill = overlap_frames
    return torch.cat(segs, dim=2)
