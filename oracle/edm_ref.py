"""ORACLE — test infrastructure only.  NOT product code.

CPU (torch fp32 / fp64) restatement of the reference's few-step sampling hot path for the EDM
SongUNet ("DDPM++") network.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import this file; the product path (fastgen_amd/) never does and fails loudly when the HIP
library is missing.

Parity status: PINNED.  tests/golden/*.pt were produced by oracle/gen_golden.py, which imports the
reference itself (PYTHONPATH=/root/reference, inert stubs for absent third-party imports) and records
its outputs on seeded inputs; tests/test_oracle_golden.py checks this restatement against them.

Written as pure functions over a flat {reference state-dict key: tensor} mapping — a different shape
from the reference's nn.Module tree on purpose (independent restatement, not a copy).  Every function
names the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# Network description


@dataclass
class SongUNetConfig:
    """kwargs of EDM_CIFAR10_Config, fastgen/configs/net.py:29-48 + SongUNet defaults EDM/network.py:347-367."""

    img_resolution: int = 32
    img_channels: int = 3
    label_dim: int = 10
    augment_dim: int = 9
    model_channels: int = 128
    channel_mult: Sequence[int] = (2, 2, 2)
    channel_mult_emb: int = 4
    num_blocks: int = 4
    attn_resolutions: Sequence[int] = (16,)
    channel_mult_noise: int = 1
    sigma_data: float = 0.5
    sigma_shift: float = 0.0
    r_timestep: bool = False
    drop_precond: Optional[str] = None  # None | "input" | "output" | "both" (EDM/network.py:818, 929-934, 959-960)
    schedule: str = "edm"  # "edm" | "rf"  (noise_schedule.py:729-777 / 1306-1346)

    @property
    def emb_channels(self) -> int:
        return self.model_channels * self.channel_mult_emb

    @property
    def noise_channels(self) -> int:
        return self.model_channels * self.channel_mult_noise


CIFAR10 = SongUNetConfig()
# MeanFlow CIFAR-10 student: configs/methods/config_mean_flow.py:137-145 (r_timestep, label_dim 0) +
# configs/experiments/EDM/config_mf_cifar10.py:37-46,62 (drop_precond 'both', rf schedule, flow prediction, augment_dim 6)
CIFAR10_MEANFLOW = SongUNetConfig(label_dim=0, augment_dim=6, r_timestep=True, drop_precond="both", schedule="rf")


@dataclass
class BlockSpec:
    key: str  # state-dict prefix, e.g. "model.enc.32x32_block0"
    kind: str  # "conv" | "block" | "aux_norm" | "aux_conv"
    cin: int
    cout: int
    res: int  # OUTPUT resolution
    up: bool = False
    down: bool = False
    attn: bool = False
    skip_from: Optional[int] = None  # decoder: number of channels concatenated from the skip stack


def layout(cfg: SongUNetConfig):
    """Encoder / decoder module order of SongUNet with encoder_type=decoder_type='standard'.

    Follows EDM/network.py:417-486 (construction) which also fixes the iteration order of
    forward (:527-561).
    """
    enc: List[BlockSpec] = []
    cout = cfg.img_channels
    for level, mult in enumerate(cfg.channel_mult):
        res = cfg.img_resolution >> level
        if level == 0:
            cin, cout = cout, cfg.model_channels
            enc.append(BlockSpec(f"model.enc.{res}x{res}_conv", "conv", cin, cout, res))
        else:
            enc.append(BlockSpec(f"model.enc.{res}x{res}_down", "block", cout, cout, res, down=True))
        for idx in range(cfg.num_blocks):
            cin, cout = cout, cfg.model_channels * mult
            enc.append(
                BlockSpec(f"model.enc.{res}x{res}_block{idx}", "block", cin, cout, res, attn=res in cfg.attn_resolutions)
            )
    skips = [b.cout for b in enc]
    dec: List[BlockSpec] = []
    nlev = len(cfg.channel_mult)
    for level, mult in reversed(list(enumerate(cfg.channel_mult))):
        res = cfg.img_resolution >> level
        if level == nlev - 1:
            dec.append(BlockSpec(f"model.dec.{res}x{res}_in0", "block", cout, cout, res, attn=True))
            dec.append(BlockSpec(f"model.dec.{res}x{res}_in1", "block", cout, cout, res))
        else:
            dec.append(BlockSpec(f"model.dec.{res}x{res}_up", "block", cout, cout, res, up=True))
        for idx in range(cfg.num_blocks + 1):
            sk = skips.pop()
            cin, cout = cout + sk, cfg.model_channels * mult
            attn = idx == cfg.num_blocks and res in cfg.attn_resolutions
            dec.append(BlockSpec(f"model.dec.{res}x{res}_block{idx}", "block", cin, cout, res, attn=attn, skip_from=sk))
        if level == 0:
            dec.append(BlockSpec(f"model.dec.{res}x{res}_aux_norm", "aux_norm", cout, cout, res))
            dec.append(BlockSpec(f"model.dec.{res}x{res}_aux_conv", "aux_conv", cout, cfg.img_channels, res))
    return enc, dec


def param_shapes(cfg: SongUNetConfig) -> Dict[str, tuple]:
    """Every state-dict entry (name -> shape) of EDMPrecond(SongUNet) — parameters and the persistent
    `resample_filter` buffers (EDM/network.py:89-91).  Used to check names against the reference."""
    out: Dict[str, tuple] = {}
    C, E, N = cfg.model_channels, cfg.emb_channels, cfg.noise_channels
    cond = N * (2 if cfg.r_timestep else 1)
    if cfg.label_dim:
        out["model.map_label.weight"] = (cond, cfg.label_dim)
        out["model.map_label.bias"] = (cond,)
    if cfg.augment_dim:
        out["model.map_augment.weight"] = (cond, cfg.augment_dim)
    out["model.map_layer0.weight"] = (E, cond)
    out["model.map_layer0.bias"] = (E,)
    out["model.map_layer1.weight"] = (E, E)
    out["model.map_layer1.bias"] = (E,)
    enc, dec = layout(cfg)
    for b in enc + dec:
        k = b.key
        if b.kind in ("conv", "aux_conv"):
            out[f"{k}.weight"] = (b.cout, b.cin, 3, 3)
            out[f"{k}.bias"] = (b.cout,)
        elif b.kind == "aux_norm":
            out[f"{k}.weight"] = (b.cin,)
            out[f"{k}.bias"] = (b.cin,)
        else:
            out[f"{k}.norm0.weight"] = (b.cin,)
            out[f"{k}.norm0.bias"] = (b.cin,)
            out[f"{k}.conv0.weight"] = (b.cout, b.cin, 3, 3)
            out[f"{k}.conv0.bias"] = (b.cout,)
            if b.up or b.down:
                out[f"{k}.conv0.resample_filter"] = (1, 1, 2, 2)
            out[f"{k}.affine.weight"] = (b.cout, E)
            out[f"{k}.affine.bias"] = (b.cout,)
            out[f"{k}.norm1.weight"] = (b.cout,)
            out[f"{k}.norm1.bias"] = (b.cout,)
            out[f"{k}.conv1.weight"] = (b.cout, b.cout, 3, 3)
            out[f"{k}.conv1.bias"] = (b.cout,)
            if b.cin != b.cout or b.up or b.down:
                out[f"{k}.skip.weight"] = (b.cout, b.cin, 1, 1)
                out[f"{k}.skip.bias"] = (b.cout,)
                if b.up or b.down:
                    out[f"{k}.skip.resample_filter"] = (1, 1, 2, 2)
            if b.attn:
                out[f"{k}.norm2.weight"] = (b.cout,)
                out[f"{k}.norm2.bias"] = (b.cout,)
                out[f"{k}.qkv.weight"] = (3 * b.cout, b.cout, 1, 1)
                out[f"{k}.qkv.bias"] = (3 * b.cout,)
                out[f"{k}.proj.weight"] = (b.cout, b.cout, 1, 1)
                out[f"{k}.proj.bias"] = (b.cout,)
    out["model.logvar_linear.weight"] = (1, N)
    out["model.logvar_linear.bias"] = (1,)
    return out


def random_state_dict(cfg: SongUNetConfig, seed: int = 1234, dtype=torch.float32) -> Dict[str, Tensor]:
    """Seeded re-randomised weights (the reference default init scales conv1/proj/aux_conv by
    1e-5, which would hide a wrong kernel).  Scales keep activations O(1) through 33 residual blocks:
    matrices ~ N(0, 1/fan_in); norm gains ~ 1 + 0.1 N(0,1); biases ~ 0.1 N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("resample_filter"):
            sd[name] = torch.full(shape, 0.25, dtype=dtype)
        elif len(shape) == 1:
            base = 1.0 if (".norm" in name or "aux_norm" in name) and name.endswith("weight") else 0.0
            sd[name] = (base + 0.1 * torch.randn(shape, generator=g)).to(dtype)
        else:
            fan_in = int(np.prod(shape[1:]))
            sd[name] = (torch.randn(shape, generator=g) / math.sqrt(fan_in)).to(dtype)
    return sd


# ----------------------------------------------------------------------------------------------
# Layers (functional)


def group_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """GroupNorm.forward, EDM/network.py:133-149: groups = min(32, C // 4)."""
    c = x.shape[1]
    return F.group_norm(x.contiguous(), min(32, c // 4), w.to(x.dtype), b.to(x.dtype), eps)


def conv2d(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], up=False, down=False) -> Tensor:
    """Conv2d.forward non-fused branch with resample_filter=[1,1], EDM/network.py:113-126.

    up:   conv_transpose2d with (f*4)=ones(2,2), stride 2  == nearest-neighbour 2x replication.
    down: depthwise conv2d with f = 0.25*ones(2,2), stride 2 == 2x2 average pool.
    Resampling happens BEFORE the weight convolution; bias is added last.
    """
    c = x.shape[1]
    if up:
        f = torch.ones(c, 1, 2, 2, dtype=x.dtype)
        x = F.conv_transpose2d(x, f, groups=c, stride=2)
    if down:
        f = torch.full((c, 1, 2, 2), 0.25, dtype=x.dtype)
        x = F.conv2d(x, f, groups=c, stride=2)
    if w is not None:
        x = F.conv2d(x, w.to(x.dtype), padding=w.shape[-1] // 2)
    if b is not None:
        x = x + b.to(x.dtype).reshape(1, -1, 1, 1)
    return x


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """Linear.forward, EDM/network.py:47-51."""
    y = x @ w.to(x.dtype).t()
    return y if b is None else y + b.to(x.dtype)


def attention(qkv: Tensor, heads: int = 1) -> Tensor:
    """qkv split + AttentionOp + value product, EDM/network.py:160-168, 290-296.

    qkv: [B, 3C, H, W] -> reshape(B*heads, C/heads, 3, T): channel o = c*3 + j, j in {q,k,v}.
    w = softmax_k( q^T (k / sqrt(C)) ) in fp32; a[c,q] = sum_k w[q,k] v[c,k]."""
    B, C3, H, W = qkv.shape
    C = C3 // 3
    q, k, v = qkv.reshape(B * heads, C // heads, 3, H * W).unbind(2)
    w = torch.einsum("ncq,nck->nqk", q.float(), (k / math.sqrt(k.shape[1])).float()).softmax(dim=2).to(q.dtype)
    a = torch.einsum("nqk,nck->ncq", w, v)
    return a.reshape(B, C, H, W)


SKIP_SCALE = math.sqrt(0.5)  # block_kwargs.skip_scale, EDM/network.py:385
BLOCK_EPS = 1e-6  # block_kwargs.eps, EDM/network.py:386 (also aux_norm :483)


def unet_block(sd: Dict[str, Tensor], b: BlockSpec, x: Tensor, emb: Tensor, drop_keep: Optional[Tensor] = None) -> Tensor:
    """UNetBlock.forward with adaptive_scale=False, EDM/network.py:274-299.  drop_keep: the training-mode dropout of conv1's
    operand (:283-284, F.dropout(silu(norm1(x)), p)) as explicit keep factors (0 or 1/(1-p), same shape as the operand)."""
    k = b.key
    orig = x
    h = F.silu(group_norm(x, sd[f"{k}.norm0.weight"], sd[f"{k}.norm0.bias"], BLOCK_EPS))
    h = conv2d(h, sd[f"{k}.conv0.weight"], sd[f"{k}.conv0.bias"], up=b.up, down=b.down)
    h = h + linear(emb, sd[f"{k}.affine.weight"], sd[f"{k}.affine.bias"])[:, :, None, None]
    h = F.silu(group_norm(h, sd[f"{k}.norm1.weight"], sd[f"{k}.norm1.bias"], BLOCK_EPS))
    if drop_keep is not None:
        h = h * drop_keep
    h = conv2d(h, sd[f"{k}.conv1.weight"], sd[f"{k}.conv1.bias"])
    if f"{k}.skip.weight" in sd:
        orig = conv2d(orig, sd[f"{k}.skip.weight"], sd[f"{k}.skip.bias"], up=b.up, down=b.down)
    x = (h + orig) * SKIP_SCALE
    if b.attn:
        qkv = conv2d(
            group_norm(x, sd[f"{k}.norm2.weight"], sd[f"{k}.norm2.bias"], BLOCK_EPS),
            sd[f"{k}.qkv.weight"],
            sd[f"{k}.qkv.bias"],
        )
        a = attention(qkv)
        x = (conv2d(a, sd[f"{k}.proj.weight"], sd[f"{k}.proj.bias"]) + x) * SKIP_SCALE
    return x


def positional_embedding(t: Tensor, num_channels: int, max_positions: int = 10000) -> Tensor:
    """PositionalEmbedding(endpoint=True).forward, EDM/network.py:306-319 -> [cos | sin]."""
    half = num_channels // 2
    freqs = torch.arange(half, dtype=torch.float32) / (half - 1)
    freqs = (1.0 / max_positions) ** freqs
    ang = t.ger(freqs.to(t.dtype))
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def mapping(sd, cfg: SongUNetConfig, noise_labels: Tensor, class_labels: Optional[Tensor], r_noise_labels=None,
            augment_labels: Optional[Tensor] = None):
    """Embedding MLP, SongUNet.forward EDM/network.py:500-521; the augment branch (:518-519) only when the training-time
    augmentation pipeline supplies labels (None on the sampling path, :916-917)."""
    emb = positional_embedding(noise_labels, cfg.noise_channels)
    emb = emb.reshape(emb.shape[0], 2, -1).flip(1).reshape(*emb.shape)  # -> [sin | cos]
    if r_noise_labels is not None:
        er = positional_embedding(r_noise_labels, cfg.noise_channels)
        er = er.reshape(er.shape[0], 2, -1).flip(1).reshape(*er.shape)
        emb = torch.cat([emb, er], dim=-1)
    if cfg.label_dim:
        emb = emb + linear(class_labels * math.sqrt(cfg.label_dim), sd["model.map_label.weight"], sd["model.map_label.bias"])
    if augment_labels is not None and "model.map_augment.weight" in sd:
        emb = emb + linear(augment_labels, sd["model.map_augment.weight"], None)
    emb = F.silu(linear(emb, sd["model.map_layer0.weight"], sd["model.map_layer0.bias"]))
    emb = F.silu(linear(emb, sd["model.map_layer1.weight"], sd["model.map_layer1.bias"]))
    return emb


def song_unet(sd, cfg: SongUNetConfig, x: Tensor, noise_labels: Tensor, class_labels, trace: Optional[dict] = None,
              r_noise_labels: Optional[Tensor] = None, augment_labels: Optional[Tensor] = None, drop_keeps=None):
    """SongUNet.forward (standard encoder/decoder), EDM/network.py:489-574."""
    if r_noise_labels is not None and not cfg.r_timestep:
        raise ValueError("r_noise_labels provided, but r_timestep is not set")  # :510
    emb = mapping(sd, cfg, noise_labels, class_labels, r_noise_labels, augment_labels)
    if trace is not None:
        trace["emb"] = emb
    enc, dec = layout(cfg)
    skips: List[Tensor] = []
    for b in enc:
        if b.kind == "conv":
            x = conv2d(x, sd[f"{b.key}.weight"], sd[f"{b.key}.bias"])
        else:
            x = unet_block(sd, b, x, emb, None if drop_keeps is None else drop_keeps[b.key])
        skips.append(x)
        if trace is not None:
            trace[b.key] = x
    out = None
    tmp = None
    for b in dec:
        if b.kind == "aux_norm":
            tmp = group_norm(x, sd[f"{b.key}.weight"], sd[f"{b.key}.bias"], BLOCK_EPS)
        elif b.kind == "aux_conv":
            out = conv2d(F.silu(tmp), sd[f"{b.key}.weight"], sd[f"{b.key}.bias"])
        else:
            if x.shape[1] != b.cin:
                x = torch.cat([x, skips.pop()], dim=1)
            x = unet_block(sd, b, x, emb, None if drop_keeps is None else drop_keeps[b.key])
            if trace is not None:
                trace[b.key] = x
    return out


def edm_precond_forward(sd, cfg: SongUNetConfig, x_t: Tensor, t: Tensor, condition: Optional[Tensor], trace=None,
                        r: Optional[Tensor] = None, augment_labels: Optional[Tensor] = None, drop_keeps=None,
                        training: bool = False) -> Tensor:
    """EDMPrecond.forward with fwd_pred_type = net_pred_type (identity conversion), EDM/network.py:881-974 (training=True:
    the module in train() mode, which only changes sigma_shift here, :956 - dropout comes in through drop_keeps);
    precond_input :755-778 (clamp_min 1e-6 from the scheduler, :930), precond_output :781-805; drop_precond :929-934,
    :959-960.  t (and r) are float64 on entry; coefficients are computed in float64 and cast to x_t.dtype before use."""
    B = x_t.shape[0]
    t = t.to(torch.float64).reshape(-1)
    if r is not None:
        r = r.to(torch.float64).reshape(-1)
    if cfg.label_dim == 0:
        class_labels = None
    elif condition is None:
        class_labels = torch.zeros(1, cfg.label_dim, dtype=x_t.dtype)
    else:
        class_labels = condition.reshape(-1, cfg.label_dim)
    x_in, t_in, r_in = x_t, t, r
    if cfg.drop_precond not in ("input", "both"):
        c_in = (1.0 / (cfg.sigma_data**2 + t**2).sqrt()).to(x_t.dtype).reshape(B, 1, 1, 1)
        x_in = c_in * x_t
        t_in = t.clamp(min=1e-6).log() / 4
        if r is not None:
            r_in = r.clamp(min=1e-6).log() / 4
    t_in = t_in.to(x_t.dtype)
    if r_in is not None:
        r_in = r_in.to(x_t.dtype)
    F_x = song_unet(sd, cfg, x_in, t_in, class_labels, trace=trace, r_noise_labels=r_in, augment_labels=augment_labels,
                    drop_keeps=drop_keeps)  # drop_keeps: {block key: keep factors} of a training-mode call, or None
    if cfg.drop_precond in ("output", "both"):
        return F_x
    ts = t if training else t - cfg.sigma_shift  # the shift is applied in eval mode only (EDM/network.py:956)
    c_skip = (cfg.sigma_data**2 / (ts**2 + cfg.sigma_data**2)).to(x_t.dtype).reshape(B, 1, 1, 1)
    c_out = (ts * cfg.sigma_data / (ts**2 + cfg.sigma_data**2).sqrt()).to(x_t.dtype).reshape(B, 1, 1, 1)
    return c_skip * x_t + c_out * F_x


# ----------------------------------------------------------------------------------------------
# EDM noise schedule + sampler


def edm_sigmas(num_steps=1000, min_t=0.002, max_t=80.0, rho=7.0) -> Tensor:
    """EDMNoiseSchedule.__init__ sigma table, noise_schedule.py:752-756 (float64, increasing)."""
    ramp = torch.linspace(0, 1, num_steps, dtype=torch.float64)
    a, b = min_t ** (1 / rho), max_t ** (1 / rho)
    return torch.flip((b + ramp * (a - b)) ** rho, [0])


def edm_t_list(sample_steps: int, num_steps=1000, max_t=80.0) -> Tensor:
    """EDMNoiseSchedule.get_t_list, noise_schedule.py:940-973: indices linspace(998, 2, N+1).long()
    into the sigma table, last entry := 0, clamped to max_t."""
    sig = edm_sigmas(num_steps, max_t=max_t)
    lo, hi = int(0.002 * num_steps), int(0.998 * num_steps)
    idx = torch.linspace(hi, lo, sample_steps + 1).long()
    t = sig[idx].clone()
    t[-1] = 0.0
    return t.clamp(max=max_t)


def latents(noise: Tensor, t_init: Tensor) -> Tensor:
    """BaseNoiseSchedule.latents, noise_schedule.py:72-88 (sigma(t)=t for EDM)."""
    return (noise.to(torch.float64) * t_init.to(torch.float64)).to(noise.dtype)


def rf_t_list(sample_steps: int, max_t=0.999) -> Tensor:
    """RFNoiseSchedule.get_t_list = BaseNoiseSchedule.get_t_list, noise_schedule.py:259-272."""
    return torch.linspace(max_t, 0, sample_steps + 1, dtype=torch.float64).clamp(max=max_t)


def _alpha(tt: Tensor, schedule: str) -> Tensor:
    """alpha(t): 1 for EDM (noise_schedule.py:773-774), 1 - t for rectified flow (:1337-1338); sigma(t) = t for both."""
    return torch.ones_like(tt) if schedule == "edm" else 1 - tt


def forward_process(x: Tensor, eps: Tensor, t: Tensor, schedule: str = "edm") -> Tensor:
    """BaseNoiseSchedule.forward_process, noise_schedule.py:425-449."""
    tt = t.to(torch.float64).reshape(-1, *([1] * (x.dim() - 1)))
    return (x.to(torch.float64) * _alpha(tt, schedule) + eps.to(torch.float64) * tt).to(x.dtype)


def x0_to_eps(xt: Tensor, x0: Tensor, t: Tensor, clamp_min=1e-6, schedule: str = "edm") -> Tensor:
    """BaseNoiseSchedule.x0_to_eps, noise_schedule.py:544-574; non_zero_clamp :123-129."""
    tt = t.to(torch.float64).reshape(-1, *([1] * (xt.dim() - 1)))
    s = torch.where(tt >= 0, tt.clamp(min=clamp_min), tt.clamp(max=-clamp_min))
    return ((xt.to(torch.float64) - x0.to(torch.float64) * _alpha(tt, schedule)) / s).to(xt.dtype)


def student_sample_loop(sd, cfg, x: Tensor, t_list: Tensor, condition, sample_type="sde", eps_list=None, trace=None):
    """FastGenModel._student_sample_loop, methods/model.py:315-372.  'sde' noise is injected through
    eps_list (one tensor per non-final step) instead of torch.randn_like so CPU and GPU see the same
    numbers."""
    B = x.shape[0]
    x_pred = x
    step = 0
    for t_cur, t_next in zip(t_list[:-1], t_list[1:]):
        x_pred = edm_precond_forward(sd, cfg, x, t_cur.expand(B), condition)
        if trace is not None:
            trace.setdefault("x_pred", []).append(x_pred)
        if t_next > 0:
            if sample_type == "sde":
                eps = eps_list[step]
            elif sample_type == "ode":
                eps = x0_to_eps(x, x_pred, t_cur.expand(B), schedule=cfg.schedule)
            else:
                raise NotImplementedError(sample_type)
            x = forward_process(x_pred, eps, t_next.expand(B), cfg.schedule)
        step += 1
    return x_pred


def meanflow_sample_loop(sd, cfg, x: Tensor, t_list: Tensor, condition, sample_type="sde", eps_list=None, trace=None):
    """MeanFlowModel._student_sample_loop, methods/consistency_model/mean_flow.py:336-381: the network returns the
    average velocity u(x, t, r) ('flow' prediction).  'sde' noise is injected through eps_list as above."""
    B = x.shape[0]
    step = 0
    for t_cur, t_next in zip(t_list[:-1], t_list[1:]):
        if sample_type == "sde":
            delta_t = t_cur.reshape(1, 1, 1, 1).to(x.dtype)
            u = edm_precond_forward(sd, cfg, x, t_cur.expand(B), condition, r=torch.zeros_like(t_next.expand(B)))
            x = x - delta_t * u
            if t_next > 0:
                x = forward_process(x, eps_list[step], t_next.expand(B), cfg.schedule)
        elif sample_type == "ode":
            delta_t = (t_cur - t_next).reshape(1, 1, 1, 1).to(x.dtype)
            u = edm_precond_forward(sd, cfg, x, t_cur.expand(B), condition, r=t_next.expand(B))
            x = x - delta_t * u
        else:
            raise NotImplementedError(sample_type)
        if trace is not None:
            trace.setdefault("x", []).append(x)
        step += 1
    return x


def generator_fn(sd, cfg, noise: Tensor, condition, student_sample_steps=4, t_list=None, sample_type="sde",
                 eps_list=None, trace=None, loop="x0") -> Tensor:
    """FastGenModel.generator_fn, methods/model.py:374-420 (fp32, no autocast, no `data`); loop selects the class's
    _student_sample_loop: 'x0' (FastGenModel) or 'meanflow' (MeanFlowModel)."""
    with torch.inference_mode():
        if t_list is None:
            t_list = edm_t_list(student_sample_steps) if cfg.schedule == "edm" else rf_t_list(student_sample_steps)
        else:
            assert len(t_list) - 1 == student_sample_steps
            t_list = torch.as_tensor(t_list, dtype=torch.float64)
        assert t_list[-1].item() == 0
        x = latents(noise, t_list[0])
        fn = meanflow_sample_loop if loop == "meanflow" else student_sample_loop
        return fn(sd, cfg, x, t_list, condition, sample_type, eps_list, trace).to(noise.dtype)


def edm_sample(sd, cfg, noise: Tensor, condition=None, neg_condition=None, guidance_scale: Optional[float] = 5.0,
               num_steps: int = 50) -> Tensor:
    """EDMPrecond.sample — deterministic Euler sampler of the (teacher) network with optional classifier-free guidance,
    EDM/network.py:976-1026.  As in the reference, `d = (x - x0) / t` divides by the float64 timestep, so x (and every
    network evaluation after the first step) is float64 from the second step on."""
    sigmas = edm_t_list(num_steps)
    x = latents(noise, sigmas[0])
    for sigma, sigma_next in zip(sigmas[:-1], sigmas[1:]):
        t = sigma.expand(x.shape[0])
        if guidance_scale is not None and guidance_scale > 1.0 and neg_condition is not None:
            x0 = edm_precond_forward(sd, cfg, torch.cat([x, x], 0), torch.cat([t, t], 0),
                                     torch.cat([neg_condition, condition], 0).to(x.dtype))
            x0_uncond, x0_cond = x0.chunk(2)
            x0 = x0_uncond + guidance_scale * (x0_cond - x0_uncond)
        else:
            x0 = edm_precond_forward(sd, cfg, x, t, None if condition is None else condition.to(x.dtype))
        d = (x - x0) / t.reshape(-1, 1, 1, 1)
        x = x + (sigma_next - sigma).to(x.dtype) * d
    return x


def images_to_uint8(images: torch.Tensor) -> torch.Tensor:
    """The sample writer's conversion that follows generator_fn (scripts/fid/compute_fid_from_ckpts.py:199):
    fp32 `images * 127.5 + 128`, clip to [0, 255], truncate to uint8, NCHW -> NHWC.  (No reference function to import:
    it is one expression inside a script; pinned by the hand-derived known answers in tests/test_oracle_golden.py.)"""
    v = images.to(torch.float32) * 127.5 + 128
    return v.clip(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def conv_weight_grad(act: torch.Tensor, dy: torch.Tensor, ks: int) -> torch.Tensor:
    """Gradient of `Conv2d.forward` (EDM/network.py:93-126: F.conv2d(x, w, padding=ks // 2), no resampling) with respect
    to its weight, as autograd computes it in a training step.  act [B, Cin, H, W] (the conv's input operand), dy
    [B, Cout, H, W] (gradient of its output) -> [Cout, Cin, ks, ks]; fp32."""
    act = act.to(torch.float32)
    dy = dy.to(torch.float32)
    w = torch.zeros(dy.shape[1], act.shape[1], ks, ks, dtype=torch.float32, requires_grad=True)
    with torch.enable_grad():
        torch.nn.functional.conv2d(act, w, padding=ks // 2).backward(dy)
    return w.grad.detach()


def discriminator_edm(sd: Dict[str, Tensor], feats, in_res) -> Tensor:
    """Discriminator_EDM.forward (fastgen/networks/discriminators.py:62-137): per tapped resolution a head of strided 4x4 convs,
    each followed by GroupNorm(32, eps 1e-5) + SiLU, down to 1x1, then a 1x1 conv to one logit; logits concatenated per head.
    sd uses the reference's keys (`discriminator_heads.{i}.{j}.weight|bias`)."""
    logits = []
    for i, (res, x) in enumerate(zip(in_res, feats)):
        j, r = 0, res
        pre = f"discriminator_heads.{i}."
        while r > 4:  # stride-2 convs down to 4x4 (the reference's loop to 8x8 plus its fixed 8 -> 4 layer)
            x = F.conv2d(x, sd[f"{pre}{j}.weight"], sd[f"{pre}{j}.bias"], stride=2, padding=1)
            x = F.silu(F.group_norm(x, 32, sd[f"{pre}{j + 1}.weight"], sd[f"{pre}{j + 1}.bias"], eps=1e-5))
            j, r = j + 3, r // 2
        x = F.conv2d(x, sd[f"{pre}{j}.weight"], sd[f"{pre}{j}.bias"], stride=4, padding=0)
        x = F.silu(F.group_norm(x, 32, sd[f"{pre}{j + 1}.weight"], sd[f"{pre}{j + 1}.bias"], eps=1e-5))
        x = F.conv2d(x, sd[f"{pre}{j + 3}.weight"], sd[f"{pre}{j + 3}.bias"])
        logits.append(x.reshape(-1, 1))
    return torch.cat(logits, dim=1)


def edm_precond_jvp(sd, cfg: SongUNetConfig, x_t: Tensor, t: Tensor, condition, vx: Tensor, vt: Tensor, r: Optional[Tensor] = None,
                    vr: Optional[Tensor] = None):
    """(output, directional derivative) of EDMPrecond.forward along the tangents (vx, vt[, vr]) of (x_t, t[, r]): what
    `torch.func.jvp(net_wrapper, (x_t, t, r), tangents)` computes in MeanFlowModel._jvp / sCM (consistency_model/mean_flow.py:
    240-250, sCM.py:179), with the reference's hand-written AttentionOp.jvp (EDM/network.py:186-196) - here plain forward-mode
    AD through the functional restatement.  t / r and their tangents are given in x_t's dtype, as the reference's tangents are."""
    if r is None:
        return torch.func.jvp(lambda a, b: edm_precond_forward(sd, cfg, a, b, condition), (x_t, t), (vx, vt))
    return torch.func.jvp(lambda a, b, c: edm_precond_forward(sd, cfg, a, b, condition, r=c), (x_t, t, r), (vx, vt, vr))

