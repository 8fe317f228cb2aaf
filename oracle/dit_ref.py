"""CPU oracle of the DiT forward (test infrastructure: imported by tests/, oracle/gen_golden.py and nothing else).

Functional restatement over a flat state dict with the reference's key names, each function citing the reference lines it
follows (fastgen/networks/DiT/network.py unless noted).  Three small classes of that file come from the un-vendored `timm`
dependency (`from timm.models.vision_transformer import PatchEmbed, Attention, Mlp`, DiT/network.py:15-16; `timm` is unpinned
in requirements.txt:20 and absent from this image).  Their published algorithm is restated here and in
`oracle/_timm_restated.py`, from which `oracle/gen_golden.py dit` builds the reference's own `DiT` class to record the golden
vectors - so for the three timm pieces parity is "restated", for everything else in the file it is pinned by running the
reference's code:

    PatchEmbed(img, patch, in_ch, dim, bias)   proj = Conv2d(in_ch, dim, kernel = stride = patch); flatten(2).transpose(1, 2)
    Attention(dim, heads, qkv_bias=True)       qkv = Linear(dim, 3 dim): reshape (B, N, 3, heads, hd).permute(2, 0, 3, 1, 4);
                                               softmax(q k^T / sqrt(hd)) v; transpose(1, 2).reshape(B, N, dim); proj = Linear(dim, dim)
    Mlp(dim, hidden, act=GELU(tanh))           fc2(act(fc1(x)))
"""
import math
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass(frozen=True)
class DiTConfig:
    input_size: int = 32
    patch_size: int = 2
    in_channels: int = 4
    hidden_size: int = 1152
    depth: int = 28
    num_heads: int = 16
    mlp_ratio: float = 4.0
    num_classes: int = 1000
    class_dropout_prob: float = 0.1   # > 0: the embedding table has one more row, the "unconditional" class (:116-118)
    r_timestep: bool = False
    time_cond_type: str = "abs"
    scale_t: bool = True
    use_sit_convention: bool = False


XL_2 = DiTConfig()                                                     # configs/net.py:124-127
S_2 = DiTConfig(hidden_size=384, depth=12, num_heads=6)               # configs/net.py:98-112


def pos_embed_2d(dim: int, grid: int) -> Tensor:
    """compute_sinusoidal_2d_embeddings / _encode_1d_positions, :712-791: [x-part (sin | cos) | y-part (sin | cos)], float64
    frequencies, float32 positions."""
    y, x = np.meshgrid(np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32), indexing="ij")

    def enc(pos, d):
        f = 1.0 / (10000.0 ** (np.arange(d // 2, dtype=np.float64) / float(d // 2)))
        a = np.outer(pos.reshape(-1), f)
        return np.concatenate([np.sin(a), np.cos(a)], axis=1)

    return torch.from_numpy(np.concatenate([enc(x, dim // 2), enc(y, dim // 2)], axis=1)).float()


def param_shapes(cfg: DiTConfig) -> Dict[str, tuple]:
    """State-dict entries of the reference DiT (:233-290), in module order; `pos_embed` is a persistent buffer."""
    D, p, C = cfg.hidden_size, cfg.patch_size, cfg.in_channels
    H = int(D * cfg.mlp_ratio)
    sh = {"pos_embed": (1, (cfg.input_size // p) ** 2, D),
          "x_embedder.proj.weight": (D, C, p, p), "x_embedder.proj.bias": (D,)}
    for e in ("t_embedder",) + (("r_embedder",) if cfg.r_timestep else ()):
        sh.update({f"{e}.proj_net.0.weight": (D, 256), f"{e}.proj_net.0.bias": (D,),
                   f"{e}.proj_net.2.weight": (D, D), f"{e}.proj_net.2.bias": (D,)})
    sh["y_embedder.class_embeddings.weight"] = (cfg.num_classes + (1 if cfg.class_dropout_prob > 0 else 0), D)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        sh.update({b + "attention.qkv.weight": (3 * D, D), b + "attention.qkv.bias": (3 * D,),
                   b + "attention.proj.weight": (D, D), b + "attention.proj.bias": (D,),
                   b + "feed_forward.fc1.weight": (H, D), b + "feed_forward.fc1.bias": (H,),
                   b + "feed_forward.fc2.weight": (D, H), b + "feed_forward.fc2.bias": (D,),
                   b + "conditioning_net.1.weight": (6 * D, D), b + "conditioning_net.1.bias": (6 * D,)})
    sh.update({"final_layer.projection.weight": (p * p * C, D), "final_layer.projection.bias": (p * p * C,),
               "final_layer.adaptive_params.1.weight": (2 * D, D), "final_layer.adaptive_params.1.bias": (2 * D,),
               "logvar_linear.weight": (1, D), "logvar_linear.bias": (1,)})
    return sh


def random_state_dict(cfg: DiTConfig, seed: int = 0) -> Dict[str, Tensor]:
    """Seeded weights with O(1) signal in every branch (the reference zero-initialises the conditioning and output layers,
    :321-331, which would hide wrong kernels - SURVEY H1): matrices N(0, 1/fan_in), biases 0.1 N(0, 1)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, s in param_shapes(cfg).items():
        if k == "pos_embed":
            sd[k] = pos_embed_2d(cfg.hidden_size, cfg.input_size // cfg.patch_size).unsqueeze(0)
        elif k.endswith(".bias"):
            sd[k] = 0.1 * torch.randn(s, generator=g)
        elif k.startswith("y_embedder"):
            sd[k] = 0.5 * torch.randn(s, generator=g)
        else:
            sd[k] = torch.randn(s, generator=g) / math.sqrt(int(np.prod(s[1:])))
    return sd


def fourier_features(t: Tensor, dim: int = 256, max_freq: float = 10000.0) -> Tensor:
    """FourierTimeEmbedding.encode_timesteps, :67-96: [cos | sin] of t * exp(-ln(max_freq) * i / half), fp32."""
    half = dim // 2
    freq = torch.exp(-math.log(max_freq) * torch.arange(0, half, dtype=torch.float32) / half)
    ang = t[:, None].float() * freq[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1).to(t.dtype)


def time_embedding(sd, prefix: str, t: Tensor) -> Tensor:
    """FourierTimeEmbedding.forward, :98-101: Linear -> SiLU -> Linear on the Fourier features."""
    h = F.linear(fourier_features(t), sd[prefix + ".proj_net.0.weight"], sd[prefix + ".proj_net.0.bias"])
    return F.linear(F.silu(h), sd[prefix + ".proj_net.2.weight"], sd[prefix + ".proj_net.2.bias"])


def layer_norm(x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)  # elementwise_affine=False, :167, 171, 212


def modulate(x: Tensor, shift: Tensor, scale: Tensor) -> Tensor:
    return x * (1.0 + scale.unsqueeze(1)) + shift.unsqueeze(1)  # apply_adaptive_modulation, :29-41


def attention(sd, b: str, x: Tensor, heads: int) -> Tensor:
    """timm Attention (restated, see the module docstring) as DiTBlock uses it (:168, 191)."""
    B, N, D = x.shape
    hd = D // heads
    qkv = F.linear(x, sd[b + "attention.qkv.weight"], sd[b + "attention.qkv.bias"]).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, D)
    return F.linear(o, sd[b + "attention.proj.weight"], sd[b + "attention.proj.bias"])


def dit_block(sd, i: int, x: Tensor, c: Tensor, heads: int) -> Tensor:
    """DiTBlock.forward, :184-201."""
    b = f"blocks.{i}."
    p = F.linear(F.silu(c), sd[b + "conditioning_net.1.weight"], sd[b + "conditioning_net.1.bias"]).chunk(6, dim=1)
    a_shift, a_scale, a_gate, f_shift, f_scale, f_gate = p
    x = x + a_gate.unsqueeze(1) * attention(sd, b, modulate(layer_norm(x), a_shift, a_scale), heads)
    h = modulate(layer_norm(x), f_shift, f_scale)
    h = F.gelu(F.linear(h, sd[b + "feed_forward.fc1.weight"], sd[b + "feed_forward.fc1.bias"]), approximate="tanh")
    h = F.linear(h, sd[b + "feed_forward.fc2.weight"], sd[b + "feed_forward.fc2.bias"])
    return x + f_gate.unsqueeze(1) * h


def dit_forward(sd, cfg: DiTConfig, x_t: Tensor, t: Tensor, condition: Tensor, r: Optional[Tensor] = None, trace=None) -> Tensor:
    """DiT.forward with fwd_pred_type = net_pred_type (identity conversion) in eval mode, :464-574.  t, r: the schedule's
    timesteps (float64 on entry, rescaled by num_steps = 1000 for the 'rf' schedule when scale_t, :457-462 and
    noise_schedule.py:1325-1326, then cast to x_t.dtype); condition: one-hot [B, num_classes] (an all-zero row is the
    unconditional class, :493-498) or class indices [B]."""
    B, C, Hh, Ww = x_t.shape
    p, D = cfg.patch_size, cfg.hidden_size
    if condition.ndim == 2:
        mask = torch.any(condition != 0, dim=1)
        condition = torch.where(~mask, cfg.num_classes, condition.argmax(dim=1))
    prep = lambda v: None if v is None else ((v * 1000.0) if cfg.scale_t else v).to(x_t.dtype)  # noqa: E731
    t_, r_ = prep(t), prep(r)
    if cfg.use_sit_convention:
        t_ = 1 - t_
    # PatchEmbed (restated): Conv2d(kernel = stride = patch) -> [B, N, D], + pos_embed (:511)
    x = F.conv2d(x_t, sd["x_embedder.proj.weight"], sd["x_embedder.proj.bias"], stride=p).flatten(2).transpose(1, 2) + sd["pos_embed"]
    t_emb = time_embedding(sd, "t_embedder", t_)
    if cfg.r_timestep and r_ is not None:
        r_emb = time_embedding(sd, "r_embedder", (t_ - r_) if cfg.time_cond_type == "diff" else r_)
    else:
        r_emb = torch.zeros_like(t_emb)
    c = t_emb + sd["y_embedder.class_embeddings.weight"][condition] + r_emb  # :517-533
    if trace is not None:
        trace["c"] = c
    for i in range(cfg.depth):
        x = dit_block(sd, i, x, c, cfg.num_heads)
        if trace is not None:
            trace[f"block{i}"] = x
    shift, scale = F.linear(F.silu(c), sd["final_layer.adaptive_params.1.weight"], sd["final_layer.adaptive_params.1.bias"]).chunk(2, dim=1)
    x = F.linear(modulate(layer_norm(x), shift, scale), sd["final_layer.projection.weight"], sd["final_layer.projection.bias"])
    g = Hh // p  # unpatchify, :437-455
    x = torch.einsum("bhwpqc->bchpwq", x.reshape(B, g, g, p, p, C)).reshape(B, C, g * p, g * p)
    if cfg.use_sit_convention:
        x = -x  # flow prediction under the SiT convention, :555-558
    return x
