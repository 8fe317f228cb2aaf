"""Stand-ins for the three `timm` classes the reference's DiT imports (`from timm.models.vision_transformer import PatchEmbed,
Attention, Mlp`, fastgen/networks/DiT/network.py:15-16).  `timm` (unpinned, requirements.txt:20) is not vendored in the reference
and not installed in this image, so their published algorithm is RESTATED here - with timm's attribute names, because those are
the state-dict keys (`x_embedder.proj.*`, `blocks.N.attention.qkv.*` / `.proj.*`, `blocks.N.feed_forward.fc1.*` / `.fc2.*`).
Used ONLY by oracle/gen_golden.py to build the reference's own `DiT` class in the build container (test infrastructure).
Defaults as DiT uses them: no qk-norm, no dropout, non-fused attention math (DiT calls `timm.layers.set_fused_attn(False)`
by default, DiT/network.py:250,256)."""
import torch
import torch.nn as nn


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True, bias=True, **kw):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
        self.norm = nn.Identity()

    def forward(self, x):
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_norm=False, attn_drop=0.0, proj_drop=0.0, **kw):
        super().__init__()
        assert dim % num_heads == 0 and not qk_norm
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm, self.k_norm = nn.Identity(), nn.Identity()
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, attn_mask=None):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        q, k = self.q_norm(q), self.k_norm(k)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, **kw):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def install():
    """Register the restated classes under the module paths the reference imports them from."""
    import sys
    import types

    timm = types.ModuleType("timm")
    timm.__path__ = []
    layers = types.ModuleType("timm.layers")
    layers.set_fused_attn = lambda *a, **k: None
    models = types.ModuleType("timm.models")
    models.__path__ = []
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.PatchEmbed, vt.Attention, vt.Mlp = PatchEmbed, Attention, Mlp
    timm.layers, timm.models, models.vision_transformer = layers, models, vt
    sys.modules.update({"timm": timm, "timm.layers": layers, "timm.models": models, "timm.models.vision_transformer": vt})
