"""Discriminator_EDM on the MI355X path — drop-in for `fastgen.networks.discriminators.Discriminator_EDM`
(fastgen/networks/discriminators.py:62-137): same constructor, same `feature_indices` / `in_res` attributes, the same module tree
(`discriminator_heads.{i}.{j}` = Conv2d / GroupNorm / SiLU in the reference's order) and therefore the same state-dict keys and
shapes; `forward(feats) -> [B, number of heads]` logits.  The torch sub-modules only hold the parameters: every head runs as one
`fg_disc_edm_run` call (csrc/disc.hip, bf16 activations, fp32 parameters), forward and — under autograd — backward with respect
to the parameters and the feature maps.  No CPU / torch fallback."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Set

import torch
from torch import nn

from .. import _lib


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, res, *params):
        L = _lib.lib()
        B, dev = feat.shape[0], feat.device
        f32 = feat.detach().to(torch.float32).contiguous()
        ps = [p.detach().to(torch.float32).contiguous() for p in params]
        arr = (ctypes.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
        need = L.fg_disc_edm_workspace_bytes(res, B)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        logits = torch.empty(B, dtype=torch.float32, device=dev)
        _lib.check(L.fg_disc_edm_run(ctypes.c_void_p(f32.data_ptr()), res, arr, ctypes.c_void_p(logits.data_ptr()), None, None, None, B,
                                     ctypes.c_void_p(ws.data_ptr()), need, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        ctx.res = res
        ctx.save_for_backward(f32, *ps)
        return logits.reshape(B, 1)

    @staticmethod
    def backward(ctx, dlogits):
        L = _lib.lib()
        f32, *ps = ctx.saved_tensors
        B, dev, res = f32.shape[0], f32.device, ctx.res
        arr = (ctypes.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
        grads = [torch.zeros_like(p) if ctx.needs_input_grad[2 + i] else None for i, p in enumerate(ps)]
        garr = (ctypes.c_void_p * len(ps))(*[g.data_ptr() if g is not None else None for g in grads])
        dfeat = torch.empty_like(f32) if ctx.needs_input_grad[0] else None
        need = L.fg_disc_edm_workspace_bytes(res, B)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        logits = torch.empty(B, dtype=torch.float32, device=dev)
        dl = dlogits.detach().to(torch.float32).reshape(B).contiguous()
        _lib.check(L.fg_disc_edm_run(ctypes.c_void_p(f32.data_ptr()), res, arr, ctypes.c_void_p(logits.data_ptr()),
                                     ctypes.c_void_p(dl.data_ptr()), ctypes.c_void_p(dfeat.data_ptr() if dfeat is not None else None),
                                     garr, B, ctypes.c_void_p(ws.data_ptr()), need,
                                     ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return (dfeat, None, *grads)


class Discriminator_EDM(nn.Module):
    def __init__(self, feature_indices: Optional[Set[int]] = None, all_res: List[int] = [32, 16, 8], in_channels: int = 256):
        super().__init__()
        if in_channels != 256:
            raise NotImplementedError("fastgen_amd Discriminator_EDM: in_channels must be 256 (the EDM encoder's width)")
        self.feature_indices = feature_indices
        if self.feature_indices is None:
            self.feature_indices = {len(all_res) - 1}  # the bottleneck feature (discriminators.py:72-73)
        self.feature_indices = {i for i in self.feature_indices if i < len(all_res)}
        self.in_res = [all_res[i] for i in sorted(self.feature_indices)]
        self.in_channels = in_channels
        self.discriminator_heads = nn.ModuleList()
        for res in self.in_res:
            if res not in (8, 16, 32):
                raise NotImplementedError(f"fastgen_amd Discriminator_EDM: resolution {res} is not covered (8, 16, 32)")
            layers: List[nn.Module] = []
            r = res
            while r > 8:  # halve down to 8x8 (discriminators.py:80-95)
                layers += [nn.Conv2d(in_channels, in_channels, kernel_size=4, stride=2, padding=1), nn.GroupNorm(32, in_channels), nn.SiLU()]
                r //= 2
            layers += [nn.Conv2d(in_channels, in_channels, kernel_size=4, stride=2, padding=1), nn.GroupNorm(32, in_channels), nn.SiLU(),
                       nn.Conv2d(in_channels, in_channels, kernel_size=4, stride=4, padding=0), nn.GroupNorm(32, in_channels), nn.SiLU(),
                       nn.Conv2d(in_channels, 1, kernel_size=1, stride=1, padding=0)]
            self.discriminator_heads.append(nn.Sequential(*layers))

    @staticmethod
    def _head_params(head: nn.Sequential) -> List[torch.Tensor]:
        mods = [m for m in head if not isinstance(m, nn.SiLU)]
        out: List[torch.Tensor] = []
        for m in mods:
            out += [m.weight, m.bias]
        return out

    def forward(self, feats: List[torch.Tensor]) -> torch.Tensor:
        assert isinstance(feats, list)
        if len(feats) != len(self.in_res):
            raise ValueError(f"Number of feature maps {len(feats)} does not match the number of resolutions {len(self.in_res)}")
        logits = []
        for i, res in enumerate(self.in_res):
            f = feats[i]
            assert res == f.shape[-1]
            if f.device.type != "cuda":
                raise RuntimeError("fastgen_amd Discriminator_EDM runs on a HIP GPU only (no CPU path)")
            if f.shape[1] != self.in_channels:
                raise ValueError(f"feature map {i} has {f.shape[1]} channels, expected {self.in_channels}")
            ps = self._head_params(self.discriminator_heads[i])
            if torch.is_grad_enabled() and (f.requires_grad or any(p.requires_grad for p in ps)):
                lg = _HeadFn.apply(f, res, *ps)
            else:
                with torch.no_grad():
                    lg = _HeadFn.apply(f, res, *ps)
            logits.append(lg.to(f.dtype).reshape(-1, 1))
        return torch.cat(logits, dim=1)
