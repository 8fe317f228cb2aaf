"""`DiT` drop-in for the reference's `fastgen.networks.DiT.network.DiT` (class-conditional diffusion transformer, DiT/network.py:228-574),
backed by libfastgen_amd.so (`fg_dit_*`: token GEMMs on the conv kernel's transformer modes, LayerNorm + adaLN modulation, multi-head
attention, output projection - fastgen_amd/csrc/dit.hip, engine_dit.inc).

Same constructor kwargs, `forward(x_t, t, condition, r, ..., fwd_pred_type)` semantics, `.noise_scheduler`, `.sample()`, and the same
state-dict keys and shapes (`tests/golden/dit_*_state_dict_keys.txt`, recorded from the reference class; the names of the patch
embedding / attention / MLP layers are those of the `timm` classes the reference imports).  Select it with
`_target_: fastgen_amd.networks.DiT.network.DiT` in a config built on `DiT_IN256_*_Config` (fastgen/configs/net.py:98-127).

Inference / sampling: `forward`, and the three sampling loops as single library calls replayed as hipGraphs (`few_step_sample`:
`FastGenModel` / `MeanFlowModel._student_sample_loop`; `sample()`: the RF Euler sampler with classifier-free guidance) - `fg_dit_sampler_run`.
Raises (never falls back): autograd through the network, token counts other than 256, and any device but a HIP GPU.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Any, Dict, List, Optional, Set

import numpy as np
import torch
import torch.nn as nn

from fastgen_amd import _lib
from fastgen_amd.networks import _weights
from fastgen_amd.networks.EDM import network as _edm
from fastgen_amd.networks.network import FastGenNetwork
from fastgen_amd.networks.noise_schedule import NET_PRED_TYPES


def sinusoidal_2d_table(dim: int, grid: int) -> torch.Tensor:
    """`compute_sinusoidal_2d_embeddings` (DiT/network.py:712-791): [x (sin | cos) | y (sin | cos)] with float64 frequency bands."""
    y, x = np.meshgrid(np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32), indexing="ij")

    def enc(pos, d):
        f = 1.0 / (10000.0 ** (np.arange(d // 2, dtype=np.float64) / float(d // 2)))
        a = np.outer(pos.reshape(-1), f)
        return np.concatenate([np.sin(a), np.cos(a)], axis=1)

    return torch.from_numpy(np.concatenate([enc(x, dim // 2), enc(y, dim // 2)], axis=1)).float().unsqueeze(0)


class DiT(FastGenNetwork):
    def __init__(self, input_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16, mlp_ratio=4.0,
                 class_dropout_prob=0.1, enable_class_dropout=False, num_classes=1000, learn_sigma=False, r_timestep=False,
                 scale_t=True, time_cond_type="abs", net_pred_type="flow", schedule_type="rf", enable_fused_attn=False,
                 use_sit_convention=False, compute_dtype: Optional[str] = None, **model_kwargs):
        super().__init__(net_pred_type=net_pred_type, schedule_type=schedule_type, **model_kwargs)
        if learn_sigma:
            raise NotImplementedError("learn_sigma=True (SiT checkpoints with unused variance channels) is not implemented")
        if time_cond_type not in ("abs", "diff"):
            raise ValueError(f"Invalid time_cond_type: {time_cond_type}")
        self.input_size, self.patch_size, self.in_channels, self.out_channels = input_size, patch_size, in_channels, in_channels
        self.num_heads, self.num_classes, self.learn_sigma = num_heads, num_classes, learn_sigma
        self.img_resolution = input_size * 8
        self.scale_t, self.r_timestep, self.time_cond_type = scale_t, bool(r_timestep), time_cond_type
        self.use_sit_convention = use_sit_convention
        self.class_dropout_prob = class_dropout_prob if enable_class_dropout else 0.0  # ConditionalEmbedding.cfg_dropout_rate, :114
        self.compute_dtype = compute_dtype or os.environ.get("FASTGEN_AMD_COMPUTE_DTYPE") or None
        if self.compute_dtype not in (None, "fp32", "bf16x3", "bf16"):
            raise ValueError(f"compute_dtype must be 'fp32', 'bf16x3', 'bf16' or None, got {self.compute_dtype!r}")
        cfg = _lib.fg_dit_config()
        cfg.input_size, cfg.patch_size, cfg.in_channels, cfg.hidden_size = input_size, patch_size, in_channels, hidden_size
        cfg.depth, cfg.num_heads, cfg.mlp_hidden = depth, num_heads, int(hidden_size * mlp_ratio)
        cfg.embedding_rows = num_classes + (1 if class_dropout_prob > 0 else 0)  # :116-118
        cfg.r_timestep = int(self.r_timestep)
        self._cfg = cfg
        self.hidden_size = hidden_size
        self._engines: Dict[int, ctypes.c_void_p] = {}
        self._bound_sig: Dict[int, Dict[str, tuple]] = {}
        self._ws: Dict[int, torch.Tensor] = {}
        # module tree with the reference's key paths; names / shapes come from the library's own plan
        self._names: List[str] = []
        h = self._make_engine(_lib.FG_DTYPE_F32)
        L = _lib.lib()
        name, ndim, shape = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int64 * 4)()
        for i in range(L.fg_dit_num_params(h)):
            _lib.check(L.fg_dit_param_info(h, i, ctypes.byref(name), ctypes.byref(ndim), shape))
            full, shp = name.value.decode(), tuple(shape[j] for j in range(ndim.value))
            self._names.append(full)
            node, parts = self, full.split(".")
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _edm._Node())
                node = node._modules[p]
            if full == "pos_embed":
                self.register_buffer("pos_embed", sinusoidal_2d_table(hidden_size, input_size // patch_size), persistent=True)
            else:
                node.register_parameter(parts[-1], nn.Parameter(self._init_value(full, shp)))
        self._engines[_lib.FG_DTYPE_F32] = h

    @staticmethod
    def _init_value(name: str, shape) -> torch.Tensor:
        """The reference's `initialize_weights` (:291-331): xavier-uniform linears with zero biases, N(0, 0.02) embeddings and time
        MLPs, zero adaLN / output layers."""
        if name.endswith(".bias") or "conditioning_net" in name or name.startswith("final_layer"):
            return torch.zeros(shape)
        if name.startswith(("t_embedder", "r_embedder", "y_embedder")):
            return 0.02 * torch.randn(shape)
        fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
        return math.sqrt(6 / (fan_in + fan_out)) * (torch.rand(*shape) * 2 - 1)

    def _make_engine(self, dtype: int):
        cfg = _lib.fg_dit_config.from_buffer_copy(self._cfg)
        cfg.compute_dtype = dtype
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().fg_dit_create(ctypes.byref(cfg), ctypes.byref(h)))
        return h

    def __del__(self):
        try:
            for h in getattr(self, "_engines", {}).values():
                _lib.lib().fg_dit_destroy(h)
        except Exception:
            pass

    def reset_parameters(self):
        with torch.no_grad():
            tensors = dict(self.named_parameters())
            for n in self._names:
                if n != "pos_embed":
                    tensors[n].copy_(self._init_value(n, tuple(tensors[n].shape)))
            self.pos_embed.copy_(sinusoidal_2d_table(self.hidden_size, self.input_size // self.patch_size))
        super().reset_parameters()

    def _select_dtype(self) -> int:
        if self.compute_dtype is not None:
            return _lib.DTYPE_NAMES[self.compute_dtype]
        if torch.is_autocast_enabled():
            ad = torch.get_autocast_gpu_dtype()
            if ad == torch.bfloat16:
                return _lib.FG_DTYPE_BF16
            if ad != torch.float32:
                raise NotImplementedError(f"autocast dtype {ad} is not implemented (bf16 or fp32)")
        return _lib.DTYPE_NAMES[_edm.DEFAULT_FP32_MODE]

    def _engine(self, device: torch.device):
        dt = self._select_dtype()
        if dt not in self._engines:
            self._engines[dt] = self._make_engine(dt)
        h = self._engines[dt]
        L = _lib.lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        _weights.sync_weights(
            self, self._names, self._bound_sig.setdefault(dt, {}),
            tensors=lambda: {**dict(self.named_parameters()), "pos_embed": self.pos_embed},
            bind=lambda n, q: _lib.check(L.fg_dit_bind_param(h, n.encode(), ctypes.c_void_p(q.data_ptr()), q.numel())),
            pack_group=lambda pre, exc: _lib.check(L.fg_dit_pack_group(h, pre.encode(), exc.encode() if exc else None, stream)))
        return dt, h

    def fully_shard(self, **kwargs):
        """FSDP2 with the reference's grouping (DiT/network.py:402-420): one parameter group per transformer block, one each for the
        patch / time / class (/ r) embedders and the output layer; `pos_embed` and `logvar_linear` stay whole on the root, as there.
        Between calls every grouped parameter is a sharded DTensor - state dict, optimizer and checkpointing see what they see in the
        reference.  The engine keeps its own packed copy of the weights (fastgen_amd/networks/_weights.py): when a group's parameters
        have changed it is all-gathered, packed and resharded, one group at a time with the next group's all-gather in flight."""
        from torch.distributed.fsdp import fully_shard

        for block in self._modules["blocks"]._modules.values():
            fully_shard(block, **kwargs)
        for name in ("x_embedder", "t_embedder", "y_embedder", "final_layer", "r_embedder"):
            if name in self._modules:
                fully_shard(self._modules[name], **kwargs)

    def _class_ids(self, condition: Optional[torch.Tensor], B: int, dev) -> torch.Tensor:
        """one-hot -> class index, an all-zero row -> the extra "unconditional" row (:493-498); training-time label dropout (:120-149)."""
        if condition is None:
            raise ValueError("DiT needs a class condition (one-hot [B, num_classes] or class indices [B])")
        if condition.ndim == 2:
            mask = torch.any(condition != 0, dim=1)
            if self._cfg.embedding_rows == self.num_classes and not bool(mask.all()):
                # (the table has the extra row only for class_dropout_prob > 0, :116-118; the reference's nn.Embedding device-asserts here)
                raise ValueError("an all-zero (unconditional) condition row needs the extra row of y_embedder.class_embeddings, "
                                 "which exists only for class_dropout_prob > 0")
            condition = torch.where(~mask, self.num_classes, condition.argmax(dim=1))
        cls = condition.to(device=dev, dtype=torch.int64)
        if self.training and self.class_dropout_prob > 0:
            cls = torch.where(torch.rand(B, device=dev) < self.class_dropout_prob, self.num_classes, cls)
        if int(cls.numel()) == 1 and B > 1:
            cls = cls.expand(B)
        return cls.contiguous()

    def prepare_t(self, t: Optional[torch.Tensor], dtype) -> Optional[torch.Tensor]:
        if t is None:
            return None
        if self.scale_t:
            t = self.noise_scheduler.rescale_t(t)
        return t.to(dtype=dtype)

    def forward(self, x_t: torch.Tensor, t: torch.Tensor, condition: Optional[torch.Tensor] = None, r: Optional[torch.Tensor] = None,
                return_features_early: bool = False, feature_indices: Optional[Set[int]] = None, return_logvar: bool = False,
                fwd_pred_type: Optional[str] = None, **fwd_kwargs):
        if feature_indices is None:
            feature_indices = {}
        if fwd_kwargs:
            raise TypeError(f"unexpected forward kwargs: {sorted(fwd_kwargs)}")
        if fwd_pred_type is None:
            fwd_pred_type = self.net_pred_type
        else:
            assert fwd_pred_type in NET_PRED_TYPES, f"{fwd_pred_type} is not supported as fwd_pred_type"
        if torch.is_grad_enabled() and (x_t.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("fastgen_amd.DiT: the backward pass is not implemented; call under torch.no_grad() / "
                                      "torch.inference_mode() (sampling)")
        if x_t.device.type != "cuda":
            raise RuntimeError("fastgen_amd runs on a HIP GPU only (no CPU path); got a tensor on " + str(x_t.device))
        B, dev = x_t.shape[0], x_t.device
        if tuple(x_t.shape[1:]) != (self.in_channels, self.input_size, self.input_size):
            raise ValueError(f"x_t must be [B,{self.in_channels},{self.input_size},{self.input_size}], got {tuple(x_t.shape)}")
        cls = self._class_ids(condition, B, dev)
        x32 = x_t.detach().to(torch.float32).contiguous()
        t_in = torch.atleast_1d(t.detach()).to(dev)
        if t_in.numel() == 1 and B > 1:
            t_in = t_in.expand(B)
        t_e = self.prepare_t(t_in, torch.float32)
        r_e = None
        if self.r_timestep and r is not None:
            r_in = torch.atleast_1d(r.detach()).to(dev)
            if r_in.numel() == 1 and B > 1:
                r_in = r_in.expand(B)
            r_e = self.prepare_t(r_in, torch.float32)
        if self.use_sit_convention:
            t_e = 1 - t_e  # :503-504
        if r_e is not None and self.time_cond_type == "diff":
            r_e = t_e - r_e  # :520-521
        t_e = t_e.contiguous()
        r_e = r_e.contiguous() if r_e is not None else None
        dt, h = self._engine(dev)
        L = _lib.lib()
        need = L.fg_dit_workspace_bytes(h, B)
        ws = self._ws.get(dt)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = self._ws[dt] = torch.empty(need, dtype=torch.uint8, device=dev)
        # block-output taps (:536-543): token tensors [B, 256, D] behind the requested blocks; with return_features_early the call
        # stops at the last of them (no requested index: after the first block, an empty list - as the reference's loop does)
        depth = self._cfg.depth
        taps = sorted(i for i in set(feature_indices) if 0 <= i < depth)
        if return_features_early and len(taps) == len(feature_indices) and not taps:
            return []
        early = return_features_early and len(taps) == len(feature_indices)  # (an index past the last block: the reference never returns early)
        feats = [torch.empty(B, 256, self.hidden_size, dtype=torch.float32, device=dev) for _ in taps]
        out = None if early else torch.empty_like(x32)
        _lib.check(L.fg_dit_forward_features(
            h, ctypes.c_void_p(x32.data_ptr()), ctypes.c_void_p(t_e.data_ptr()), ctypes.c_void_p(r_e.data_ptr() if r_e is not None else None),
            ctypes.c_void_p(cls.data_ptr()), ctypes.c_void_p(out.data_ptr() if out is not None else None), None,
            (ctypes.c_int * len(taps))(*taps) if taps else None, (ctypes.c_void_p * len(taps))(*[f.data_ptr() for f in feats]) if taps else None,
            len(taps), B, ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        feats = [f.to(x_t.dtype) for f in feats]
        if early:
            return feats
        out = out.to(x_t.dtype)
        if self.use_sit_convention and self.net_pred_type == "flow":
            out = -out  # :555-558
        out = self.noise_scheduler.convert_model_output(x_t, out, t_in, src_pred_type=self.net_pred_type, target_pred_type=fwd_pred_type)
        if len(feature_indices) != 0:
            out = [out, feats]
        if return_logvar:
            return out, self._logvar(t_e)
        return out

    def _logvar(self, t_e: torch.Tensor) -> torch.Tensor:
        """logvar_linear(t_embedder(t)) (:568-570) - a [B, D] x [D, 1] product on the module's own parameters, evaluated with torch."""
        half = 128
        freq = torch.exp(-math.log(10000.0) * torch.arange(0, half, dtype=torch.float32, device=t_e.device) / half)
        ang = t_e[:, None].float() * freq[None, :]
        f = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
        te = self._modules["t_embedder"]._modules["proj_net"]
        hmid = torch.nn.functional.silu(f @ te._modules["0"].weight.t() + te._modules["0"].bias)
        t_emb = hmid @ te._modules["2"].weight.t() + te._modules["2"].bias
        lv = self._modules["logvar_linear"]
        return t_emb @ lv.weight.t() + lv.bias

    # ---- the sampling loops as one library call ----------------------------------------------------------------------------
    def supports_fused_loop(self, kind: str) -> bool:
        """Which loops `fg_dit_sampler_run` restates for this network: 'x0' (FastGenModel._student_sample_loop, flow- or
        x0-predicting network), 'meanflow' (MeanFlowModel._student_sample_loop: flow-predicting r_timestep network), 'euler' (`sample()`)."""
        if self.net_pred_type not in ("flow", "x0") or self.schedule_type not in ("rf", "rectified_flow", "edm"):
            return False
        if kind == "x0":
            return True
        if kind == "meanflow":
            return self.r_timestep and self.net_pred_type == "flow" and not self.use_sit_convention
        return kind == "euler" and self.net_pred_type == "flow"

    def few_step_sample(self, noise: torch.Tensor, condition: Optional[torch.Tensor], t_list, sample_type: str = "sde",
                        eps: Optional[torch.Tensor] = None, seed: Optional[int] = None, use_graph: bool = True, loop: str = "x0",
                        neg_condition: Optional[torch.Tensor] = None, guidance_scale: float = 1.0) -> torch.Tensor:
        """The whole sampling loop as ONE `fg_dit_sampler_run` call / one hipGraph replay: latents = noise * sigma(t_list[0]), then per
        step the network call and the schedule arithmetic of `loop` ('x0' | 'meanflow' | 'euler').  t_list: steps + 1 decreasing
        timesteps ending in 0; 'sde' re-noises with `eps` ([steps - 1, B, C, H, W], injected) or device normals drawn from `seed`."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("the fused sampler builds no autograd graph; call it under torch.no_grad() / torch.inference_mode()")
        if not self.supports_fused_loop(loop):
            raise NotImplementedError(f"the fused sampler has no loop {loop!r} for net_pred_type={self.net_pred_type!r}, "
                                      f"r_timestep={self.r_timestep}, schedule_type={self.schedule_type!r}")
        if noise.device.type != "cuda":
            raise RuntimeError("fastgen_amd runs on a HIP GPU only (no CPU path); got a tensor on " + str(noise.device))
        if sample_type not in ("sde", "ode"):
            raise NotImplementedError(f"student_sample_type must be one of 'sde', 'ode' but got {sample_type}")
        B, dev = noise.shape[0], noise.device
        if tuple(noise.shape[1:]) != (self.in_channels, self.input_size, self.input_size):
            raise ValueError(f"noise must be [B,{self.in_channels},{self.input_size},{self.input_size}], got {tuple(noise.shape)}")
        tl = [float(v) for v in (t_list.tolist() if isinstance(t_list, torch.Tensor) else t_list)]
        steps = len(tl) - 1
        assert tl[-1] == 0, "t_list[-1] must be zero"
        n32 = noise if (noise.dtype == torch.float32 and noise.is_contiguous()) else noise.to(torch.float32).contiguous()
        cls = self._class_ids(condition, B, dev)
        neg = self._class_ids(neg_condition, B, dev) if (loop == "euler" and neg_condition is not None) else None
        if eps is not None:
            eps = eps.to(device=dev, dtype=torch.float32).contiguous()
            if eps.numel() != max(steps - 1, 0) * n32.numel():
                raise ValueError(f"eps must hold steps-1 = {steps - 1} noise tensors shaped like `noise`")
        if seed is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())  # host RNG: follows torch.manual_seed / set_random_seed
        sc = _lib.fg_dit_sampler_config()
        rf = self.schedule_type != "edm"
        sc.t_scale = float(self.noise_scheduler.num_steps) if (self.scale_t and rf) else 1.0  # rescale_t (noise_schedule.py:140-148)
        sc.guidance_scale = float(guidance_scale)
        sc.use_sit_convention, sc.time_cond_diff = int(self.use_sit_convention), int(self.time_cond_type == "diff")
        sc.net_pred_flow = int(self.net_pred_type == "flow")
        sc.schedule = _lib.FG_SCHEDULE_RF if rf else _lib.FG_SCHEDULE_EDM
        dt, h = self._engine(dev)
        L = _lib.lib()
        need = L.fg_dit_sampler_workspace_bytes(h, B, int(neg is not None))
        ws = self._ws.get(dt)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = self._ws[dt] = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty_like(n32)
        self._keep = (n32, cls, neg, eps, out)  # graph replays read these buffers; keep them alive
        p = lambda a: ctypes.c_void_p(a.data_ptr() if a is not None and a.numel() else None)  # noqa: E731
        _lib.check(L.fg_dit_sampler_run(
            h, ctypes.byref(sc), p(n32), p(cls), p(neg), (ctypes.c_double * (steps + 1))(*tl), steps,
            _lib.FG_SAMPLE_SDE if sample_type == "sde" else _lib.FG_SAMPLE_ODE,
            {"x0": _lib.FG_LOOP_X0, "meanflow": _lib.FG_LOOP_MEANFLOW, "euler": _lib.FG_LOOP_EULER}[loop], p(eps), ctypes.c_uint64(seed), p(out), B,
            ctypes.c_void_p(ws.data_ptr()), ws.numel(), 1 if use_graph else 0, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out.to(noise.dtype)

    @torch.no_grad()
    def sample(self, noise: torch.Tensor, condition: Optional[torch.Tensor] = None, neg_condition: Optional[torch.Tensor] = None,
               guidance_scale: Optional[float] = 5.0, num_steps: int = 50, use_graph: bool = True, **kwargs) -> torch.Tensor:
        """Euler sampler of the flow ODE with optional classifier-free guidance (`_sample_flow`, :605-651): one `fg_dit_sampler_run`
        call (FG_LOOP_EULER).  x += fp32(t_next - t) * v per step; guided: v = v_uncond + g (v_cond - v_uncond) from one forward of the
        doubled batch."""
        if self.schedule_type != "rf":
            raise NotImplementedError(f"sample() is implemented for schedule_type='rf', got {self.schedule_type!r}")
        t_list = self.noise_scheduler.get_t_list(num_steps, device="cpu")
        guided = guidance_scale is not None and guidance_scale > 1.0 and neg_condition is not None
        return self.few_step_sample(noise, condition, t_list, sample_type="ode", loop="euler", use_graph=use_graph,
                                    neg_condition=neg_condition if guided else None, guidance_scale=guidance_scale if guided else 1.0)
