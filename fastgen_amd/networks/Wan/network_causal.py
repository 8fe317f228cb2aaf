"""`CausalWan` drop-in for the reference's `fastgen.networks.Wan.network_causal.CausalWan` (the causal video DiT that
`CausVidModel._student_sample_loop` drives chunk by chunk, fastgen/methods/distribution_matching/causvid.py:87-185), backed by
libfastgen_amd.so (`fg_wan_*`: bf16 token GEMMs on gemm.hip, per-frame adaLN, RMSNorm + RoPE with a frame offset, KV-cache
attention, output projection - fastgen_amd/csrc/wan.hip, engine_wan.inc).

The reference builds its transformer with `WanTransformer3DModel.from_pretrained(model_id)` (Wan/network.py:641-693: hub
weights and config, unavailable offline); here the fields of that config are constructor keywords (defaults = Wan2.1-T2V-1.3B) and
the parameters carry the same state-dict keys (`transformer.` + diffusers' names), so a converted checkpoint loads with
`load_state_dict`.  PARITY UNPINNED (the arithmetic lives in un-vendored diffusers: oracle/wan_ref.py restates it).

Autoregressive inference (`is_ar=True`, one cache tag), the teacher- / diffusion-forcing forward over all total_num_frames frames
(`is_ar=False`: block-wise causal mask, per-frame timesteps [B, F]), and the whole chunk-by-chunk student loop as one library call
replayed as per-chunk hipGraphs (`student_sample` -> `fg_wan_sampler_run`; CausVidModel / SelfForcingModel call it).  Raises (never falls back): autograd,
feature taps, r / image conditioning, `is_ar=False` on fewer frames, any device but a HIP GPU.
"""
from __future__ import annotations

import ctypes
import weakref
from typing import Any, Dict, Optional, Set

import torch
import torch.nn as nn

from fastgen_amd import _lib
from fastgen_amd.networks import _weights
from fastgen_amd.networks.EDM import network as _edm
from fastgen_amd.networks.network import FastGenNetwork
from fastgen_amd.networks.noise_schedule import NET_PRED_TYPES


class CausalWan(FastGenNetwork):
    def __init__(self, num_attention_heads: int = 12, attention_head_dim: int = 128, in_channels: int = 16, out_channels: int = 16,
                 text_dim: int = 4096, freq_dim: int = 256, ffn_dim: int = 8960, num_layers: int = 30, eps: float = 1e-6,
                 rope_max_seq_len: int = 1024, r_timestep: bool = False, net_pred_type: str = "flow", schedule_type: str = "rf",
                 chunk_size: int = 3, total_num_frames: int = 21, enable_logvar_linear: bool = True, **model_kwargs):
        for k in ("model_id_or_local_path", "load_pretrained", "disable_efficient_attn", "disable_grad_ckpt", "use_fsdp_checkpoint",
                  "r_embedder_init", "time_cond_type", "norm_temb", "encoder_depth", "delete_cache_on_clear"):
            v = model_kwargs.pop(k, None)  # reference knobs without meaning here (hub ids, autograd plumbing) ...
            if k in ("norm_temb",) and v:  # ... except those that change the arithmetic
                raise NotImplementedError(f"{k}={v!r} is not implemented")
            if k == "encoder_depth" and v is not None:
                raise NotImplementedError("encoder_depth is not implemented")
        super().__init__(net_pred_type=net_pred_type, schedule_type=schedule_type, **model_kwargs)
        if r_timestep:
            raise NotImplementedError("r_timestep=True (a second time embedder) is not implemented for the causal video DiT path")
        self.chunk_size, self.total_num_frames = chunk_size, total_num_frames
        cfg = _lib.fg_wan_config()
        cfg.num_heads, cfg.head_dim, cfg.in_channels, cfg.out_channels = num_attention_heads, attention_head_dim, in_channels, out_channels
        cfg.text_dim, cfg.freq_dim, cfg.ffn_dim, cfg.num_layers = text_dim, freq_dim, ffn_dim, num_layers
        cfg.rope_max_seq_len, cfg.chunk_size, cfg.total_num_frames, cfg.eps = rope_max_seq_len, chunk_size, total_num_frames, eps
        self._cfg, self.in_channels = cfg, in_channels
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().fg_wan_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        self._names, self._bound_sig, self._ws = [], {}, None
        self._text_key = None
        L = _lib.lib()
        name, ndim, shape = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int64 * 5)()
        g = torch.Generator().manual_seed(0)
        for i in range(L.fg_wan_num_params(h)):
            _lib.check(L.fg_wan_param_info(h, i, ctypes.byref(name), ctypes.byref(ndim), shape))
            full, shp = name.value.decode(), tuple(shape[j] for j in range(ndim.value))
            if "logvar_linear" in full and not enable_logvar_linear:
                continue
            self._names.append(full)
            node, parts = self, full.split(".")
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _edm._Node())
                node = node._modules[p]
            node.register_parameter(parts[-1], nn.Parameter(self._init_value(full, shp, g)))

    @staticmethod
    def _init_value(name: str, shape, g) -> torch.Tensor:
        """torch defaults of the modules diffusers builds (Linear / Conv3d: uniform(+-1/sqrt(fan_in)); RMSNorm / LayerNorm weights 1,
        biases of norms 0; scale_shift_table randn / sqrt(dim))."""
        if name.endswith("scale_shift_table"):
            return torch.randn(shape, generator=g) / shape[-1] ** 0.5
        if name.endswith(("norm_q.weight", "norm_k.weight", "norm2.weight")):
            return torch.ones(shape)
        if name.endswith("norm2.bias"):
            return torch.zeros(shape)
        fan_in = 1
        for s_ in (shape[1:] if len(shape) > 1 else shape):
            fan_in *= s_
        return (torch.rand(shape, generator=g) * 2 - 1) * fan_in ** -0.5

    def __del__(self):
        try:
            _lib.lib().fg_wan_destroy(self._h)
        except Exception:
            pass

    # ---- library plumbing -------------------------------------------------------------------------------------------------
    def _stream(self, dev):
        return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def _bind(self, dev: torch.device):
        L = _lib.lib()
        names = [n for n in self._names if "logvar_linear" not in n]
        changed = _weights.sync_weights(
            self, names, self._bound_sig, tensors=lambda: dict(self.named_parameters()),
            bind=lambda n, q: _lib.check(L.fg_wan_bind_param(self._h, n.encode(), ctypes.c_void_p(q.data_ptr()), q.numel())),
            pack_group=lambda pre, exc: _lib.check(L.fg_wan_pack_group(self._h, pre.encode(), exc.encode() if exc else None, self._stream(dev))))
        if changed:
            self._text_key = None  # the text caches were computed with the previous weights

    def fully_shard(self, **kwargs):
        """FSDP2 with the reference's grouping (Wan/network.py:761-782): one parameter group per transformer block, then the transformer
        itself (embedders, output layer, `logvar_linear`) as the root group.  The engine keeps its own packed bf16 copy of the weights
        (fastgen_amd/networks/_weights.py): a group whose parameters changed is all-gathered, packed and resharded - ONE block's fp32
        parameters whole at a time, which is what carries to the 14B network (40 blocks of 350M parameters)."""
        from torch.distributed.fsdp import fully_shard

        tr = self._modules["transformer"]
        for block in tr._modules["blocks"]._modules.values():
            fully_shard(block, **kwargs)
        fully_shard(tr, **kwargs)

    def _workspace(self, dev, B, F, H, W) -> torch.Tensor:
        need = _lib.lib().fg_wan_workspace_bytes(self._h, B, F, H, W)
        if need == 0:
            raise ValueError(f"bad chunk shape: batch {B}, frames {F}, {H}x{W} (height and width must be even)")
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws

    def _set_text(self, condition: torch.Tensor, B: int, dev, ws: torch.Tensor) -> None:
        """The text condition: embedded (and its cross-attention k / v cached) once per tensor, as the reference's static cache
        (the same tensor OBJECT at the same in-place version; a new object with recycled storage is embedded again)."""
        ver = 0 if condition.is_inference() else condition._version
        same = self._text_key is not None and self._text_key[0]() is condition and self._text_key[1:] == (ver, tuple(condition.shape))
        if same:
            return
        c32 = condition.detach().to(device=dev, dtype=torch.float32).contiguous()
        if c32.shape[0] != B:
            raise ValueError(f"condition batch {c32.shape[0]} != {B}")
        _lib.check(_lib.lib().fg_wan_set_text(self._h, ctypes.c_void_p(c32.data_ptr()), B, c32.shape[1], ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                              self._stream(dev)))
        self._text_key = (weakref.ref(condition), ver, tuple(condition.shape))

    def clear_caches(self) -> None:
        """`CausalWan.clear_caches` (network_causal.py:1030-1054)."""
        if torch.cuda.is_available():
            _lib.check(_lib.lib().fg_wan_clear_caches(self._h, self._stream(torch.device("cuda", torch.cuda.current_device()))))
        self._text_key = None

    def forward(self, x_t: torch.Tensor, t: torch.Tensor, condition: Optional[Any] = None, r: Optional[torch.Tensor] = None,
                return_features_early: bool = False, feature_indices: Optional[Set[int]] = None, return_logvar: bool = False,
                unpatchify_features: bool = True, fwd_pred_type: Optional[str] = None, skip_layers=None, cache_tag: str = "pos",
                cur_start_frame: int = 0, store_kv: bool = False, is_ar: bool = False, **fwd_kwargs):
        if feature_indices or return_features_early or return_logvar or skip_layers or r is not None:
            raise NotImplementedError("feature taps / logvar / skip_layers / r are not implemented for the causal video DiT path")
        if fwd_kwargs:
            raise TypeError(f"unexpected forward kwargs: {sorted(fwd_kwargs)}")
        if not is_ar and x_t.shape[2] != self.total_num_frames:
            # (the reference builds its block mask only for a call over all total_num_frames frames, network_causal.py:673-680)
            raise NotImplementedError(f"is_ar=False (block-wise causal mask) takes all total_num_frames = {self.total_num_frames} frames, got "
                                      f"{x_t.shape[2]}; chunks go through the autoregressive call (is_ar=True)")
        if cache_tag != "pos":
            raise NotImplementedError("one cache tag ('pos') is implemented")
        if fwd_pred_type is None:
            fwd_pred_type = self.net_pred_type
        else:
            assert fwd_pred_type in NET_PRED_TYPES, f"{fwd_pred_type} is not supported as fwd_pred_type"
        if torch.is_grad_enabled() and (x_t.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("fastgen_amd.CausalWan: the backward pass is not implemented; call under torch.no_grad() / "
                                      "torch.inference_mode() (sampling)")
        if x_t.device.type != "cuda":
            raise RuntimeError("fastgen_amd runs on a HIP GPU only (no CPU path); got a tensor on " + str(x_t.device))
        if condition is None:
            raise ValueError("CausalWan.forward needs the text condition [B, L, text_dim]")
        condition = torch.stack(condition, dim=0) if isinstance(condition, list) else condition
        B, C, F, H, W = x_t.shape
        dev = x_t.device
        if C != self.in_channels:
            raise ValueError(f"x_t must have {self.in_channels} channels, got {C}")
        self._bind(dev)
        ws = self._workspace(dev, B, F, H, W)
        L = _lib.lib()
        self._set_text(condition, B, dev, ws)
        # per-frame timesteps in the embedder's units (`_compute_timestep_inputs`, :1063-1075: rescale_t, [B] -> [B, F])
        t_in = torch.atleast_1d(t.detach()).to(dev)
        ts = self.noise_scheduler.rescale_t(t_in)
        ts = (ts.view(-1, 1).expand(B, F) if ts.ndim == 1 else ts).to(torch.float32).contiguous()
        x32 = x_t.detach().to(torch.float32).contiguous()
        out = torch.empty_like(x32)
        if is_ar:
            _lib.check(L.fg_wan_forward(self._h, ctypes.c_void_p(x32.data_ptr()), ctypes.c_void_p(ts.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                        B, F, H, W, int(cur_start_frame), int(bool(store_kv)), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                        self._stream(dev)))
        else:  # teacher / diffusion forcing: all frames under the block-wise causal mask, caches untouched
            _lib.check(L.fg_wan_forward_block_causal(self._h, ctypes.c_void_p(x32.data_ptr()), ctypes.c_void_p(ts.data_ptr()),
                                                     ctypes.c_void_p(out.data_ptr()), B, F, H, W, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                                     self._stream(dev)))
        out = out.to(x_t.dtype)
        t_conv = t_in[:, None, :, None, None] if t_in.ndim == 2 else t_in
        return self.noise_scheduler.convert_model_output(x_t, out, t_conv, src_pred_type=self.net_pred_type, target_pred_type=fwd_pred_type)

    # ---- the chunk-by-chunk student loop as one library call ------------------------------------------------------------------
    @torch.no_grad()
    def student_sample(self, x: torch.Tensor, t_list, condition: Any, sample_type: str = "sde", context_noise: float = 0.0,
                       eps: Optional[torch.Tensor] = None, seed: Optional[int] = None, use_graph: bool = True, prefill_frames: int = 0,
                       exit_steps=None) -> torch.Tensor:
        """`CausVidModel._student_sample_loop` (fastgen/methods/distribution_matching/causvid.py:87-185) over `fg_wan_sampler_run`: x
        [B, C, F, H, W] latents (already scaled to t_list[0]) are overwritten chunk by chunk with the generated frames - per chunk N x {x0
        prediction over the cached frames + this chunk; re-noise to the next timestep}, then the cache-fill call.  prefill_frames: frames at
        the head that only fill the caches (`generator_fn_extrapolation`'s bridged segment head); exit_steps: one exit index per chunk
        (`SelfForcingModel.rollout_with_gradient`).  eps: noise videos to inject instead of device draws, [steps - 1 (+ 1 if context_noise
        > 0), B, C, F, H, W].  The caches are empty afterwards, as after the reference's loop."""
        if x.device.type != "cuda":
            raise RuntimeError("fastgen_amd runs on a HIP GPU only (no CPU path); got a tensor on " + str(x.device))
        if sample_type not in ("sde", "ode"):
            raise NotImplementedError(f"student_sample_type must be one of 'sde', 'ode' but got {sample_type}")
        if self.net_pred_type not in ("flow", "x0"):
            raise NotImplementedError(f"net_pred_type {self.net_pred_type!r} has no fused loop")
        if condition is None:
            raise ValueError("CausalWan needs the text condition [B, L, text_dim]")
        condition = torch.stack(condition, dim=0) if isinstance(condition, list) else condition
        B, C, F, H, W = x.shape
        dev = x.device
        if C != self.in_channels:
            raise ValueError(f"x must have {self.in_channels} channels, got {C}")
        tl = [float(v) for v in (t_list.tolist() if isinstance(t_list, torch.Tensor) else t_list)]
        steps = len(tl) - 1
        assert tl[-1] == 0, "t_list[-1] must be zero"
        self._bind(dev)
        L = _lib.lib()
        need = L.fg_wan_sampler_workspace_bytes(self._h, B, F, H, W)
        if need == 0:
            raise ValueError(f"bad video shape: batch {B}, frames {F}, {H}x{W} (height and width must be even)")
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        ws = self._ws
        self._set_text(condition, B, dev, ws)
        x32 = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.to(torch.float32).contiguous()
        n_eps = max(steps - 1, 0) + (1 if context_noise and context_noise > 0 else 0)
        if eps is not None:
            eps = eps.to(device=dev, dtype=torch.float32).contiguous()
            if eps.numel() != n_eps * x32.numel():
                raise ValueError(f"eps must hold {n_eps} noise videos shaped like x")
        if seed is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())
        sc = _lib.fg_wan_sampler_config()
        rf = self.schedule_type != "edm"
        sc.t_scale = float(self.noise_scheduler.num_steps) if rf else 1.0
        sc.context_noise = float(context_noise or 0.0)
        sc.net_pred_flow = int(self.net_pred_type == "flow")
        sc.schedule = _lib.FG_SCHEDULE_RF if rf else _lib.FG_SCHEDULE_EDM
        sc.prefill_frames = int(prefill_frames)
        ex = None
        if exit_steps is not None:
            ex = (ctypes.c_int * len(exit_steps))(*[int(e) for e in exit_steps])
            n_chunks = max(1, F // self.chunk_size)
            if len(exit_steps) < n_chunks:
                raise ValueError(f"exit_steps must hold one index per chunk ({n_chunks})")
        self._keep = (x32, eps)
        try:
            _lib.check(L.fg_wan_sampler_run(
                self._h, ctypes.byref(sc), ctypes.c_void_p(x32.data_ptr()), (ctypes.c_double * (steps + 1))(*tl), steps,
                _lib.FG_SAMPLE_SDE if sample_type == "sde" else _lib.FG_SAMPLE_ODE, ex,
                ctypes.c_void_p(eps.data_ptr() if eps is not None and eps.numel() else None), ctypes.c_uint64(seed), B, F, H, W,
                ctypes.c_void_p(ws.data_ptr()), ws.numel(), 1 if use_graph else 0, self._stream(dev)))
        finally:
            self._text_key = None  # the loop ends with clear_caches(): the text is forgotten with them
        if x32 is not x:
            x.copy_(x32)
        return x
