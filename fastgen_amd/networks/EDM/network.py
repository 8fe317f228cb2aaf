"""`EDMPrecond` drop-in for the reference's `fastgen.networks.EDM.network.EDMPrecond` (SongUNet / DDPM++ variant,
reference EDM/network.py:808-1026), backed by libfastgen_amd.so.

Select it by pointing a config's `net._target_` at `fastgen_amd.networks.EDM.network.EDMPrecond` (the reference's
`instantiate`, fastgen/utils/__init__.py:53-98, calls `cls(**kwargs)` with the kwargs of EDM_CIFAR10_Config,
fastgen/configs/net.py:29-48).  What is kept identical to the reference:
  * constructor kwargs, `forward(x_t, t, condition, r, return_features_early, feature_indices, return_logvar,
    fwd_pred_type)` semantics, `.noise_scheduler`, `.net_pred_type`, `.schedule_type`, `.label_dim`, `.sample()`,
    `.fully_shard()`, `.reset_parameters()`;
  * `state_dict()` — 429 entries with the reference's key names and OIHW shapes (Checkpointer.load uses
    strict=False, so a wrong name would be skipped silently, utils/checkpointer.py:155-161).
Parameters are ordinary fp32 `nn.Parameter`s; the library borrows their device pointers and keeps MFMA-order
copies of the conv weights that are rebuilt whenever a parameter's storage or version changes.

Both network flavours of the reference's EDM CIFAR-10 configs are covered: the DMD2 / consistency students
(x0 prediction, EDM schedule, full preconditioning) and the MeanFlow student (configs/experiments/EDM/
config_mf_cifar10.py: r_timestep=True, drop_precond='both', schedule_type='rf', net_pred_type='flow').

`feature_indices` / `return_features_early` (the encoder `block3` taps the DMD2 discriminator reads,
EDM/network.py:525-544) are served by the same engine, forward only.

Not provided on this path (raises, never falls back): non-SongUNet model types, and any device but a
HIP GPU.
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
from fastgen_amd.networks.network import FastGenNetwork
from fastgen_amd.networks.noise_schedule import NET_PRED_TYPES, expand_like


# How a call outside bf16 autocast computes (the reference's `precision="float32"`, configs/config.py:167-171, which on its CUDA
# path is TF32 arithmetic, utils/scripts.py:43-45).  "fp32": exact fp32 products on the fp32 matrix instructions (157 TFLOP/s roof);
# "bf16x3": fp32 tensors, convolutions as three bf16 MFMAs per product (hi/lo split, ~2^-17 per product: 64x tighter than TF32,
# 833 TFLOP/s roof; held to the SAME parity tolerance as "fp32" in tests/test_gpu_parity.py).  Overridable per module (compute_dtype=...) or per process (FASTGEN_AMD_COMPUTE_DTYPE).
DEFAULT_FP32_MODE = "bf16x3"


class _Node(nn.Module):
    """Bare container: the parameter tree only has to reproduce the reference's state-dict key paths."""


class _UNet(_Node):
    """The `model` node (SongUNet in the reference).  The fused engine call runs INSIDE this module's forward - `self.model(fn)`
    just evaluates `fn()` - so that hooks registered on it fire around the call exactly as they do around SongUNet.forward in the
    reference: FSDP2's, after `fully_shard` (pre-forward all-gather of the root group, pre-backward hook on the outputs, the
    root's post-backward callback that reduce-scatters every group's gradients)."""

    def forward(self, fn):
        return fn()


def _xavier_uniform(shape, fan_in, fan_out):
    return math.sqrt(6 / (fan_in + fan_out)) * (torch.rand(*shape) * 2 - 1)


def _init_value(name: str, shape) -> torch.Tensor:
    """Same distributions as the reference's constructors (EDM/network.py:22-31, 378-380, 487): xavier-uniform
    matrices, zero biases, unit norm gains; conv1 / proj / aux_conv scaled by 1e-5, qkv by sqrt(0.2)."""
    leaf = name.rsplit(".", 2)[-2]
    if name.endswith(".bias"):
        return torch.zeros(shape)
    if "norm" in leaf:
        return torch.ones(shape)
    fan_in = int(np.prod(shape[1:]))
    fan_out = shape[0] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[0]
    if leaf == "logvar_linear":
        return math.sqrt(1 / fan_in) * torch.randn(*shape)
    w = _xavier_uniform(shape, fan_in, fan_out)
    if leaf in ("conv1", "proj") or leaf.endswith("aux_conv"):
        w = w * 1e-5
    elif leaf == "qkv":
        w = w * math.sqrt(0.2)
    return w


class _TrainCall:
    """State of ONE differentiable forward call, shared by its three autograd nodes (see `_EDMForwardFn`)."""

    def __init__(self, net, taps, aug, drop, early, names):
        self.net, self.taps, self.aug, self.drop, self.early = net, taps, aug, drop, early
        self.training = net.training  # the backward differentiates the forward that ran, whatever mode the module is in by then
        # parameters by the part of the backward pass that finishes their gradients (fg_edm_backward_part)
        self.names = {
            _lib.FG_BWD_DECODER: tuple(n for n in names if n.startswith("model.dec.")),
            _lib.FG_BWD_ENCODER: tuple(n for n in names if n.startswith("model.enc.")),
            _lib.FG_BWD_EMBED: tuple(n for n in names if not n.startswith(("model.dec.", "model.enc."))),
        }
        self.dt = self.token = self.ws_ptr = self.ws_ptr_bwd = self.ntap = None
        self.saved = None          # (x32, t64, r64 | None, labels | None)
        self.want_dx = False
        # set by the first backward stage, read by the later ones
        self.d32 = self.dptrs = self.dx = self.scratch_out = None
        self.keep = []
        self.any_feat = False
        self.have_forward = 0

    def run_part(self, part: int, needs):
        """One third of the backward pass; returns the gradients of this part's parameters (None where not needed)."""
        net = self.net
        x32, t64, r64, labels = self.saved
        dev, B = x32.device, x32.shape[0]
        L = _lib.lib()
        with torch.no_grad():
            dt, h = net._engine(dev, self.dt)
        _lib.check(L.fg_edm_set_training(h, int(self.training)))
        named = net._named_weights(self.names[part])
        need_w = [p.requires_grad and nd for (_, p), nd in zip(named, needs)]
        # one zero-filled fp32 buffer, one view per trainable parameter (a fill per parameter costs 400+ launches)
        flat = torch.zeros(sum(p.numel() for (_, p), nw in zip(named, need_w) if nw), dtype=torch.float32, device=dev)
        grads, off = [], 0
        for (_, p), nw in zip(named, need_w):
            if nw:
                grads.append(flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
            else:
                grads.append(None)
        ws = net._train_workspace(h, B, dev)
        if part == _lib.FG_BWD_DECODER:
            self.have_forward = int(getattr(net, "_train_token", None) is self.token and ws.data_ptr() == self.ws_ptr)
            # from here on the workspace holds THIS call's backward state (its own kept forward, or the one the decoder part re-runs);
            # any training forward of the module in between takes the ownership away again (`_bwd_owner = None`)
            net._bwd_owner = self
            if not self.have_forward:
                net._train_token = object()  # the re-run forward overwrites whichever call's kept state the workspace held
        elif ws.data_ptr() != self.ws_ptr_bwd:
            raise RuntimeError("the training workspace was replaced between two parts of one backward pass")
        elif getattr(net, "_bwd_owner", None) is not self:
            raise RuntimeError("another differentiable forward (or backward) of this module ran between two parts of one backward pass: "
                               "the encoder / embedding gradients would be computed from its workspace state")
        self.ws_ptr_bwd = ws.data_ptr()
        if self.drop is not None:
            _lib.check(L.fg_edm_set_dropout(h, self.drop[0], self.drop[1]))
        p_ = lambda a: ctypes.c_void_p(a.data_ptr() if a is not None else None)  # noqa: E731
        try:
            for (n, _), g in zip(named, grads):
                if g is not None:
                    _lib.check(L.fg_edm_bind_grad(h, n.encode(), ctypes.c_void_p(g.data_ptr()), g.numel()))
            if self.aug is not None:
                _lib.check(L.fg_edm_set_augment(h, ctypes.c_void_p(self.aug.data_ptr())))
            _lib.check(L.fg_edm_backward_part(
                h, p_(x32), p_(t64), p_(r64), p_(labels), p_(self.d32), self.dptrs if (self.any_feat or self.early) else None,
                p_(self.scratch_out), p_(self.dx), self.have_forward, part, B, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                net._stream(dev)))
        finally:
            if self.drop is not None:
                _lib.check(L.fg_edm_set_dropout(h, 0.0, 0))
            if self.aug is not None:
                _lib.check(L.fg_edm_set_augment(h, None))
            for (n, _), g in zip(named, grads):
                if g is not None:
                    _lib.check(L.fg_edm_bind_grad(h, n.encode(), None, 0))
        return [g.to(p.dtype) if g is not None else None for (_, p), g in zip(named, grads)]


class _EDMForwardFn(torch.autograd.Function):
    """EDMPrecond.forward under autograd, as a chain of THREE nodes so that a data-parallel wrapper can overlap its gradient
    reduction with the rest of the backward pass (torch DDP reduces a bucket as soon as the hooks of all its parameters have
    fired, fastgen/utils/distributed/ddp.py:44-72; with one node all 418 hooks would fire together at the end):

        _EDMForwardFn (x_t, embedding-MLP parameters)  -> raw outputs, token      forward: fg_edm_forward_train (everything)
        _EDMEncoderFn (token, encoder parameters)      -> token                   forward: nothing
        _EDMDecoderFn (token, raw outputs, decoder parameters) -> outputs         forward: nothing

    Autograd runs them in reverse: _EDMDecoderFn.backward receives the output gradients and runs FG_BWD_DECODER (head + decoder:
    its parameters' hooks fire, their buckets start reducing), _EDMEncoderFn.backward runs FG_BWD_ENCODER, this node's backward
    FG_BWD_EMBED and hands back dL/dx_t.  The tokens carry no data (the state between the parts lives in the engine's workspace).
    Differentiable outputs: the prediction (unless the feature taps are returned early) and the requested taps; differentiable
    inputs: the parameters the call reads (so autograd accumulates into their .grad, as for the reference module) and x_t."""

    @staticmethod
    def forward(ctx, call, x32, t64, r64, labels, *weights):
        ctx.call = call
        # outputs nothing depends on (DMD2 detaches the teacher's prediction and keeps its taps) arrive in backward as None, not as
        # zero tensors: the backward then leaves the decoder alone
        ctx.set_materialize_grads(False)
        net, taps, drop, early = call.net, call.taps, call.drop, call.early
        dev, B = x32.device, x32.shape[0]
        L = _lib.lib()
        dt, h = net._engine(dev)
        call.dt = dt  # the backward differentiates in the mode the forward ran in, whatever autocast state it is called under
        if drop is not None:  # before the workspace is sized: one more tensor per block
            _lib.check(L.fg_edm_set_dropout(h, drop[0], drop[1]))
        ws = net._train_workspace(h, B, dev)
        out = None if early else torch.empty_like(x32)
        ntap = L.fg_edm_num_feature_taps(h)
        ptrs = (ctypes.c_void_p * max(ntap, 1))()
        feats = []
        ch, res = ctypes.c_int(), ctypes.c_int()
        for i in taps:
            _lib.check(L.fg_edm_feature_info(h, i, None, ctypes.byref(ch), ctypes.byref(res)))
            f = torch.empty(B, ch.value, res.value, res.value, dtype=torch.float32, device=dev)
            ptrs[i] = f.data_ptr()
            feats.append(f)
        with net._AugmentScope(h, call.aug):
            _lib.check(L.fg_edm_forward_train(
                h, ctypes.c_void_p(x32.data_ptr()), ctypes.c_void_p(t64.data_ptr()),
                ctypes.c_void_p(r64.data_ptr() if r64 is not None else None),
                ctypes.c_void_p(labels.data_ptr() if labels is not None else None),
                ctypes.c_void_p(out.data_ptr() if out is not None else None), ptrs if taps else None,
                B, ctypes.c_void_p(ws.data_ptr()), ws.numel(), net._stream(dev)))
        if drop is not None:
            _lib.check(L.fg_edm_set_dropout(h, 0.0, 0))
        # the training workspace now holds this call's state; any later training forward of the module replaces the token, and
        # the backward of this call then recomputes its forward
        call.token = net._train_token = object()
        net._bwd_owner = None
        call.ws_ptr, call.ntap = ws.data_ptr(), ntap
        call.saved = (x32, t64, r64, labels)  # plain inputs (no graph): held by the call object
        call.want_dx = bool(x32.requires_grad)
        token = torch.zeros(1, dtype=torch.float32, device=dev)
        return tuple(([] if early else [out]) + feats + [token])

    @staticmethod
    def backward(ctx, *douts):
        call = ctx.call
        grads = call.run_part(_lib.FG_BWD_EMBED, ctx.needs_input_grad[5:])
        return (None, call.dx if ctx.needs_input_grad[1] else None, None, None, None, *grads)


class _EDMEncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, call, token, *weights):
        ctx.call = call
        ctx.set_materialize_grads(False)
        return token.view_as(token)

    @staticmethod
    def backward(ctx, dtoken):
        grads = ctx.call.run_part(_lib.FG_BWD_ENCODER, ctx.needs_input_grad[2:])
        return (None, dtoken, *grads)


class _EDMDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, call, nraw, token, *rest):
        ctx.call, ctx.nraw = call, nraw
        ctx.set_materialize_grads(False)
        return tuple(r.view_as(r) for r in rest[:nraw])

    @staticmethod
    def backward(ctx, *douts):
        call = ctx.call
        net = call.net
        x32 = call.saved[0]
        # under FSDP2 the root group was all-gathered by its pre-backward hook; the blocks' groups are gathered here and stay
        # unsharded until the root's post-backward callback reduce-scatters their gradients and reshards them
        net._unshard_all()
        douts = list(douts)
        d_out = None if call.early else douts.pop(0)
        call.keep = []  # contiguous fp32 copies must outlive the three engine calls
        call.d32 = None
        if d_out is not None:
            call.d32 = d_out.detach().to(torch.float32).contiguous()
        call.dptrs = (ctypes.c_void_p * max(call.ntap, 1))()
        call.any_feat = False
        for i, g in zip(call.taps, douts):
            if g is not None:
                g32 = g.detach().to(torch.float32).contiguous()
                call.keep.append(g32)
                call.dptrs[i] = g32.data_ptr()
                call.any_feat = True
        if call.d32 is None and not call.early and not call.any_feat:
            call.d32 = torch.zeros_like(x32)  # nothing carries gradient (autograd does not normally call us then)
        # d32 None here: only the feature taps carry gradient (DMD2's GAN branch detaches the teacher's output, dmd2.py:137-146) -
        # the engine then differentiates the encoder alone, as after an early return
        call.dx = torch.empty_like(x32) if call.want_dx else None
        call.scratch_out = None if call.d32 is None else torch.empty_like(x32)
        grads = call.run_part(_lib.FG_BWD_DECODER, ctx.needs_input_grad[3 + ctx.nraw:])
        return (None, None, torch.zeros(1, dtype=torch.float32, device=x32.device), *([None] * ctx.nraw), *grads)


class EDMPrecond(FastGenNetwork):
    def __init__(
        self,
        img_resolution,
        img_channels,
        label_dim=0,
        sigma_data=0.5,
        sigma_shift=0.0,
        model_type="DhariwalUNet",
        drop_precond=None,
        net_pred_type="x0",
        schedule_type="edm",
        compute_dtype: Optional[str] = None,  # extension: "fp32" | "bf16x3" | "bf16" | None (= follow torch.autocast)
        **model_kwargs,
    ):
        super().__init__(net_pred_type=net_pred_type, schedule_type=schedule_type, **model_kwargs)
        if model_type != "SongUNet":
            raise ValueError(f"fastgen_amd implements model_type='SongUNet' only, got '{model_type}'")
        if drop_precond is not None and drop_precond not in ["input", "output", "both"]:
            raise ValueError(f"drop_precond must be one of 'input', 'output', 'both', or None, got {drop_precond}")
        if self.noise_scheduler.schedule_id < 0:
            raise NotImplementedError(f"schedule_type={schedule_type!r} is not implemented by the fused MI355X path")
        mk = dict(model_kwargs)
        unsupported = {
            "embedding_type": "positional", "encoder_type": "standard", "decoder_type": "standard",
        }
        self.label_dropout = float(mk.get("label_dropout", 0) or 0)  # class-label dropout in training mode (EDM/network.py:515-516)
        for k, want in unsupported.items():
            if mk.get(k, want) != want:
                raise NotImplementedError(f"{k}={mk[k]!r} is not implemented by the fused MI355X path (only {want!r})")
        if list(mk.get("resample_filter", [1, 1])) != [1, 1]:
            raise NotImplementedError("resample_filter other than [1, 1] is not implemented")
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.label_dim = label_dim
        self.sigma_data = sigma_data
        self.sigma_shift = sigma_shift
        self.drop_precond = drop_precond
        self.r_timestep = bool(mk.get("r_timestep", False))
        self.dropout = mk.get("dropout", 0.10)
        self.compute_dtype = compute_dtype or os.environ.get("FASTGEN_AMD_COMPUTE_DTYPE") or None
        if self.compute_dtype not in (None, "fp32", "bf16x3", "bf16"):
            raise ValueError(f"compute_dtype must be 'fp32', 'bf16x3', 'bf16' or None, got {self.compute_dtype!r}")

        cfg = _lib.fg_edm_config()
        cfg.img_resolution, cfg.img_channels, cfg.label_dim = img_resolution, img_channels, label_dim
        cfg.augment_dim = mk.get("augment_dim", 0)
        self.augment_dim = int(cfg.augment_dim)
        cfg.model_channels = mk.get("model_channels", 128)
        mult = list(mk.get("channel_mult", [1, 2, 2, 2]))
        attn = list(mk.get("attn_resolutions", [16]))
        if len(mult) > _lib.FG_MAX_LEVELS or len(attn) > _lib.FG_MAX_LEVELS:
            raise ValueError("too many resolution levels")
        cfg.num_levels = len(mult)
        for i, m in enumerate(mult):
            cfg.channel_mult[i] = m
        cfg.channel_mult_emb = mk.get("channel_mult_emb", 4)
        cfg.num_blocks = mk.get("num_blocks", 4)
        cfg.num_attn_resolutions = len(attn)
        for i, a in enumerate(attn):
            cfg.attn_resolutions[i] = a
        cfg.channel_mult_noise = mk.get("channel_mult_noise", 1)
        cfg.sigma_data, cfg.sigma_shift = float(sigma_data), float(sigma_shift)
        cfg.r_timestep = int(self.r_timestep)
        cfg.drop_precond = {None: 0, "input": _lib.FG_DROP_PRECOND_INPUT, "output": _lib.FG_DROP_PRECOND_OUTPUT,
                            "both": _lib.FG_DROP_PRECOND_INPUT | _lib.FG_DROP_PRECOND_OUTPUT}[drop_precond]
        cfg.schedule = self.noise_scheduler.schedule_id
        self._cfg = cfg
        self._engines: Dict[int, ctypes.c_void_p] = {}
        self._bound_sig: Dict[int, Any] = {}
        self._pack_refs: Dict[int, list] = {}
        self._ws: Dict[int, torch.Tensor] = {}
        self._noise_channels = cfg.model_channels * cfg.channel_mult_noise

        # parameter tree with the reference's key paths; names/shapes come from the library's own plan
        self.model = _UNet()
        self._param_names: List[str] = []
        h = self._make_engine(_lib.FG_DTYPE_F32)
        L = _lib.lib()
        name, ndim, shape = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int64 * 4)()
        for i in range(L.fg_edm_num_params(h)):
            _lib.check(L.fg_edm_param_info(h, i, ctypes.byref(name), ctypes.byref(ndim), shape))
            full = name.value.decode()
            shp = tuple(shape[j] for j in range(ndim.value))
            self._param_names.append(full)
            node, parts = self, full.split(".")
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            node.register_parameter(parts[-1], nn.Parameter(_init_value(full, shp)))
            # the reference keeps the constant 2x2 resampling kernel as a persistent buffer (EDM/network.py:89-91)
            if parts[-1] == "weight" and parts[-2] in ("conv0", "skip") and (parts[-3].endswith("_down") or parts[-3].endswith("_up")):
                node.register_buffer("resample_filter", torch.full((1, 1, 2, 2), 0.25))
        self._engines[_lib.FG_DTYPE_F32] = h
        # (leaf module, parameter name) of every engine parameter, in engine order (see _engine)
        self._leafs = []
        for full in self._param_names:
            node, parts = self, full.split(".")
            for p_ in parts[:-1]:
                node = node._modules[p_]
            self._leafs.append((node, parts[-1]))

    # ------------------------------------------------------------------------------------------------
    def _make_engine(self, dtype: int):
        cfg = _lib.fg_edm_config.from_buffer_copy(self._cfg)
        cfg.compute_dtype = dtype
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().fg_edm_create(ctypes.byref(cfg), ctypes.byref(h)))
        return h

    def __del__(self):
        try:
            for h in getattr(self, "_engines", {}).values():
                _lib.lib().fg_edm_destroy(h)
        except Exception:
            pass

    def _named_weights(self, names=None):
        sd = dict(self.named_parameters())
        return [(n, sd[n]) for n in (self._param_names if names is None else names)]

    def _diff_names(self, with_augment: bool):
        """Parameters the engine's forward reads, i.e. the inputs of the autograd Function.  The rest stays out of the graph and
        keeps `.grad = None`, as in the reference (which trains with ddp_find_unused_parameters=True for exactly these,
        configs/config.py:158-161): `logvar_linear` (differentiated by ordinary autograd when return_logvar is used) and
        `map_augment` when the call carries no augmentation labels - so AdamW neither decays them nor advances their state."""
        return tuple(n for n in self._param_names
                     if not n.startswith("model.logvar_linear") and (with_augment or n != "model.map_augment.weight"))

    def _select_dtype(self) -> int:
        if self.compute_dtype is not None:
            return _lib.DTYPE_NAMES[self.compute_dtype]
        if torch.is_autocast_enabled():
            ad = torch.get_autocast_gpu_dtype()
            if ad == torch.bfloat16:
                return _lib.FG_DTYPE_BF16
            if ad != torch.float32:
                raise NotImplementedError(f"autocast dtype {ad} is not implemented (bf16 or fp32)")
        return _lib.DTYPE_NAMES[DEFAULT_FP32_MODE]

    def _engine(self, device: torch.device, dt: Optional[int] = None):
        """Engine for the active (or the given) compute dtype with up-to-date weights bound and packed."""
        if dt is None:
            dt = self._select_dtype()
        if dt not in self._engines:
            self._engines[dt] = self._make_engine(dt)
        h = self._engines[dt]
        # Is what the engine bound still what the module holds?  One pass over the 421 parameters (storage pointer, in-place
        # version counter, dtype: any optimizer step, load_state_dict, .to() or FSDP2 all-gather changes one of them), read straight
        # from the leaf modules' parameter dicts - no named_parameters() generator, no per-call name -> tensor dict.
        ps = [leaf._parameters[nm] for leaf, nm in self._leafs]
        for n, p in zip(self._param_names, ps):
            if hasattr(p, "_local_tensor"):
                raise RuntimeError(
                    f"parameter {n} is a sharded DTensor: the engine reads whole parameters.  Call the network through "
                    "forward() / jvp() / few_step_sample() / generator_fn(), which all-gather FSDP2 groups around the call")
        sig = tuple((p.data_ptr(), p._version, p.dtype) for p in ps)
        if self._bound_sig.get(dt) != sig:
            weights = list(zip(self._param_names, ps))
            L = _lib.lib()
            refs = []
            for n, p in weights:
                if p.device.type != "cuda":
                    raise RuntimeError(f"parameter {n} is on {p.device}; fastgen_amd runs on a HIP GPU only (no CPU path)")
                q = p.detach()
                if q.dtype != torch.float32 or not q.is_contiguous():
                    q = q.to(torch.float32).contiguous()
                refs.append(q)
                _lib.check(L.fg_edm_bind_param(h, n.encode(), ctypes.c_void_p(q.data_ptr()), q.numel()))
            _lib.check(L.fg_edm_pack_weights(h, ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
            self._pack_refs[dt] = refs
            self._bound_sig[dt] = sig
        # train() / eval() as far as the arithmetic sees it: sigma_shift applies in eval mode only (EDM/network.py:956)
        _lib.check(_lib.lib().fg_edm_set_training(h, int(self.training)))
        return dt, h

    def _workspace(self, dt: int, h, batch: int, device) -> torch.Tensor:
        need = _lib.lib().fg_edm_workspace_bytes(h, batch)
        ws = self._ws.get(dt)
        if ws is None or ws.numel() < need or ws.device != device:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._ws[dt] = ws
        return ws

    def _train_workspace(self, h, batch: int, device) -> torch.Tensor:
        need = _lib.lib().fg_edm_backward_workspace_bytes(h, batch)
        ws = self._ws.get("bwd")
        if ws is None or ws.numel() < need or ws.device != device:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._ws["bwd"] = ws
        return ws

    @staticmethod
    def _stream(device) -> ctypes.c_void_p:
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def _needs_grad(self, x_t: Optional[torch.Tensor] = None) -> bool:
        return torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                            or (x_t is not None and x_t.requires_grad))

    def _check_trainable_call(self, return_logvar=False):
        """Autograd through the module (fg_edm_backward_ex): gradients of the prediction and of the feature taps with respect to
        the parameters and to x_t - what the DMD2 student / fake-score updates and its GAN branch need (dmd2.py) - in the
        split-bf16 mode (fp32 tensors, the default outside autocast: the reference's `precision="float32"` training,
        configs/config.py:167-169) and in the bf16 mode (under bf16 autocast).  The exact-fp32 mode has no backward."""
        if self._select_dtype() == _lib.FG_DTYPE_F32:
            raise NotImplementedError(
                "fastgen_amd.EDMPrecond: the backward pass runs in the 'bf16x3' (default) and 'bf16' compute modes, not in the "
                "exact-fp32 mode (compute_dtype='fp32'); use torch.no_grad() for inference in that mode")
        # return_logvar needs no special handling: logvar_linear(posemb(c_noise)) is a [B, 128] x [128, 1] product evaluated with
        # torch on the module's own parameters (host-side plumbing) and differentiates through ordinary autograd

    def _labels(self, condition, batch: int, device) -> Optional[torch.Tensor]:
        if isinstance(condition, dict) and "aug_condition" in condition:
            condition = condition.get("orig_condition")  # the augmentation part is taken by _augment()
        if self.label_dim == 0 or condition is None:
            return None  # the library broadcasts map_label(zeros) as the reference does (EDM/network.py:919-925)
        c = condition.reshape(-1, self.label_dim).to(device=device, dtype=torch.float32)
        if c.shape[0] == 1 and batch > 1:
            c = c.expand(batch, -1)
        if c.shape[0] != batch:
            raise ValueError(f"condition has {c.shape[0]} rows, expected {batch}")
        if self.training and self.label_dropout:
            # whole label rows are zeroed with probability label_dropout (classifier-free-guidance training), same draw as the
            # reference: torch.rand([B, 1]) on the input's device (EDM/network.py:515-516)
            c = c * (torch.rand([batch, 1], device=device) >= self.label_dropout).to(c.dtype)
        return c.contiguous()

    def _augment(self, condition, batch: int, device) -> Optional[torch.Tensor]:
        """Augmentation labels of a {"aug_condition", "orig_condition"} condition (EDM/network.py:903-915): [B, augment_dim] fp32,
        or None; a width that does not match map_augment is ignored, as the reference does (with a warning there)."""
        if not (isinstance(condition, dict) and "aug_condition" in condition):
            return None
        aug = condition["aug_condition"]
        if aug is None or self.augment_dim == 0 or aug.shape[-1] != self.augment_dim:
            return None
        a = aug.detach().reshape(-1, self.augment_dim).to(device=device, dtype=torch.float32)
        if a.shape[0] != batch:
            raise ValueError(f"aug_condition has {a.shape[0]} rows, expected {batch}")
        return a.contiguous()

    class _AugmentScope:
        """fg_edm_set_augment(h, aug) for the duration of one library call."""

        def __init__(self, h, aug):
            self.h, self.aug = h, aug

        def __enter__(self):
            if self.aug is not None:
                _lib.check(_lib.lib().fg_edm_set_augment(self.h, ctypes.c_void_p(self.aug.data_ptr())))

        def __exit__(self, *exc):
            if self.aug is not None:
                _lib.check(_lib.lib().fg_edm_set_augment(self.h, None))
            return False

    # ------------------------------------------------------------------------------------------------
    def reset_parameters(self):
        with torch.no_grad():
            for n, p in self._named_weights():
                p.copy_(_init_value(n, tuple(p.shape)).to(p.dtype))
        super().reset_parameters()

    @torch.no_grad()
    def randomize_parameters_(self, seed: int = 0):
        """Benchmark / test initialisation: matrices ~ N(0, 1/fan_in), norm gains ~ 1 + 0.1 N(0,1), biases ~ 0.1 N(0,1).
        Unlike the reference's default init (conv1 / proj / aux_conv scaled by 1e-5) every branch of every block
        carries O(1) signal, so timings and parity checks see non-trivial data."""
        g = torch.Generator().manual_seed(seed)
        for n, p in self._named_weights():
            if p.dim() == 1:
                base = 1.0 if ("norm" in n and n.endswith("weight")) else 0.0
                v = base + 0.1 * torch.randn(p.shape, generator=g)
            else:
                v = torch.randn(p.shape, generator=g) / math.sqrt(int(np.prod(p.shape[1:])))
            p.copy_(v.to(device=p.device, dtype=p.dtype))
        return self

    def fully_shard(self, **kwargs):
        """FSDP2 with the reference's wrapping granularity (EDM/network.py:861-879): one parameter group per UNetBlock, then the
        U-Net as the root group (embedding MLP, stem, head).  Between calls every parameter is a sharded DTensor, as in the
        reference; the state dict, optimizer and checkpointing see exactly what they see there.

        How the fused engine runs under it (`_fsdp_call`): the call is made inside `self.model`'s forward, so FSDP2's own hooks
        all-gather / reshard the root group, hook the outputs for the backward and queue the root's post-backward callback; the
        blocks' groups are all-gathered explicitly (`FSDPModule.unshard`, asynchronously, all in flight together) before the
        engine binds the whole parameters, and resharded after it.  In the backward the same happens in reverse: the autograd
        Function all-gathers the blocks, the engine writes whole gradients, autograd accumulates them into the unsharded
        parameters' .grad, and FSDP2's post-backward callback reduce-scatters every group (reduce dtype fp32, as configured by
        utils/distributed/fsdp.py:116-122) and reshards."""
        from torch.distributed.fsdp import fully_shard

        for group in (self.model._modules["enc"], self.model._modules["dec"]):
            for _, block in group._modules.items():
                if "conv0" in block._modules:
                    fully_shard(block, **kwargs)
        fully_shard(self.model, **kwargs)

    def _fsdp_modules(self):
        """FSDP2-managed modules of the tree, root (`self.model`) first; [] when `fully_shard` was not applied."""
        try:
            from torch.distributed.fsdp import FSDPModule
        except ImportError:  # pragma: no cover
            return []
        return [m for m in self.model.modules() if isinstance(m, FSDPModule)]

    def _unshard_all(self, skip_root: bool = False):
        mods = self._fsdp_modules()
        if skip_root:
            mods = [m for m in mods if m is not self.model]
        handles = [m.unshard(async_op=True) for m in mods]  # every all-gather in flight before the first wait
        for hd in handles:
            if hd is not None:
                hd.wait()
        return mods

    def _fsdp_call(self, fn):
        """Run `fn` (binds parameters, calls the engine) with whole parameters.  Plain module: just `fn()`."""
        if not self._fsdp_modules():
            return fn()

        def run():
            kids = self._unshard_all(skip_root=True)  # the root group is gathered by FSDP2's pre-forward hook of self.model
            try:
                return fn()
            finally:
                for m in kids:
                    m.reshard()

        res = self.model(run)
        if not torch.is_grad_enabled():
            # FSDP2 leaves the ROOT group gathered after a forward (a backward would need it next).  Without autograd there is no
            # backward: reshard, so that inference leaves every parameter sharded (memory released, and an optimizer built
            # afterwards sees the sharded parameters)
            self.model.reshard()
        return res

    # ------------------------------------------------------------------------------------------------
    def forward(
        self,
        x_t: torch.Tensor,
        t: torch.Tensor,
        condition: Optional[torch.Tensor] = None,
        r: Optional[torch.Tensor] = None,
        return_features_early: bool = False,
        feature_indices: Optional[Set[int]] = None,
        return_logvar: bool = False,
        fwd_pred_type: Optional[str] = None,
        **fwd_kwargs,
    ):
        if feature_indices is None:
            feature_indices = {}
        if return_features_early and len(feature_indices) == 0:
            return []
        if fwd_pred_type is None:
            fwd_pred_type = self.net_pred_type
        else:
            assert fwd_pred_type in NET_PRED_TYPES, f"{fwd_pred_type} is not supported as fwd_pred_type"
        if r is not None and not self.r_timestep:
            raise ValueError("r_noise_labels provided, but r_timestep is not set")
        if r is None and self.r_timestep:
            raise ValueError("this network was built with r_timestep=True: forward() needs r")
        if fwd_kwargs:
            raise TypeError(f"unexpected forward kwargs: {sorted(fwd_kwargs)}")
        # training mode with dropout (EDM/network.py:283-284): a fresh mask per call, regenerated by the backward from the seed
        drop = None
        if self.training and self.dropout:
            drop = (float(self.dropout), int(torch.randint(0, 2**62, (1,)).item()))
        needs_grad = self._needs_grad(x_t)
        if needs_grad or (self.training and self.dropout):
            self._check_trainable_call(return_logvar)
        if x_t.device.type != "cuda":
            raise RuntimeError("fastgen_amd runs on a HIP GPU only (no CPU path); got a tensor on " + str(x_t.device))
        if x_t.dim() != 4 or x_t.shape[1] != self.img_channels or x_t.shape[2] != self.img_resolution or x_t.shape[3] != self.img_resolution:
            raise ValueError(f"x_t must be [B,{self.img_channels},{self.img_resolution},{self.img_resolution}], got {tuple(x_t.shape)}")
        B, dev = x_t.shape[0], x_t.device
        x32 = (x_t if needs_grad and x_t.requires_grad else x_t.detach()).to(torch.float32).contiguous()
        t64 = torch.atleast_1d(t.detach()).to(device=dev, dtype=torch.float64)
        if t64.numel() == 1 and B > 1:
            t64 = t64.expand(B)
        t64 = t64.contiguous()
        if t64.numel() != B:
            raise ValueError(f"t has {t64.numel()} entries, expected {B}")
        r64 = None
        if r is not None:
            r64 = torch.atleast_1d(r.detach()).to(device=dev, dtype=torch.float64)
            if r64.numel() == 1 and B > 1:
                r64 = r64.expand(B)
            r64 = r64.contiguous()
            if r64.numel() != B:
                raise ValueError(f"r has {r64.numel()} entries, expected {B}")
        labels = self._labels(condition, B, dev)
        aug = self._augment(condition, B, dev)

        def engine_call():
            """Everything that needs whole parameters: bind / pack, the engine call (or the autograd Function around it)."""
            dt, h = self._engine(dev)
            ws = self._workspace(dt, h, B, dev)
            L = _lib.lib()
            features: List[torch.Tensor] = []
            if needs_grad or drop is not None:  # the training entry points (under no_grad the Function simply records nothing)
                ntap = L.fg_edm_num_feature_taps(h)
                taps = tuple(i for i in range(ntap) if i in feature_indices)
                if return_features_early:
                    assert len(taps) == len(feature_indices), f"{len(taps)} != {len(feature_indices)}"
                call = _TrainCall(self, taps, aug, drop, bool(return_features_early), self._diff_names(aug is not None))
                W = {k: [p_ for _, p_ in self._named_weights(v)] for k, v in call.names.items()}
                raw = _EDMForwardFn.apply(call, x32, t64, r64, labels, *W[_lib.FG_BWD_EMBED])
                token = _EDMEncoderFn.apply(call, raw[-1], *W[_lib.FG_BWD_ENCODER])
                res = _EDMDecoderFn.apply(call, len(raw) - 1, token, *raw[:-1], *W[_lib.FG_BWD_DECODER])
                res = list(res)
                out = None if return_features_early else res.pop(0)
                features = [f.to(x_t.dtype) for f in res]
                if return_features_early:
                    return None, features
            elif len(feature_indices):
                # tap i = the i-th encoder block named *block3* (EDM/network.py:535-539); indices beyond the taps are ignored by
                # the reference's loop and trip its length assert only when returning early (:543)
                ntap = L.fg_edm_num_feature_taps(h)
                ptrs = (ctypes.c_void_p * max(ntap, 1))()
                ch, res = ctypes.c_int(), ctypes.c_int()
                for i in range(ntap):
                    if i in feature_indices:
                        _lib.check(L.fg_edm_feature_info(h, i, None, ctypes.byref(ch), ctypes.byref(res)))
                        f = torch.empty(B, ch.value, res.value, res.value, dtype=torch.float32, device=dev)
                        ptrs[i] = f.data_ptr()
                        features.append(f)
                if return_features_early:
                    assert len(features) == len(feature_indices), f"{len(features)} != {len(feature_indices)}"
                out = None if return_features_early else torch.empty_like(x32)
                with self._AugmentScope(h, aug):
                    _lib.check(L.fg_edm_forward_features(
                    h, ctypes.c_void_p(x32.data_ptr()), ctypes.c_void_p(t64.data_ptr()),
                    ctypes.c_void_p(r64.data_ptr() if r64 is not None else None),
                    ctypes.c_void_p(labels.data_ptr() if labels is not None else None),
                    ctypes.c_void_p(out.data_ptr() if out is not None else None), ptrs, B, ctypes.c_void_p(ws.data_ptr()),
                    ws.numel(), self._stream(dev)))
                features = [f.to(x_t.dtype) for f in features]
                if return_features_early:
                    return None, features
            else:
                out = torch.empty_like(x32)
                with self._AugmentScope(h, aug):
                    _lib.check(L.fg_edm_forward(
                        h, ctypes.c_void_p(x32.data_ptr()), ctypes.c_void_p(t64.data_ptr()),
                        ctypes.c_void_p(r64.data_ptr() if r64 is not None else None),
                        ctypes.c_void_p(labels.data_ptr() if labels is not None else None), ctypes.c_void_p(out.data_ptr()), None,
                        B, ctypes.c_void_p(ws.data_ptr()), ws.numel(), self._stream(dev)))
            return out, features

        out, features = self._fsdp_call(engine_call)
        if return_features_early:
            return features
        out = out.to(x_t.dtype)
        out = self.noise_scheduler.convert_model_output(x_t, out, t64, src_pred_type=self.net_pred_type,
                                                        target_pred_type=fwd_pred_type)
        if len(feature_indices):
            out = [out, features]
        if return_logvar:
            return out, self._logvar(t64)
        return out

    @torch.no_grad()
    def jvp(self, x_t: torch.Tensor, t: torch.Tensor, v_x: torch.Tensor, v_t: Optional[torch.Tensor] = None,
            condition: Optional[torch.Tensor] = None, r: Optional[torch.Tensor] = None, v_r: Optional[torch.Tensor] = None,
            fwd_pred_type: Optional[str] = None):
        """(output, directional derivative) of `forward(x_t, t, condition=condition, r=r)` along (v_x, v_t, v_r) - what
        `torch.func.jvp(net_wrapper, (x_t, t, r), tangents)` returns in MeanFlowModel._jvp / sCM (mean_flow.py:240-250, sCM.py:179),
        as one library call (fg_edm_jvp, bf16 compute mode).  No graph is built (the reference detaches the result too)."""
        if fwd_pred_type is not None and fwd_pred_type != self.net_pred_type:
            raise NotImplementedError("jvp is provided for the network's own prediction type")
        if r is None and self.r_timestep:
            raise ValueError("this network was built with r_timestep=True: jvp() needs r")
        B, dev = x_t.shape[0], x_t.device
        return self._fsdp_call(lambda: self._jvp_call(x_t, t, v_x, v_t, condition, r, v_r, B, dev))

    def _jvp_call(self, x_t, t, v_x, v_t, condition, r, v_r, B, dev):
        dt, h = self._engine(dev)
        if dt == _lib.FG_DTYPE_F32:
            raise NotImplementedError("fastgen_amd.EDMPrecond.jvp runs in the 'bf16x3' and 'bf16' compute modes, not in exact fp32")
        f32 = lambda a: None if a is None else torch.atleast_1d(a.detach()).to(device=dev, dtype=torch.float32).expand(B).contiguous()
        f64 = lambda a: None if a is None else torch.atleast_1d(a.detach()).to(device=dev, dtype=torch.float64).expand(B).contiguous()
        x32 = x_t.detach().to(torch.float32).contiguous()
        vx = v_x.detach().to(torch.float32).contiguous()
        t64, r64, vt, vr = f64(t), f64(r), f32(v_t), f32(v_r)
        labels = self._labels(condition, B, dev)
        aug = self._augment(condition, B, dev)
        L = _lib.lib()
        drop = self.training and bool(self.dropout)
        if drop:  # train() mode: a fresh mask, as a training forward would draw (before the workspace is sized)
            _lib.check(L.fg_edm_set_dropout(h, float(self.dropout), int(torch.randint(0, 2**62, (1,)).item())))
        try:
            ws = self._train_workspace(h, B, dev)
            self._train_token = object()  # the kept state of an earlier training forward is overwritten
            self._bwd_owner = None
            out, jv = torch.empty_like(x32), torch.empty_like(x32)
            p = lambda a: ctypes.c_void_p(a.data_ptr() if a is not None else None)
            with self._AugmentScope(h, aug):
                _lib.check(L.fg_edm_jvp(h, p(x32), p(t64), p(r64), p(labels), p(vx), p(vt), p(vr), p(out), p(jv), B,
                                        ctypes.c_void_p(ws.data_ptr()), ws.numel(), self._stream(dev)))
        finally:
            if drop:
                _lib.check(L.fg_edm_set_dropout(h, 0.0, 0))
        return out.to(x_t.dtype), jv.to(x_t.dtype)

    def _logvar(self, t64: torch.Tensor) -> torch.Tensor:
        """logvar_linear(PositionalEmbedding(c_noise)) — the un-flipped [cos|sin] embedding (EDM/network.py:501,571)."""
        if self.drop_precond in ("input", "both"):
            c_noise = t64.to(torch.float32)
        else:
            c_noise = (t64.clamp(min=self.noise_scheduler.clamp_min).log() / 4).to(torch.float32)
        half = self._noise_channels // 2
        freqs = torch.arange(half, dtype=torch.float32, device=t64.device) / (half - 1)
        freqs = (1 / 10000) ** freqs
        ang = c_noise.ger(freqs)
        emb = torch.cat([ang.cos(), ang.sin()], dim=1)
        lv = self.model._modules["logvar_linear"]
        # under FSDP2 the two tensors are sharded DTensors between calls: full_tensor() all-gathers (and reduce-scatters the gradient)
        w, b = (p.full_tensor() if hasattr(p, "full_tensor") else p for p in (lv.weight, lv.bias))
        return emb @ w.to(emb.dtype).t() + b.to(emb.dtype)

    # ------------------------------------------------------------------------------------------------
    def fused_loop(self) -> Optional[str]:
        """Which student sampling loop fg_sampler_run can run for this network: 'x0' (FastGenModel._student_sample_loop,
        needs an x0-predicting network without r), 'meanflow' (MeanFlowModel._student_sample_loop, needs a
        flow-predicting r_timestep network), or None (callers take the generic per-step loop)."""
        if not self.r_timestep and self.net_pred_type == "x0":
            return "x0"
        if self.r_timestep and self.net_pred_type == "flow":
            return "meanflow"
        return None

    def supports_fused_loop(self, kind: str) -> bool:
        return kind is not None and self.fused_loop() == kind

    def few_step_sample(self, noise: torch.Tensor, condition: Optional[torch.Tensor], t_list, sample_type: str = "sde",
                        eps: Optional[torch.Tensor] = None, seed: Optional[int] = None, use_graph: bool = True,
                        out: Optional[torch.Tensor] = None, loop: Optional[str] = None) -> torch.Tensor:
        """The whole student sampling loop (methods/model.py:374-420 or consistency_model/mean_flow.py:336-381) as ONE
        library call / one hipGraph replay.

        t_list: steps+1 decreasing timesteps ending in 0.  sample_type 'sde' re-noises with `eps`
        ([steps-1,B,C,H,W], injected) or, if eps is None, with normals drawn on the device from `seed`;
        'ode' re-uses the implied noise (x0 loop) / integrates the average velocity (MeanFlow loop).
        loop: 'x0' | 'meanflow' (default: `fused_loop()`)."""
        if self._needs_grad():
            raise NotImplementedError(
                "the fused sampler builds no autograd graph; call it under torch.no_grad() / torch.inference_mode() as "
                "generator_fn does (fastgen/methods/model.py:405)")
        loop = loop or self.fused_loop()
        if loop is None or loop != self.fused_loop():
            raise NotImplementedError(
                f"the fused sampler has no loop {loop!r} for net_pred_type={self.net_pred_type!r}, r_timestep={self.r_timestep}")
        if noise.device.type != "cuda":
            raise RuntimeError("fastgen_amd runs on a HIP GPU only (no CPU path); got a tensor on " + str(noise.device))
        if sample_type not in ("sde", "ode"):
            raise NotImplementedError(f"student_sample_type must be one of 'sde', 'ode' but got {sample_type}")
        B, dev = noise.shape[0], noise.device
        tl = [float(v) for v in (t_list.tolist() if isinstance(t_list, torch.Tensor) else t_list)]
        steps = len(tl) - 1
        assert tl[-1] == 0, "t_list[-1] must be zero"
        n32 = noise if (noise.dtype == torch.float32 and noise.is_contiguous()) else noise.to(torch.float32).contiguous()
        labels = self._labels(condition, B, dev)
        if eps is not None:
            eps = eps.to(device=dev, dtype=torch.float32).contiguous()
            if eps.numel() != max(steps - 1, 0) * n32.numel():
                raise ValueError(f"eps must hold steps-1 = {steps - 1} noise tensors shaped like `noise`")
        if seed is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())  # host RNG: follows torch.manual_seed / set_random_seed
        if out is None:
            out = torch.empty_like(n32)
        tl_arr = (ctypes.c_double * (steps + 1))(*tl)
        self._keep = (n32, labels, eps)  # graph replays read these buffers; keep them alive
        self._fsdp_call(lambda: self._sampler_call(n32, labels, tl_arr, steps, sample_type, loop, eps, seed, out, B, dev, use_graph))
        return out

    def _sampler_call(self, n32, labels, tl_arr, steps, sample_type, loop, eps, seed, out, B, dev, use_graph):
        dt, h = self._engine(dev)
        ws = self._workspace(dt, h, B, dev)
        _lib.check(_lib.lib().fg_sampler_run(
            h, ctypes.c_void_p(n32.data_ptr()), ctypes.c_void_p(labels.data_ptr() if labels is not None else None),
            tl_arr, steps, _lib.FG_SAMPLE_SDE if sample_type == "sde" else _lib.FG_SAMPLE_ODE,
            _lib.FG_LOOP_MEANFLOW if loop == "meanflow" else _lib.FG_LOOP_X0,
            ctypes.c_void_p(eps.data_ptr() if eps is not None and eps.numel() else None), ctypes.c_uint64(seed),
            ctypes.c_void_p(out.data_ptr()), B, ctypes.c_void_p(ws.data_ptr()), ws.numel(), 1 if use_graph else 0,
            self._stream(dev)))
        return out

    # ------------------------------------------------------------------------------------------------
    def sample(self, noise: torch.Tensor, condition: Optional[torch.Tensor] = None,
               neg_condition: Optional[torch.Tensor] = None, guidance_scale: Optional[float] = 5.0, num_steps: int = 50,
               **kwargs) -> torch.Tensor:
        """Deterministic Euler sampler of the (teacher) EDM network with optional classifier-free guidance
        (EDM/network.py:976-1026)."""
        assert self.schedule_type == "edm", f"{self.schedule_type} is not supported"
        sigmas = self.noise_scheduler.get_t_list(num_steps, device=noise.device)
        x = self.noise_scheduler.latents(noise=noise, t_init=sigmas[0])
        for sigma, sigma_next in zip(sigmas[:-1], sigmas[1:]):
            t = sigma.expand(x.shape[0])
            if guidance_scale is not None and guidance_scale > 1.0 and neg_condition is not None:
                x0 = self(torch.cat([x, x], 0), torch.cat([t, t], 0), condition=torch.cat([neg_condition, condition], 0),
                          fwd_pred_type="x0")
                x0_uncond, x0_cond = x0.chunk(2)
                x0 = x0_uncond + guidance_scale * (x0_cond - x0_uncond)
            else:
                x0 = self(x, t, condition=condition, fwd_pred_type="x0")
            d = (x - x0) / expand_like(t, x)
            x = x + (sigma_next - sigma).to(x.dtype) * d
        return x
