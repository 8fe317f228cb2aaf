"""Student sampling entry points with the reference's signatures: `FastGenModel.generator_fn` and
`FastGenModel._student_sample_loop` (fastgen/methods/model.py:315-420).

For a `fastgen_amd` EDM or DiT network the whole loop is one library call (`fg_sampler_run` / `fg_dit_sampler_run`: latents ->
N x {network forward, re-noise} as a replayed hipGraph, host syncs of the reference loop hoisted to host scalars).  Any other
`FastGenNetwork`-shaped module takes the generic per-step loop, which is what the reference itself runs.
"""
from __future__ import annotations

import contextlib
from typing import Any, List, Optional

import torch

from fastgen_amd.networks.EDM.network import EDMPrecond


@contextlib.contextmanager
def inference_mode(*modules, precision_amp: Optional[torch.dtype] = None, device_type: str = "cuda"):
    """eval() + torch.inference_mode() (+ autocast), restoring .training on exit (utils/basic_utils.py:89-125)."""
    mods = [m for m in modules if isinstance(m, torch.nn.Module)]
    prev = [m.training for m in mods]
    try:
        for m in mods:
            m.eval()
        with torch.inference_mode(), torch.autocast(dtype=precision_amp, device_type=device_type,
                                                    enabled=precision_amp is not None):
            yield
    finally:
        for m, was in zip(mods, prev):
            m.train(was)


_FUSED_LOOPS = set()  # the `_student_sample_loop` implementations the library's loops restate (a subclass that overrides it opts out)


class FastGenModel:
    """Only the sampling classmethods of the reference class; they are what `scripts/inference/*`, `scripts/fid/*` and
    the wandb callback call."""

    # which fg_sampler_run loop restates this class's _student_sample_loop (see EDMPrecond.fused_loop)
    _fused_loop = "x0"

    @classmethod
    def _student_sample_loop(cls, net, x: torch.Tensor, t_list: torch.Tensor, condition: Any = None,
                             student_sample_type: str = "sde", **kwargs) -> torch.Tensor:
        """Generic per-step loop (methods/model.py:315-372) for networks without a fused sampler: predict x0 at t_i, and unless
        t_{i+1} is 0 re-noise it to t_{i+1} — with fresh noise ('sde') or with the noise implied by (x, x0) ('ode').  Networks
        with a `preserve_conditioning` hook get it applied to every prediction and every re-noised state."""
        n = x.shape[0]
        keep = getattr(net, "preserve_conditioning", None)
        sched = net.noise_scheduler
        x0 = x
        for i in range(len(t_list) - 1):
            t = t_list[i].expand(n)
            x0 = net(x, t, condition=condition, fwd_pred_type="x0")
            if keep is not None:
                x0 = keep(x0, condition)
            if t_list[i + 1] > 0:
                if student_sample_type not in ("sde", "ode"):  # as in the reference, only a re-noising step can object
                    raise NotImplementedError(
                        f"student_sample_type must be one of 'sde', 'ode' but got {student_sample_type}")
                eps = torch.randn_like(x0) if student_sample_type == "sde" else sched.x0_to_eps(xt=x, x0=x0, t=t)
                x = sched.forward_process(x0, eps, t_list[i + 1].expand(n))
                if keep is not None:
                    x = keep(x, condition)
        return x0

    @classmethod
    def generator_fn(cls, net, noise: torch.Tensor, student_sample_steps: int = 1, t_list: Optional[List[float]] = None,
                     data: torch.Tensor = None, precision_amp: Optional[torch.dtype] = None, **kwargs) -> torch.Tensor:
        """Single- or multi-step generation with the distilled network (methods/model.py:374-420).

        Extra keyword arguments understood by the fused path: eps (injected 'sde' noise, [steps-1,B,C,H,W]),
        seed (device RNG seed), use_graph (default True)."""
        with inference_mode(net, precision_amp=precision_amp, device_type=noise.device.type):
            if t_list is None:
                t_list = net.noise_scheduler.get_t_list(sample_steps=student_sample_steps, device="cpu")
            else:
                assert len(t_list) - 1 == student_sample_steps, (
                    f"t_list length (excluding zero) != student_sample_steps: {len(t_list) - 1} != {student_sample_steps}")
                t_list = torch.tensor(t_list, dtype=net.noise_scheduler.t_precision)
            assert t_list[-1].item() == 0, "t_list[-1] must be zero"
            fused = (getattr(net, "supports_fused_loop", lambda k: False)(cls._fused_loop) and data is None
                     and not hasattr(net, "preserve_conditioning") and cls._student_sample_loop.__func__ in _FUSED_LOOPS)
            if fused:
                kw = dict(kwargs)
                out = net.few_step_sample(noise, kw.pop("condition", None), t_list,
                                          sample_type=kw.pop("student_sample_type", "sde"), eps=kw.pop("eps", None),
                                          seed=kw.pop("seed", None), use_graph=kw.pop("use_graph", True),
                                          loop=cls._fused_loop)
                if kw:
                    raise TypeError(f"unexpected generator_fn kwargs: {sorted(kw)}")
                return out.to(dtype=noise.dtype)
            t_dev = t_list.to(noise.device)
            latents = net.noise_scheduler.latents(noise=noise, t_init=t_dev[0])
            if data is not None:
                latents = latents + data
            for k in ("eps", "seed", "use_graph"):
                kwargs.pop(k, None)
            return cls._student_sample_loop(net, latents, t_list=t_dev, **kwargs).to(dtype=noise.dtype)


_FUSED_LOOPS.add(FastGenModel._student_sample_loop.__func__)
