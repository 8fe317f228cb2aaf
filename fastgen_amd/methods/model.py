"""Student sampling entry points with the reference's signatures: `FastGenModel.generator_fn` and
`FastGenModel._student_sample_loop` (fastgen/methods/model.py:315-420).

For a `fastgen_amd` EDM network the whole loop is one library call (`fg_sampler_run`: latents -> N x {U-Net forward,
re-noise} as a replayed hipGraph, host syncs of the reference loop hoisted to host scalars).  Any other
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


class FastGenModel:
    """Only the sampling classmethods of the reference class; they are what `scripts/inference/*`, `scripts/fid/*` and
    the wandb callback call."""

    # which fg_sampler_run loop restates this class's _student_sample_loop (see EDMPrecond.fused_loop)
    _fused_loop = "x0"

    @classmethod
    def _student_sample_loop(cls, net, x: torch.Tensor, t_list: torch.Tensor, condition: Any = None,
                             student_sample_type: str = "sde", **kwargs) -> torch.Tensor:
        """Generic per-step loop (methods/model.py:315-372) for networks without a fused sampler."""
        batch_size = x.shape[0]
        has_hook = hasattr(net, "preserve_conditioning")
        x_pred = x
        for t_cur, t_next in zip(t_list[:-1], t_list[1:]):
            t_batch = t_cur.expand(batch_size)
            x_pred = net(x, t_batch, condition=condition, fwd_pred_type="x0")
            if has_hook:
                x_pred = net.preserve_conditioning(x_pred, condition)
            if t_next > 0:
                if student_sample_type == "sde":
                    eps = torch.randn_like(x_pred)
                elif student_sample_type == "ode":
                    eps = net.noise_scheduler.x0_to_eps(xt=x, x0=x_pred, t=t_batch)
                else:
                    raise NotImplementedError(
                        f"student_sample_type must be one of 'sde', 'ode' but got {student_sample_type}")
                x = net.noise_scheduler.forward_process(x_pred, eps, t_next.expand(batch_size))
                if has_hook:
                    x = net.preserve_conditioning(x, condition)
        return x_pred

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
            fused = (isinstance(net, EDMPrecond) and net.fused_loop() == cls._fused_loop and data is None
                     and not hasattr(net, "preserve_conditioning"))
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
