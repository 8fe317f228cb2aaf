"""`SelfForcingModel.rollout_with_gradient` with the reference's signature (fastgen/methods/distribution_matching/self_forcing.py:92-241):
the training-time generator of Self-Forcing - chunk by chunk, denoise from t_list[0] down to a sampled exit step (the same
for all chunks or one per chunk), keep that step's x0 prediction as the chunk's output, then one more network call on it
(optionally re-noised to `context_noise`) that fills the KV cache.  Sampling at test time is CausVid's loop (inherited).

The exit step of the reference runs with autograd enabled when `enable_gradient` is set and gradients are on; the causal
video DiT of this package has no backward yet and says so (NotImplementedError from its forward) - under `torch.no_grad()`
or with `enable_gradient=False` the rollout is exactly the reference's sequence of network calls."""
from __future__ import annotations

from typing import Any, List, Optional

import torch
import torch.distributed as dist

from fastgen_amd.methods.distribution_matching.causvid import CausVidModel


class SelfForcingModel(CausVidModel):
    def __init__(self, config, net=None, device: Optional[torch.device] = None):
        """config: the reference's ModelConfig fields read here - student_sample_steps, student_sample_type, sample_t_cfg.t_list,
        same_step_across_blocks, last_step_only, context_noise, enable_gradient_in_rollout, start_gradient_frame
        (configs/methods/config_self_forcing.py:23-29).  net: the causal network (the reference builds it from config.net)."""
        self.config = config
        self.net = net
        self.device = device if device is not None else (next(net.parameters()).device if net is not None else torch.device("cpu"))

    def _sample_denoising_end_steps(self, num_blocks: int) -> List[int]:
        """One exit index per block, drawn on rank 0 and broadcast (self_forcing.py:73-90)."""
        steps = self.config.student_sample_steps
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not multi or dist.get_rank() == 0:
            if self.config.last_step_only:
                idx = torch.full((num_blocks,), steps - 1, dtype=torch.long, device=self.device)
            else:
                idx = torch.randint(low=0, high=steps, size=(num_blocks,), device=self.device)
        else:
            idx = torch.empty(num_blocks, dtype=torch.long, device=self.device)
        if multi:
            dist.broadcast(idx, src=0)
        return idx.tolist()

    def rollout_with_gradient(self, noise: torch.Tensor, condition: Optional[Any] = None, enable_gradient: bool = True,
                              start_gradient_frame: int = 0) -> torch.Tensor:
        net, cfg = self.net, self.config
        net.clear_caches()
        batch_size, _, num_frames = noise.shape[:3]
        chunk_size = net.chunk_size
        num_blocks, remaining = num_frames // chunk_size, num_frames % chunk_size
        steps = cfg.student_sample_steps
        sched = net.noise_scheduler
        end_steps = self._sample_denoising_end_steps(num_blocks)
        t_list = cfg.sample_t_cfg.t_list
        if t_list is None:
            t_list = sched.get_t_list(steps, device=noise.device)
        else:
            assert len(t_list) - 1 == steps, f"t_list length (excluding zero) != student_sample_steps: {len(t_list) - 1} != {steps}"
            t_list = torch.tensor(t_list, device=noise.device, dtype=sched.t_precision)
        call = dict(condition=condition, cache_tag="pos", fwd_pred_type="x0", is_ar=True)

        blocks = []
        for b in range(num_blocks):
            start = 0 if b == 0 else chunk_size * b + remaining
            end = chunk_size * (b + 1) + remaining
            x = noise[:, :, start:end]
            exit_step = end_steps[0] if cfg.same_step_across_blocks else end_steps[b]
            for step, t_cur in enumerate(t_list):
                t = t_cur.expand(batch_size)
                if step != exit_step:
                    with torch.no_grad():
                        x0 = net(x, t, store_kv=False, cur_start_frame=start, **call)
                    t_next = t_list[step + 1].expand(batch_size)
                    if cfg.student_sample_type == "sde":
                        eps = torch.randn_like(x0)
                    elif cfg.student_sample_type == "ode":
                        eps = sched.x0_to_eps(xt=x, x0=x0, t=t)
                    else:
                        raise NotImplementedError(f"student_sample_type must be one of 'sde', 'ode' but got {cfg.student_sample_type}")
                    x = sched.forward_process(x0, eps, t_next)
                else:
                    grad = enable_gradient and torch.is_grad_enabled() and start >= start_gradient_frame
                    with torch.set_grad_enabled(grad):
                        x0 = net(x, t, store_kv=False, cur_start_frame=start, **call)
                    break
            blocks.append(x0)
            with torch.no_grad():
                if cfg.context_noise > 0:
                    t_cache = torch.full((batch_size,), cfg.context_noise, device=noise.device, dtype=noise.dtype)
                    x_cache = sched.forward_process(x0, torch.randn_like(x0), t_cache)
                else:
                    x_cache, t_cache = x0, torch.zeros(batch_size, device=noise.device, dtype=noise.dtype)
                net(x_cache, t_cache, store_kv=True, cur_start_frame=start, **call)
        out = torch.cat(blocks, dim=2) if blocks else torch.empty_like(noise)
        net.clear_caches()
        return out

    def gen_data_from_net(self, input_student: torch.Tensor, t_student: torch.Tensor, condition: Optional[Any] = None) -> torch.Tensor:
        del t_student
        return self.rollout_with_gradient(noise=input_student, condition=condition, enable_gradient=self.config.enable_gradient_in_rollout,
                                          start_gradient_frame=self.config.start_gradient_frame)
