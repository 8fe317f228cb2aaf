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
        """Block by block: denoise from t_list[0] down to the block's sampled exit step, keep that step's x0 prediction as the block's
        output, then the cache-fill call on it (self_forcing.py:92-241) - one `fg_wan_sampler_run` call with the exit steps."""
        net, cfg = self.net, self.config
        num_frames = noise.shape[2]
        chunk_size = net.chunk_size
        num_blocks, remaining = num_frames // chunk_size, num_frames % chunk_size
        steps = cfg.student_sample_steps
        sched = net.noise_scheduler
        end_steps = self._sample_denoising_end_steps(num_blocks)
        t_list = cfg.sample_t_cfg.t_list
        if t_list is None:
            t_list = sched.get_t_list(steps, device="cpu")
        else:
            assert len(t_list) - 1 == steps, f"t_list length (excluding zero) != student_sample_steps: {len(t_list) - 1} != {steps}"
            t_list = torch.tensor(t_list, dtype=sched.t_precision)
        if num_blocks == 0:
            net.clear_caches()
            return torch.empty_like(noise)
        # the exit step of a block from start_gradient_frame on runs with autograd in the reference; this network has no backward
        starts = [0 if b == 0 else chunk_size * b + remaining for b in range(num_blocks)]
        if (enable_gradient and torch.is_grad_enabled() and any(s >= start_gradient_frame for s in starts)
                and any(p.requires_grad for p in net.parameters())):
            raise NotImplementedError("fastgen_amd.CausalWan: the backward pass is not implemented (the exit step of the rollout would run "
                                      "with gradients); call under torch.no_grad() or with enable_gradient=False")
        exits = [end_steps[0] if cfg.same_step_across_blocks else end_steps[b] for b in range(num_blocks)]
        with torch.no_grad():
            return net.student_sample(noise.clone(), t_list, condition, sample_type=cfg.student_sample_type,
                                      context_noise=cfg.context_noise or 0.0, exit_steps=exits)

    def gen_data_from_net(self, input_student: torch.Tensor, t_student: torch.Tensor, condition: Optional[Any] = None) -> torch.Tensor:
        del t_student
        return self.rollout_with_gradient(noise=input_student, condition=condition, enable_gradient=self.config.enable_gradient_in_rollout,
                                          start_gradient_frame=self.config.start_gradient_frame)
