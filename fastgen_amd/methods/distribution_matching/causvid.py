"""`CausVidModel` sampling classmethods with the reference's signatures (fastgen/methods/distribution_matching/causvid.py:87-185):
the chunk-by-chunk student loop of the causal video DiT - per chunk N x {x0 prediction over the cached frames + this chunk;
re-noise to the next timestep}, then one network call on the finished chunk that fills the KV cache."""
from __future__ import annotations

from typing import Any, Optional

import torch

from fastgen_amd.methods.model import FastGenModel


class CausVidModel(FastGenModel):
    @classmethod
    def _student_sample_loop(cls, net, x: torch.Tensor, t_list: torch.Tensor, condition: Any = None, student_sample_type: str = "sde",
                             context_noise: Optional[float] = 0, **kwargs) -> torch.Tensor:
        net.clear_caches()
        batch_size, num_frames = x.shape[0], x.shape[2]
        chunk_size = net.chunk_size
        num_chunks, remaining = num_frames // chunk_size, num_frames % chunk_size
        sched = net.noise_scheduler
        for i in range(max(1, num_chunks)):
            if num_chunks == 0:
                start, end = 0, remaining
            else:
                start = 0 if i == 0 else chunk_size * i + remaining
                end = chunk_size * (i + 1) + remaining
            x_next = x[:, :, start:end, ...]
            for step in range(len(t_list) - 1):
                t_cur = t_list[step].expand(batch_size)
                x_cur = x_next
                x_next = net(x_cur, t_cur, condition=condition, fwd_pred_type="x0", cache_tag="pos", cur_start_frame=start, store_kv=False,
                             is_ar=True, **kwargs)
                t_next = t_list[step + 1]
                if t_next > 0:
                    if student_sample_type == "sde":
                        eps = torch.randn_like(x_next)
                    elif student_sample_type == "ode":
                        eps = sched.x0_to_eps(xt=x_cur, x0=x_next, t=t_cur)
                    else:
                        raise NotImplementedError(f"student_sample_type must be one of 'sde', 'ode' but got {student_sample_type}")
                    x_next = sched.forward_process(x_next, eps, t_next.expand(batch_size))
            x[:, :, start:end, ...] = x_next
            x_cache, t_cache = x_next, t_list[-1].expand(batch_size)
            if context_noise > 0:
                t_cache = torch.full((batch_size,), context_noise, device=x.device, dtype=x.dtype)
                x_cache = sched.forward_process(x_next, torch.randn_like(x_next), t_cache)
            net(x_cache, t_cache, condition=condition, fwd_pred_type="x0", cache_tag="pos", cur_start_frame=start, store_kv=True, is_ar=True,
                **kwargs)
        net.clear_caches()
        return x
