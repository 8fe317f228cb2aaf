"""`CausVidModel` sampling classmethods with the reference's signatures (fastgen/methods/distribution_matching/causvid.py:87-185):
the chunk-by-chunk student loop of the causal video DiT - per chunk N x {x0 prediction over the cached frames + this chunk;
re-noise to the next timestep}, then one network call on the finished chunk that fills the KV cache."""
from __future__ import annotations

from typing import Any, Optional

import torch

from fastgen_amd.methods.model import FastGenModel, inference_mode


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

    @classmethod
    def generator_fn_extrapolation(cls, net, noise: torch.Tensor, condition: Any = None, *, num_segments: int, overlap_frames: int,
                                   student_sample_steps: int = 1, student_sample_type: str = "sde", t_list=None,
                                   precision_amp: Optional[torch.dtype] = None, context_noise: Optional[float] = 0, **kwargs) -> torch.Tensor:
        """Several segments one after the other (causvid.py:188-397): every segment is the chunked student loop over its own cleared
        KV caches; with `overlap_frames` > 0 the last frames of a finished segment are decoded and re-encoded by `net.vae` (first
        overlapped latent) or reused as they are (the rest), placed at the head of the next segment and run through the network at t = 0
        to fill the caches before its remaining frames are generated.  Returns [B, C, num_segments * T - (num_segments - 1) * overlap, H, W]."""
        with inference_mode(net, precision_amp=precision_amp, device_type=noise.device.type):
            if num_segments < 1:
                raise ValueError("num_segments must be >= 1")
            if overlap_frames > 0 and getattr(net, "vae", None) is None:
                raise ValueError("generator_fn_extrapolation requires a VAE instance via `vae` when overlap_frames > 0")
            batch_size, _, segment_frames = noise.shape[:3]
            dtype, device, chunk_size, sched = noise.dtype, noise.device, net.chunk_size, net.noise_scheduler
            if segment_frames % chunk_size != 0:
                raise ValueError(f"Segment length {segment_frames} must be divisible by chunk_size {chunk_size}")
            if overlap_frames < 0 or overlap_frames >= segment_frames:
                raise ValueError("overlap_frames must be in [0, segment_frames)")
            if overlap_frames % chunk_size != 0:
                raise ValueError("overlap_frames must be divisible by chunk_size")
            if t_list is None:
                t_list = sched.get_t_list(student_sample_steps, device=device).to(torch.float32)
            else:
                assert len(t_list) - 1 == student_sample_steps, (
                    f"t_list length (excluding zero) != student_sample_steps: {len(t_list) - 1} != {student_sample_steps}")
                t_list = torch.tensor(t_list, device=device, dtype=torch.float32)
            assert t_list[-1].item() == 0, "t_list[-1] must be zero"
            call = dict(condition=condition, fwd_pred_type="x0", cache_tag="pos", is_ar=True, **kwargs)

            def run_segment(segment_latents: torch.Tensor, prefill_frames: int) -> torch.Tensor:
                x = segment_latents.clone()
                net.clear_caches()
                for start in range(0, prefill_frames, chunk_size):  # the bridged head: cache fill at t = 0
                    net(x[:, :, start:min(start + chunk_size, prefill_frames)], t_list[-1].expand(batch_size), cur_start_frame=start, store_kv=True,
                        **call)
                if prefill_frames == 0:
                    x = sched.latents(x, t_init=t_list[0])
                else:
                    x[:, :, prefill_frames:] = sched.latents(x[:, :, prefill_frames:], t_init=t_list[0])
                for start in range(prefill_frames, segment_frames, chunk_size):
                    end = min(start + chunk_size, segment_frames)
                    x_next = x[:, :, start:end]
                    for step in range(len(t_list) - 1):
                        t_cur = t_list[step].expand(batch_size)
                        x_cur = x_next
                        x_next = net(x_cur, t_cur, cur_start_frame=start, store_kv=False, **call)
                        t_next = t_list[step + 1]
                        if t_next > 0:
                            if student_sample_type == "sde":
                                eps = torch.randn_like(x_next)
                            elif student_sample_type == "ode":
                                eps = sched.x0_to_eps(xt=x_cur, x0=x_next, t=t_cur)
                            else:
                                raise NotImplementedError(f"student_sample_type must be one of 'sde', 'ode' but got {student_sample_type}")
                            x_next = sched.forward_process(x_next, eps, t_next.expand(batch_size))
                    x[:, :, start:end] = x_next
                    x_cache, t_cache = x_next, t_list[-1].expand(batch_size)
                    if context_noise and context_noise > 0:
                        t_cache = torch.full((batch_size,), context_noise, device=device, dtype=dtype)
                        x_cache = sched.forward_process(x_next, torch.randn_like(x_next), t_cache)
                    net(x_cache, t_cache, cur_start_frame=start, store_kv=True, **call)
                net.clear_caches()
                return x

            segments, current, prefill = [], noise, 0
            for i in range(num_segments):
                seg = run_segment(current, prefill)
                segments.append(seg if i == 0 or overlap_frames == 0 else seg[:, :, overlap_frames:])
                if i == num_segments - 1:
                    break
                if overlap_frames == 0:
                    current, prefill = torch.randn_like(noise), 0
                    continue
                tail = net.vae.encode(net.vae.decode(seg)[:, :, -overlap_frames:]).to(dtype=dtype, device=device)
                if overlap_frames > 1:  # all but the first overlapped latent are reused directly
                    tail = torch.cat([tail[:, :, :1], seg[:, :, -(overlap_frames - 1):]], dim=2)
                current = torch.randn_like(seg)
                current[:, :, :overlap_frames] = tail
                prefill = overlap_frames
            net.clear_caches()
            return torch.cat(segments, dim=2).to(dtype=noise.dtype)
