"""`CausVidModel` sampling classmethods with the reference's signatures (fastgen/methods/distribution_matching/causvid.py:87-185):
the chunk-by-chunk student loop of the causal video DiT - per chunk N x {x0 prediction over the cached frames + this chunk;
re-noise to the next timestep}, then one network call on the finished chunk that fills the KV cache.  The loop itself lives in the
library (`fg_wan_sampler_run`, csrc/engine_sampler.inc); these are thin callers."""
from __future__ import annotations

from typing import Any, Optional

import torch

from fastgen_amd.methods.model import FastGenModel, inference_mode


class CausVidModel(FastGenModel):
    @classmethod
    def _student_sample_loop(cls, net, x: torch.Tensor, t_list: torch.Tensor, condition: Any = None, student_sample_type: str = "sde",
                             context_noise: Optional[float] = 0, **kwargs) -> torch.Tensor:
        """The reference's signature (causvid.py:87-98) over `CausalWan.student_sample`: the whole loop - chunking, the N denoising calls
        and the cache-fill call per chunk, re-noising, RNG - runs inside the library (`fg_wan_sampler_run`, one hipGraph per chunk)."""
        return net.student_sample(x, t_list, condition, sample_type=student_sample_type, context_noise=context_noise or 0.0, **kwargs)

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

            def run_segment(segment_latents: torch.Tensor, prefill_frames: int) -> torch.Tensor:
                # the bridged head [0, prefill_frames) fills the caches at t = 0 as it is; the rest starts from latents at t_list[0]
                x = segment_latents.clone()
                if prefill_frames == 0:
                    x = sched.latents(x, t_init=t_list[0])
                else:
                    x[:, :, prefill_frames:] = sched.latents(x[:, :, prefill_frames:], t_init=t_list[0])
                return net.student_sample(x, t_list, condition, sample_type=student_sample_type, context_noise=context_noise or 0.0,
                                          prefill_frames=prefill_frames, **kwargs)

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
