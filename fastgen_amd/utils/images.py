"""Samples -> image bytes on the device: the step that follows `generator_fn` in the reference's sample writer
(scripts/fid/compute_fid_from_ckpts.py:199,
`(images * 127.5 + 128).clip(0, 255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()`).

Done by one HIP kernel (fg_op_images_to_u8, csrc/misc.hip) on the stream the sampler ran on, so what crosses PCIe is
3 KB per CIFAR image instead of 12 KB and no fp32 NHWC intermediate is materialised.  No CPU fallback."""
from __future__ import annotations

import torch

from .. import _lib


def images_to_uint8(images: torch.Tensor) -> torch.Tensor:
    """[B, C, H, W] float samples in [-1, 1] -> [B, H, W, C] uint8 on the same device (fp32 multiply, add, clip, truncate).
    `images_to_uint8(x).cpu().numpy()` is what the reference hands to PIL."""
    if images.ndim != 4:
        raise ValueError(f"expected [B, C, H, W] images, got shape {tuple(images.shape)}")
    if not images.is_cuda:
        raise RuntimeError("fastgen_amd: images_to_uint8 runs on the GPU only (no CPU fallback)")
    x = images.detach().to(torch.float32).contiguous()
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), dtype=torch.uint8, device=x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _lib.check(_lib.lib().fg_op_images_to_u8(x.data_ptr(), out.data_ptr(), B, C, H, W, stream))
    return out
