"""Time the student network's training-side work on one MI355X: forward (fg_edm_forward) and forward+backward
(fg_edm_backward: forward keeping block inputs, block-wise recomputation, all parameter gradients) at a training batch.
Algorithmic work: 42.383 GFLOP / image forward, 2x that for the backward (data + weight gradients)."""
import sys
import time

import torch

from fastgen_amd.networks.EDM.network import EDMPrecond

KW = dict(img_resolution=32, img_channels=3, label_dim=10, model_type="SongUNet", augment_dim=9, model_channels=128,
          channel_mult=[2, 2, 2], num_blocks=4, attn_resolutions=[16], embedding_type="positional", encoder_type="standard",
          decoder_type="standard", resample_filter=[1, 1], dropout=0.0)
MODE = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--mode=")), "bf16")  # bf16 | bf16x3 (the fp32-grade mode)
net = EDMPrecond(compute_dtype=MODE, **KW).randomize_parameters_(seed=1).cuda().eval()
print("compute mode:", MODE)
ONLY = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]  # e.g. --only=backward: time (profile) that leg alone
for B in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [64, 128]:
    x = torch.randn(B, 3, 32, 32, device="cuda") * 3
    t = torch.full((B,), 2.5, dtype=torch.float64, device="cuda")
    cond = torch.nn.functional.one_hot(torch.arange(B, device="cuda") % 10, 10).float()

    def fwd():
        with torch.no_grad():
            return net(x, t, condition=cond)

    def fwd_bwd():
        net.zero_grad(set_to_none=True)
        net(x, t, condition=cond).square().mean().backward()

    vx = torch.randn_like(x)

    def fwd_jvp():
        return net.jvp(x, t, vx, torch.ones(B, device="cuda"), condition=cond)

    for name, fn, mult in (("forward", fwd, 1.0), ("forward + backward", fwd_bwd, 3.0), ("forward + jvp (fg_edm_jvp)", fwd_jvp, 2.0)):
        if ONLY and not any(o in name for o in ONLY):
            continue
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"B={B:4d} {name:28s}: {dt * 1e3:8.2f} ms  {B / dt:8.1f} img/s  {B * 42.383e9 * mult / dt / 1e12:7.1f} algorithmic TFLOP/s")
