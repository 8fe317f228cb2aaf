"""Time one supervised (denoising score matching) training iteration of the EDM CIFAR-10 SFT config
(configs/experiments/EDM/config_sft_edm_cifar10.py: dropout 0.13, augmentation labels, x0-prediction loss) on one MI355X with the
network on the fastgen_amd module, synthetic data: x_t = x0 + t eps, loss = mse(net(x_t, t, {"aug_condition", "orig_condition"}), x0),
AdamW.  A measurement script; loss weighting and the augmentation of the images themselves are omitted.
Usage: python scripts/sft_step_bench.py [batch ...]"""
import sys
import time

import torch

from fastgen_amd.networks.EDM.network import EDMPrecond

KW = dict(img_resolution=32, img_channels=3, label_dim=10, model_type="SongUNet", augment_dim=9, model_channels=128,
          channel_mult=[2, 2, 2], num_blocks=4, attn_resolutions=[16], embedding_type="positional", encoder_type="standard",
          decoder_type="standard", resample_filter=[1, 1], dropout=0.13)
dev = torch.device("cuda")
net = EDMPrecond(compute_dtype="bf16", **KW).randomize_parameters_(seed=1).to(dev).train()
opt = torch.optim.AdamW(net.parameters(), lr=1e-5)
sched = net.noise_scheduler

for B in [int(a) for a in sys.argv[1:]] or [64, 128]:
    g = torch.Generator(device=dev).manual_seed(B)
    x0 = torch.randn(B, 3, 32, 32, device=dev, generator=g).clamp(-1, 1)
    eps = torch.randn(B, 3, 32, 32, device=dev, generator=g)
    cond = {"orig_condition": torch.nn.functional.one_hot(torch.arange(B, device=dev) % 10, 10).float(),
            "aug_condition": torch.randn(B, 9, device=dev, generator=g)}
    t = sched.sample_t(B, time_dist_type="lognormal", device=dev)

    def step():
        x_t = sched.forward_process(x0, eps, t)
        loss = torch.nn.functional.mse_loss(net(x_t, t, condition=cond, fwd_pred_type="x0"), x0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:4d} SFT iteration (dropout 0.13, augment labels; forward + backward + AdamW): {dt * 1e3:8.2f} ms  {B / dt:8.1f} img/s per GPU")
    assert all(torch.isfinite(p).all() for p in net.parameters())
