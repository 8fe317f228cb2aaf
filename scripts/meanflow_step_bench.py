"""Time one MeanFlow training iteration of the CIFAR-10 config (configs/experiments/EDM/config_mf_cifar10.py) on one MI355X with the
network on the fastgen_amd module, synthetic data.  Structure of the reference's `MeanFlowModel.single_train_step` (JVP branch,
fastgen/methods/consistency_model/mean_flow.py:240-335): x_t on the rectified-flow path, u = net(x_t, t, r) with grad, the tangent
d/dt u along (v, 1, 0) by one fg_edm_jvp call (no grad), target = v - (t - r) * du/dt, loss = mse(u, target.detach()), AdamW.
A measurement script; the loss weighting of the reference is omitted.   Usage: python scripts/meanflow_step_bench.py [batch ...]"""
import sys
import time

import torch

from fastgen_amd.methods.consistency_model.mean_flow import MeanFlowModel
from fastgen_amd.networks.EDM.network import EDMPrecond

KW = dict(img_resolution=32, img_channels=3, label_dim=0, model_type="SongUNet", augment_dim=6, model_channels=128,
          channel_mult=[2, 2, 2], num_blocks=4, attn_resolutions=[16], embedding_type="positional", encoder_type="standard",
          decoder_type="standard", resample_filter=[1, 1], dropout=0.0, r_timestep=True, drop_precond="both", schedule_type="rf",
          net_pred_type="flow")
dev = torch.device("cuda")
net = EDMPrecond(compute_dtype="bf16", **KW).randomize_parameters_(seed=1).to(dev).eval()
opt = torch.optim.AdamW(net.parameters(), lr=1e-5)
sched = net.noise_scheduler


def step(B, x0, eps, t, r):
    x_t = sched.forward_process(x0, eps, t)
    v = eps - x0                                                   # conditional velocity of the rectified-flow path
    # the tangent first: it shares the training workspace, and a differentiable forward issued before it would have to be
    # recomputed by the backward
    du_dt = MeanFlowModel.network_jvp(net, x_t, t, r, v)           # d/dt u(x_t + s v, t + s, r): fg_edm_jvp, detached
    u = net(x_t, t, r=r, fwd_pred_type="flow")
    target = v - (t - r).reshape(B, 1, 1, 1).float() * du_dt
    loss = torch.nn.functional.mse_loss(u, target.detach())
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for B in [int(a) for a in sys.argv[1:]] or [64, 128]:
    g = torch.Generator(device=dev).manual_seed(B)
    x0 = torch.randn(B, 3, 32, 32, device=dev, generator=g).clamp(-1, 1)
    eps = torch.randn(B, 3, 32, 32, device=dev, generator=g)
    t = sched.sample_t(B, device=dev)
    r = (t * torch.rand(B, device=dev, dtype=torch.float64, generator=g)).clamp(min=0.0)
    for _ in range(2):
        step(B, x0, eps, t, r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        step(B, x0, eps, t, r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:4d} MeanFlow iteration (forward + jvp + backward + AdamW): {dt * 1e3:8.2f} ms  {B / dt:8.1f} img/s per GPU")
    assert all(torch.isfinite(p).all() for p in net.parameters())
